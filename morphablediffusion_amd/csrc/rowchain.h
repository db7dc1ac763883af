// Row-chain kernel (k_rowchain.hip): the row-local tail of a SpatialTransformer block as ONE launch,
//
//   [t2 = to_out(attn) + b + attn2 + t0]  ->  LayerNorm3  ->  FF1  ->  GEGLU  ->  FF2  ->  + t2   [-> proj_out -> + x_in]
//
// (reference ldm/modules/attention.py:37-73 FeedForward / GEGLU, :265-269 BasicTransformerBlock._forward, :325-336
// SpatialTransformer.forward).  Every one of these layers acts on a pixel row by itself, so a wave keeps its 32 pixel rows in
// registers from the first layer to the last and only the weights move: they are pre-packed (rowchain_pack) into ONE stream
// of 1 KiB MFMA A-operand fragments in exactly the order the kernel consumes them and flow L2 -> LDS through a ring of 128
// fragments filled by buffer_load ... lds.  Shared by the pack kernel, the launcher and the kernel: the stream layout.
#pragma once

struct RcLayout {
  int C, ao, po;
  int F;         // 32-channel blocks of a row (C / 32)
  int KC;        // 16-wide reduction steps over the channels (C / 16)
  int K1;        // ... + 1: the step that carries the FF1 bias (augmented K: activations 1, weights bias hi / lo)
  int NU;        // hidden units of 32 GEGLU outputs (4C / 32)
  int G1, G2;    // fragments of one unit's FF1 part (value + gate per step) and FF2 part
  int BODY_RAW, BODY, UNR;
  int AO_N, PRO_REAL, PRO, PRO_PAD, LOOP, TAIL_REAL, TAIL, NT, NT_ALLOC;
};

// Stream (positions in fragments; the ring slot of position q is q % 128, a half-body = 32 consecutive positions):
//   [0, PRO)              : PRO_PAD unused | to_out (AO_N = KC * F, order (k-step, row block)) | FF1 of unit 0 | FF1 of unit 1
//   PRO + u * BODY, u < NU: FF1 of unit u + 2 (G1; zeros past the last unit) | FF2 of unit u (G2, order (half, row block)) | pad
//                           (a three-stage pipeline: FF1 of unit u + 2 and FF2 of unit u run beside the GEGLU of unit u + 1)
//   PRO + LOOP            : FF2 bias (F) | proj_out (KC * F) | pad to a half-body     -- TAIL
//   + 96 positions of slack: the ring runs three half-bodies ahead without a bounds test
constexpr RcLayout rc_layout(int C, bool ao, bool po) {
  RcLayout L{};
  L.C = C; L.ao = ao; L.po = po;
  L.F = C / 32; L.KC = C / 16; L.K1 = L.KC + 1; L.NU = C / 8;
  L.G1 = 2 * L.K1; L.G2 = 2 * L.F; L.BODY_RAW = L.G1 + L.G2;
  L.BODY = (L.BODY_RAW + 31) / 32 * 32;
  L.UNR = 128 / L.BODY;
  L.AO_N = ao ? L.KC * L.F : 0;
  L.PRO_REAL = L.AO_N + 2 * L.G1;
  L.PRO = (L.PRO_REAL + 127) / 128 * 128;
  L.PRO_PAD = L.PRO - L.PRO_REAL;
  L.LOOP = L.NU * L.BODY;
  L.TAIL_REAL = L.F + (po ? L.KC * L.F : 0);
  L.TAIL = (L.TAIL_REAL + 31) / 32 * 32;
  L.NT = L.PRO + L.LOOP + L.TAIL;
  L.NT_ALLOC = L.NT + 96;
  return L;
}
constexpr bool rc_supported_c(int C) { return C == 64 || C == 128 || C == 256 || C == 320; }

// fp32 sources in the reference's layouts (device pointers); the stream bakes LayerNorm3's gain into FF1's columns and its
// bias (through FF1) into FF1's bias
struct RcWeights {
  const float* w_ao;   // [C][C]     attn1.to_out.0.weight          (ao only)
  const float* ln_g;   // [C]        norm3.weight
  const float* ln_b;   // [C]        norm3.bias
  const float* w1;     // [8C][C]    ff.net.0.proj.weight (rows 0..4C-1 value, 4C..8C-1 gate)
  const float* b1;     // [8C]
  const float* w2;     // [C][4C]    ff.net.2.weight
  const float* b2;     // [C]
  const float* w_po;   // [C][C]     proj_out.weight (1x1 conv)     (po only)
};

struct RowChain {
  const half_t* stream;
  int rows, T;          // T: rows per sample (a multiple of 32)
  const half_t* ao;     // ao: attention output fp16 [rows][ld_ao]
  int ld_ao;
  const float* xin;     // ao: t0 (the residual the to_out projection adds), else t2 (the input of LayerNorm3); fp32 [rows][ld_x]
  int ld_x;
  const float* b_ao;    // ao: to_out bias [C]
  const float* rowbias; // ao: per-sample constant (attn2 over the single context token) [B][rb_ld] or null
  int rb_ld;
  const float* b_po;    // po: proj_out bias [C]
  const float* resid;   // po: the block input fp32 [rows][ld_r]
  int ld_r;
  void* out;            // po: fp32 [rows][ld_o]; else fp16 [rows][ld_o] (x + ff(x): the proj_out operand)
  int ld_o;
  int out_split;        // !po: 0, or C -> [hi | lo | hi] rows of 3C halfs (extended-precision consumer, IGemm::out_split)
};

bool rowchain_takes(int C, int rows, int T);
size_t rowchain_stream_halfs(int C, int ao, int po);
// tmp: 8C floats of scratch (the folded FF1 bias)
int rowchain_pack(const RcWeights& w, int C, int ao, int po, float* tmp, half_t* stream, hipStream_t s);
int launch_rowchain(const RowChain& p, int C, int ao, int po, hipStream_t s);
