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
//   PRO + LOOP            : FF2 bias (F) | proj_out (KC * F; extended precision: (hi, lo) per (k step, row block)) | pad to a half-body -- TAIL
//   + 96 positions of slack: the ring runs three half-bodies ahead without a bounds test
// po: 0 = no proj_out, 1 = proj_out, 2 = proj_out in extended precision (weights as hi and lo fragments, three products)
constexpr RcLayout rc_layout(int C, bool ao, int po) {
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
  L.TAIL_REAL = L.F + (po == 1 ? L.KC * L.F : (po == 2 ? 2 * L.KC * L.F : 0));
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

// ---- "row head" (k_rowchain.hip: rowhead_kernel): the row-local FRONT of the block, behind the GroupNorm and in front of the attention,
//   t0 = proj_in(n0) + b   ->   LayerNorm1   ->   q | k | v = to_q / to_k / to_v (no bias)        modules/attention.py:325-332, 266, 186-190
// Stream (C = 320 only): 56 unused | proj_in (20 k steps x 10 row blocks, natural k: its B operand is the fp16 GroupNorm output read
// from memory) | 10 groups of 64: (21 k steps x 3 row blocks of the stacked, LayerNorm-folded q|k|v matrix, the 21st step carrying
// W beta as fp16 hi + lo) + one unused position | 96 of slack
// With an extended-precision proj_in (xp) its section holds (hi, lo) weight fragments per (k step, row block) -- 400 fragments behind
// 112 unused -- and the GroupNorm output arrives as [hi | lo | hi] rows of 3C halfs.
constexpr int RH_C = 320, RH_F = 10, RH_KC = 20, RH_K1 = 21, RH_NG = 10, RH_BODY = 64;
constexpr int rh_pi(bool xp) { return xp ? 400 : 200; }
constexpr int rh_pro(bool xp) { return xp ? 512 : 256; }
constexpr int rh_pad(bool xp) { return rh_pro(xp) - rh_pi(xp); }
constexpr int rh_nt(bool xp) { return rh_pro(xp) + RH_NG * RH_BODY; }
constexpr int rh_nt_alloc(bool xp) { return rh_nt(xp) + 96; }
struct RhWeights {
  const float* w_pi;   // [C][C]   proj_in.weight (1x1 conv)
  const float* ln_g;   // [C]      norm1.weight
  const float* ln_b;   // [C]      norm1.bias
  const float* w_q;    // [C][C]   attn1.to_q.weight
  const float* w_k;    // [C][C]
  const float* w_v;    // [C][C]
};
struct RowHead {
  const half_t* stream;
  int rows;
  const half_t* n0;     // GroupNorm output fp16 [rows][ld_n0] ([hi | lo | hi], ld_n0 >= 3C, for the extended-precision form)
  int ld_n0;
  const float* b_pi;    // proj_in bias [C]
  float* t0;            // fp32 [rows][ld_t0]: the residual the to_out projection adds later
  int ld_t0;
  half_t* qkv;          // fp16 [rows][ld_qkv], columns q | k | v
  int ld_qkv;
};
size_t rowhead_stream_halfs(int xp);
// tmp: 3C floats of scratch (the folded q|k|v bias)
int rowhead_pack(const RhWeights& w, int xp, float* tmp, half_t* stream, hipStream_t s);
int launch_rowhead(const RowHead& p, int xp, hipStream_t s);

bool rowchain_takes(int C, int rows, int T);
// which (width, to_out, proj_out) forms of the kernel exist (launch_rc_any): every supported width has (1, 0) and (1, 1); the
// extended-precision proj_out (1, 2) only C = 64 and C = 320
bool rowchain_form_instantiated(int C, int ao, int po);
size_t rowchain_stream_halfs(int C, int ao, int po);
// tmp: 8C floats of scratch (the folded FF1 bias)
int rowchain_pack(const RcWeights& w, int C, int ao, int po, float* tmp, half_t* stream, hipStream_t s);
int launch_rowchain(const RowChain& p, int C, int ao, int po, hipStream_t s);
