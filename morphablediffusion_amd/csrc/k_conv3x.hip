// 3x3 stride-1 convolution, second form ("conv3x"): the halo-tile idea of k_conv3.hip on the execution shape that the row-chain
// kernel (k_rowchain.hip) showed to run the matrix pipe near its clock-limited rate:
//
//   * 4 waves, ONE per SIMD (up to 512 registers), each owning 64 output pixels (4 rows of a 16 x 16 block) x BN = 32 NF channels:
//     2 NF accumulator fragments.  Per 16-deep reduction step a wave reads 2 activation fragments and NF weight fragments for
//     2 NF MFMAs -- 0.7 KiB of LDS reads per MFMA instead of 1.2 (k_conv3.hip: 8 waves of 32 pixels, its LDS fragment traffic was
//     27 of its 63 us at 32 x 32 x 320 -> 320).
//   * TRANSPOSED product: weights are the MFMA A operand, activations the B operand, D = out^T[channel][pixel].  The weights are
//     pre-packed (conv3x_pack) per column tile as a stream of 1 KiB A fragments in consumption order (chunk, tap, k step, row
//     block), fragment-major: lane l reads its 16 bytes at 16 l -- conflict-free ds_read_b128 with immediate offsets, no swizzle
//     arithmetic.  A ring of 4 quarters x 3 steps filled by buffer_load ... lds, one piece per MFMA pair behind a quarter
//     boundary {s_waitcnt vmcnt(n); s_barrier}: the quarter after the current one has always landed, so the register prefetch
//     (weight fragments four ahead, activation fragments one step ahead) never stops at a boundary.
//   * The activation halo tile ((16 + 2)^2 pixels x 64 channels, 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7) is
//     double buffered and loaded by LDS-DMA during the previous chunk; zero padding = out-of-range buffer offsets.
//   * D has lane = pixel, registers = 4 consecutive channels: the epilogue writes each 32 x 32 block to a wave-private scratch with
//     four ds_write_b128 and stores full 128-byte row segments (bias, per-sample bias, fp32 residual as 16-byte loads).
// Scope: fp16 channels-last input, 3x3 / stride 1 / pad 1, images of side % 16 == 0, Cin % 64 == 0, N % (32 NF) == 0, fp32 output
// (or split-K partial slabs), no activation: the UNet's ResBlock convolutions at 32 x 32 and 16 x 16 (reference
// ldm/modules/diffusionmodules/openaimodel.py:202-233).  Everything else stays on k_conv3.hip / k_gemm.hip.
#include "common.h"
#include "igemm_epilogue.h"

namespace {

// A workgroup's 256 output pixels are one 16 x 16 block of an image (IW = 16) or four whole 8 x 8 images (IW = 8, one per wave);
// every block carries its own halo ring: (IW + 2)^2 halo pixels per block.
template <int IW>
struct CxGeo {
  static constexpr int NI = 256 / (IW * IW);           // image blocks per tile
  static constexpr int HW = IW + 2;                     // halo width
  static constexpr int ROWS = NI * HW * HW;             // pixels of a halo tile (324 / 400)
  static constexpr int GROUPS = (ROWS + 7) / 8;         // 8-row DMA pieces (41 / 50)
  static constexpr int BYTES = GROUPS * 1024;
  static constexpr int HP = (GROUPS + 3) / 4;           // halo pieces per wave and chunk (11 / 13; surplus ones repeat the last)
};

// XOR key of a halo pixel's 16-byte chunks, from its position (hy, hx) inside its (IW + 2)^2 halo block.  A ds_read_b128 is served in
// 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) on 16 slots of 16 bytes; slot = (row & 1) * 8 + (chunk ^ key).
// IW = 16: a group reads 16 CONSECUTIVE columns of two image rows (8 + 8): key = hx >> 1 gives 8 distinct keys per column parity.
// IW = 8: it reads 4 columns of four image rows: the row parity moves the key by 4.  Both are conflict-free for every tap; the
// key (row >> 1) & 7 of the linear halo row put two lanes of every group on one slot (three at IW = 8): 8 (12) instead of 4 LDS
// cycles per activation fragment (tools/lds_bank_model.py conv3x).
template <int IW>
__device__ __forceinline__ int halo_key(int hy, int hx) {
  if constexpr (IW == 16) return (hx >> 1) & 7;
  else return ((hx >> 1) + 4 * (hy & 1)) & 7;
}

template <int NF, int IW>
__global__ __launch_bounds__(256) void conv3x_kernel(const IGemm g, const half_t* __restrict__ wstream) {
#if defined(__HIP_DEVICE_COMPILE__)
  using Geo = CxGeo<IW>;
  constexpr int CX_HALO_ROWS = Geo::ROWS, CX_HALO_GROUPS = Geo::GROUPS, CX_HALO_BYTES = Geo::BYTES, CX_HP = Geo::HP;
  constexpr int NI = Geo::NI, HWD = Geo::HW;
  constexpr int QF = 3 * NF;                 // fragments per ring quarter (3 reduction steps)
  constexpr int PW = (QF + 3) / 4;           // DMA pieces per wave and quarter
  constexpr int RING_BYTES = 4 * QF * 1024;
  constexpr int NSTEP = 36;                  // reduction steps per 64-channel chunk: 9 taps x 4
  constexpr int NQ = NSTEP / 3;              // quarters per chunk (12)
  constexpr int CHUNK_FRAGS = NSTEP * NF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sHalo = smem;                        // [2][CX_HALO_BYTES]
  char* sRing = smem + 2 * CX_HALO_BYTES;    // [4][QF][1 KiB]
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = g.Y, W = g.X, N = g.N, Cin = g.Cin;
  const int bx_per = W / IW, by_per = H / IW, bpi = bx_per * by_per;  // image blocks per sample
  const int nblocks = g.B * bpi;
  const int tiles_m = (nblocks + NI - 1) / NI, tiles_n = N / (32 * NF);
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap: the column tiles of one pixel tile share an L2
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  if (g.xcd_cols) {  // deep levels (weights >> activations): an XCD keeps one column tile's stream against every pixel tile
    tn = bid / tiles_m;
    tm = bid - tn * tiles_m;
  }
  const int n0 = tn * 32 * NF;
  // block gb of the tile -> sample, top-left pixel
  auto block_pos = [&](int gb, int& b, int& y0, int& x0) {
    b = gb / bpi;
    const int brem = gb - b * bpi, by = brem / bx_per;
    y0 = by * IW;
    x0 = (brem - by * bx_per) * IW;
  };
  const int ncc = Cin / 64;
  int cc_beg = 0, cc_end = ncc;
  if (g.splitk > 1) {
    const int per = (ncc + g.splitk - 1) / g.splitk;
    cc_beg = blockIdx.y * per;
    cc_end = min(ncc, cc_beg + per);
  }
  const int nchunks = cc_end - cc_beg;

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), (short)0, 0xFFFFFFFEu, 0x00020000);
  // this column tile's stream: [chunk][36 steps][NF] fragments of 1 KiB
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(wstream) + (size_t)tn * ncc * CHUNK_FRAGS * 512, (short)0,
                                                       0xFFFFFFFEu, 0x00020000);

  // halo pieces of this wave: piece i covers halo rows (wave + 4 i) * 8 .. + 8 (piece indices past the tile repeat the last one)
  unsigned h_off[CX_HP];
  int h_grp[CX_HP];
#pragma unroll
  for (int i = 0; i < CX_HP; ++i) {
    const int grp = min(wave + 4 * i, CX_HALO_GROUPS - 1);
    h_grp[i] = grp;
    const int hp = grp * 8 + (lane >> 3);
    const int hj = hp / (HWD * HWD), hr = hp - hj * (HWD * HWD);
    const int hy = hr / HWD, hx = hr - hy * HWD;
    const int chunk = (lane & 7) ^ halo_key<IW>(hy, hx);
    int b, y0, x0;
    block_pos(tm * NI + hj, b, y0, x0);
    const int y = y0 + hy - 1, x = x0 + hx - 1;
    const bool ok = hp < CX_HALO_ROWS && tm * NI + hj < nblocks && y >= 0 && y < H && x >= 0 && x < W;
    const unsigned pix = (unsigned)((b * H + y) * W + x);
    h_off[i] = ok ? (pix * (unsigned)g.lda + chunk * 8) * 2 : 0xFFFFFFFFu;
  }
  auto dma_halo_piece = [&](int i, int cc, int buf) {
    const unsigned off = h_off[i] == 0xFFFFFFFFu ? 0xFFFFFFFFu : h_off[i] + cc * 128;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr)(sHalo + buf * CX_HALO_BYTES + h_grp[i] * 1024), 16, off, 0, 0, 0);
  };
  // weight pieces: quarter jq (global index from the first chunk of this workgroup) -> ring quarter jq & 3; piece i of this wave
  const unsigned wvoff = (unsigned)(lane * 16);
  auto dma_w_piece = [&](int jq, int i) {
    const int fr = min(wave + 4 * i, QF - 1);  // the surplus pieces of the last waves repeat fragment QF - 1 (same bytes, same place)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr)(sRing + (((jq & 3) * QF + fr) << 10)), 16, wvoff + (fr << 10),
                                             (cc_beg * NQ + jq) * (QF << 10), 0, 0);
  };

  // pixel of this lane in each of the wave's two activation fragments.  IW = 16: block row 4 wave + 2 p + (pl >> 4), column
  // pl & 15 of the tile's one block; IW = 8: row 4 p + (pl >> 3), column pl & 7 of block `wave`
  int centre[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if constexpr (IW == 16) centre[p] = (4 * wave + 2 * p + (pl >> 4) + 1) * HWD + (pl & 15) + 1;
    else centre[p] = wave * (HWD * HWD) + (4 * p + (pl >> 3) + 1) * HWD + (pl & 7) + 1;
  }
  // halo-block coordinates of the centre pixel for the XOR key (the row only through its parity, the same for both fragments)
  const int cx = (IW == 16 ? (pl & 15) : (pl & 7)) + 1, cy_par = IW == 16 ? 0 : (((pl >> 3) + 1) & 1);

  f32x16 acc[2][NF];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][f][r] = 0.f;

  if (nchunks > 0) {
    // start: halo of the first chunk, weight quarters 0..2 (quarter j + 3 is issued while quarter j is consumed)
#pragma unroll
    for (int i = 0; i < CX_HP; ++i) dma_halo_piece(i, cc_beg, 0);
#pragma unroll
    for (int jq = 0; jq < 3; ++jq)
#pragma unroll
      for (int i = 0; i < PW; ++i) dma_w_piece(jq, i);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    auto rdw = [&](int slot) -> h8 { return *(const h8*)(sRing + (slot << 10) + lane * 16); };
    auto rdx = [&](int buf, int p, int step) -> h8 {
      const int tap = step >> 2, kk = step & 3;
      const int hrow = centre[p] + (tap / 3 - 1) * HWD + (tap % 3 - 1);
      const int key = halo_key<IW>(cy_par + (tap / 3 - 1) + 2, cx + (tap % 3 - 1));
      return *(const h8*)(sHalo + buf * CX_HALO_BYTES + hrow * 128 + (((2 * kk + h) ^ key) << 4));
    };
    h8 wq[4];     // weight fragments, four ahead
    h8 xf[2][2];  // activation fragments of the current and the next step
#pragma unroll
    for (int k = 0; k < 4; ++k) wq[k] = rdw(k);
    xf[0][0] = rdx(0, 0, 0);
    xf[0][1] = rdx(0, 1, 0);

#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
      const int hbuf = c & 1;
      // the halo of the next chunk; behind the last chunk the same chunk is loaded again (unused): no branch, one wait count
      const int cc_next = min(cc_beg + c + 1, ncc - 1);
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const int qd = s / 3;  // quarter of this chunk
        if (s % 3 == 0 && !(c == 0 && s == 0)) {
          // entering quarter jq = c * NQ + qd: everything issued before the previous quarter has landed once at most the loads of
          // the previous quarter are in flight (PW weight pieces + the halo pieces issued there)
          const int prev = (qd + NQ - 1) % NQ;  // the previous quarter's index within its chunk (halo pieces ride in 2, 3, ...)
          constexpr int NHQ = (CX_HP + 3) / 4, NH_LAST = CX_HP - 4 * (NHQ - 1);  // quarters that carry halo pieces; pieces in the last
          const int nh_prev = (prev >= 2 && prev < 2 + NHQ) ? (prev == 2 + NHQ - 1 ? NH_LAST : 4) : 0;
          if (nh_prev == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PW) : "memory");
          else if (nh_prev == 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PW + 4) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PW + NH_LAST) : "memory");
          __builtin_amdgcn_s_barrier();
        }
        // activation fragments of the next step (the next chunk's first step reads the other halo buffer, landed two quarters ago)
#ifdef CX_EXP_NOLDSX
        if (c < 0)
#endif
        if (s + 1 < NSTEP) {
          xf[(s + 1) & 1][0] = rdx(hbuf, 0, s + 1);
          xf[(s + 1) & 1][1] = rdx(hbuf, 1, s + 1);
        } else {
          xf[(s + 1) & 1][0] = rdx(hbuf ^ 1, 0, 0);
          xf[(s + 1) & 1][1] = rdx(hbuf ^ 1, 1, 0);
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int i = s * NF + f;             // fragment index within the chunk; ring slot (i % (4 QF))
          const int pos = (s % 3) * NF + f;     // position inside its quarter
          // DMA: the first PW fragment slots of a quarter issue the weight pieces of the quarter three ahead; quarters 2..4 of a
          // chunk also carry the next chunk's halo (4 + 4 + rest pieces)
#ifdef CX_EXP_NODMA
          if (c < 0)
#endif
          if (pos < PW) dma_w_piece(c * NQ + qd + 3, pos);
#ifdef CX_EXP_NODMA
          if (c < 0)
#endif
          if (qd >= 2 && qd < 2 + (CX_HP + 3) / 4 && pos >= PW && pos < PW + 4) {
            const int hi = (qd - 2) * 4 + (pos - PW);
            if (hi < CX_HP) dma_halo_piece(hi, cc_next, hbuf ^ 1);
          }
          const h8 a = wq[i & 3];
#ifdef CX_EXP_NOLDSW
          if (c < 0)
#endif
          wq[i & 3] = rdw((i + 4) % (4 * QF));
          acc[0][f] = MVD_MFMA_32x32x16(a, xf[s & 1][0], acc[0][f], 0, 0, 0);
          acc[1][f] = MVD_MFMA_32x32x16(a, xf[s & 1][1], acc[1][f], 0, 0, 0);
#ifndef CX_NO_SCHEDBAR
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's run-ahead loads (slack behind the stream)
    __builtin_amdgcn_s_barrier();                      // everyone is done with the halo / ring: the scratch below aliases them
  }

  // ---- epilogue: each 32 x 32 block (channels x pixels) through a wave-private scratch -> 128-byte row segments
  float* sc = (float*)(smem + wave * EPI_WAVE_BYTES);
  const int M = g.B * H * W;
  float* part = g.splitk > 1 ? g.partial + (long)blockIdx.y * M * N : nullptr;
  const int cq = (lane & 7) * 4;
  // output rows of this lane: fragment p, row (lane >> 3) + 8 i
  long row4[2][4];
  int b = 0;
  bool live = true;
  {
    int y0, x0;
    block_pos(tm * NI + (IW == 16 ? 0 : wave), b, y0, x0);
    live = tm * NI + (IW == 16 ? 0 : wave) < nblocks;  // a tile of 8 x 8 images may end past the batch
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fr = (lane >> 3) + 8 * i;  // pixel of the fragment
        const int yy = IW == 16 ? 4 * wave + 2 * p + (fr >> 4) : 4 * p + (fr >> 3), xx = IW == 16 ? (fr & 15) : (fr & 7);
        row4[p][i] = (long)(b * H + y0 + yy) * W + x0 + xx;
      }
  }
#ifndef CX_NO_RESID_PREFETCH
  // Every residual value of the tile is requested BEFORE the first transpose (2 x NF x 4 loads of 16 bytes per lane, in flight
  // together: one wave per SIMD has the registers): fragment by fragment behind each LDS round trip the epilogue kept four loads
  // in flight per lane and ran at ~4 TB/s with the matrix pipe idle (round 6; CX_NO_RESID_PREFETCH restores that form)
  float4 rpre[2][NF][4];
  const bool has_res = g.resid && !part && live;
  if (has_res) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) rpre[p][f][i] = *(const float4*)((const float*)g.resid + row4[p][i] * g.ldr + n0 + 32 * f + cq);
  }
#endif
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *(float4*)(sc + pl * EPI_LD + 8 * j + 4 * h) =
            make_float4(acc[p][f][4 * j], acc[p][f][4 * j + 1], acc[p][f][4 * j + 2], acc[p][f][4 * j + 3]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int n = n0 + 32 * f + cq;
      float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!part) {
        if (g.bias) bsum = *(const float4*)(g.bias + n);
        if (g.rowbias && live) {  // (a dead wave of the last 8 x 8 tile has b >= B: no read past the per-sample rows)
          const float4 r = *(const float4*)(g.rowbias + (long)b * g.rb_ld + n);
          bsum.x += r.x; bsum.y += r.y; bsum.z += r.z; bsum.w += r.w;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fr = (lane >> 3) + 8 * i;
        float4 v = *(const float4*)(sc + fr * EPI_LD + cq);
        if (!live) continue;
        if (part) {
          *(float4*)(part + row4[p][i] * N + n) = v;
          continue;
        }
        v.x += bsum.x; v.y += bsum.y; v.z += bsum.z; v.w += bsum.w;
        if (g.resid) {
#ifndef CX_NO_RESID_PREFETCH
          const float4 r = rpre[p][f][i];
#else
          const float4 r = *(const float4*)((const float*)g.resid + row4[p][i] * g.ldr + n);
#endif
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        *(float4*)((float*)g.out + row4[p][i] * g.ldc + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch is rewritten by the next block
    }
  }
#endif
}

// packed fp16 weights [9][N][Cin] (ConvW::w) -> per column tile [chunk][tap][k step][row block] fragments, fragment-major
__global__ void conv3x_pack_kernel(const half_t* __restrict__ w, int N, int Cin, int NF, half_t* __restrict__ out, long nfrag) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= nfrag * 64) return;
  const long q = gid >> 6;
  const int l = (int)(gid & 63), h = l >> 5, r32 = l & 31;
  const int ncc = Cin / 64, per_tile = ncc * 36 * NF;
  const int tn = (int)(q / per_tile);
  int r = (int)(q - (long)tn * per_tile);
  const int cc = r / (36 * NF);
  r -= cc * 36 * NF;
  const int step = r / NF, f = r - step * NF;
  const int tap = step >> 2, kk = step & 3;
  const int n = tn * 32 * NF + 32 * f + r32;
  h8 o = *(const h8*)(w + ((long)tap * N + n) * Cin + 64 * cc + 16 * kk + 8 * h);
  *(h8*)(out + gid * 8) = o;
}

template <int NF, int IW>
int launch_cx(const IGemm& g, const half_t* stream, hipStream_t s) {
  constexpr int LDS = 2 * CxGeo<IW>::BYTES + 4 * 3 * NF * 1024;
  static_assert(LDS <= 160 * 1024 && 4 * EPI_WAVE_BYTES <= LDS, "LDS budget");
  static bool attr_done[MVD_MAX_DEVICES] = {false};
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv3x_kernel<NF, IW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  constexpr int NI = CxGeo<IW>::NI;
  dim3 grid(cdiv(g.B * (g.Y / IW) * (g.X / IW), NI) * (g.N / (32 * NF)), g.splitk > 1 ? g.splitk : 1);
  IGemm gl = g;
  static const bool no_cols = getenv("MVD_NO_XCD_COLS") != nullptr;
  const int tiles_m = cdiv(g.B * (g.Y / IW) * (g.X / IW), NI);
  gl.xcd_cols = !no_cols && xcd_prefers_cols(tiles_m, g.N / (32 * NF), (double)tiles_m * CxGeo<IW>::ROWS * g.Cin * 2, 9.0 * g.N * g.Cin * 2);
  hipLaunchKernelGGL((conv3x_kernel<NF, IW>), grid, dim3(256), LDS, s, gl, stream);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// what the kernel takes (see the file comment); bn: 160 or 128 -- the column tile the stream was packed for
bool conv3x_eligible(const IGemm& g, int bn) {
  if (g.a_f32 || g.ntaps != 9 || g.sy != 1 || g.sx != 1 || g.ups || g.Z != 1 || g.geglu || !g.out_linear || g.npar > 0) return false;
  if (g.Cin % 64 || (bn != 160 && bn != 128) || g.N % bn) return false;
  if (g.Y != g.IY || g.X != g.IX) return false;
  if (!((g.Y % 16 == 0 && g.X % 16 == 0) || (g.Y == 8 && g.X == 8))) return false;
  if (!g.out_f32 || g.act != ACT_NONE || g.alpha != 1.0f || g.out_split || g.rowscale || g.gn_partial) return false;
  if ((g.lda & 7) || (g.ldc & 3) || (g.resid && (!g.resid_f32 || (g.ldr & 3))) || (g.rowbias && (g.rb_ld & 3))) return false;
  if ((long)g.B * g.Y * g.X * g.lda * 2 >= 0xFFFFFF00L) return false;  // 32-bit buffer offsets
  return true;
}
size_t conv3x_stream_halfs(int N, int Cin, int bn) {
  // + one ring (4 quarters) of slack per column tile is NOT needed between tiles (the next tile's fragments follow); the
  // last tile gets 4 quarters of slack
  const size_t nf = bn / 32;
  return ((size_t)(N / bn) * (Cin / 64) * 36 * nf + 4 * 3 * nf) * 512;
}
int conv3x_pack(const half_t* w, int N, int Cin, int bn, half_t* stream, hipStream_t s) {
  if (Cin % 64 || (bn != 160 && bn != 128) || N % bn) return mvd_fail("conv3x_pack: unsupported shape");
  const int nf = bn / 32;
  const long nfrag = (long)(N / bn) * (Cin / 64) * 36 * nf;
  HIP_CHECK_RET(hipMemsetAsync(stream + nfrag * 512, 0, (size_t)4 * 3 * nf * 1024, s));
  hipLaunchKernelGGL(conv3x_pack_kernel, dim3((unsigned)((nfrag * 64 + 255) / 256)), dim3(256), 0, s, w, N, Cin, nf, stream, nfrag);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_conv3x(const IGemm& g, const half_t* stream, int bn, hipStream_t s) {
  if (!conv3x_eligible(g, bn)) return mvd_fail("conv3x: shape not supported");
  if (g.splitk > 1 && !g.partial) return mvd_fail("conv3x: split-K without a partial buffer");
  if ((long)(g.N / bn) * (g.Cin / 64) * 36 * (bn / 32) * 1024 >= 0xFFFFFF00L) return mvd_fail("conv3x: weight stream exceeds 4 GiB");
  if (g.X == 8) return bn == 160 ? launch_cx<5, 8>(g, stream, s) : launch_cx<4, 8>(g, stream, s);
  return bn == 160 ? launch_cx<5, 16>(g, stream, s) : launch_cx<4, 16>(g, stream, s);
}
