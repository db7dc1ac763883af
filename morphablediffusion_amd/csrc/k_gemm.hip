// LDS-DMA implicit GEMM for every layer whose activation operand is fp16 in HBM: Linear / 1x1 layers and the
// convolutions the halo kernel does not take (strided, upsampled, 3-D, 4x4 images):
//
//   out[row(m)][n] = epilogue( sum_{tap} sum_k A[in(m,tap)][k] * W[tap][n][k] )
//
// A rows are gathered per tap (the per-lane source offsets are recomputed when the tap changes; padding and
// tails are out-of-range buffer offsets -> zeros).
// These layers have short reductions (K = 320 ... 5120, i.e. 5 ... 80 k-steps of 64) and wide outputs, so what
// limits them is not MFMA issue but (a) how far ahead of the MFMAs the operand loads are issued and (b) the
// pipeline fill/drain paid per tile.  This kernel therefore
//   * moves both operands HBM/L2 -> LDS with buffer_load ... lds (no VGPR staging, no ds_write) through a
//     3-stage ring: the loads of TWO k-steps (96-104 KiB per CU) are in flight behind every MFMA block;
//   * uses a 256 x BN tile (8 waves, wave w owns rows [32w, 32w+32) x BN columns): 11.7 B of operand traffic
//     per kFLOP instead of 15.6 for the 128 x 128 tile;
//   * lets one workgroup walk `nch` consecutive column tiles of the same 256 rows: the ring never drains at a
//     tile boundary (the first two k-steps of the next tile are already landing while the epilogue of the
//     previous one runs), and the A rows are re-read from L2;
//   * zero-fills M / N / K tails in hardware (out-of-range buffer offsets), epilogue as in the implicit GEMM
//     (bias, per-sample bias, residual, SiLU, GEGLU pairing, fp16/fp32 store, split-K partials).
//   * keeps what a workgroup executes before its first load and after its last MFMA short (most launches are ONE tile per
//     workgroup, and a kernel's first pass over its code runs at instruction-fetch speed): plain layers take a PLAIN
//     instantiation without integer divisions, biases enter as the accumulators' initial value, fp16 results leave as
//     16-byte stores (measured phase by phase with tools/gemm_timeline.py).
// LDS rows are 128 B (64 halfs) with the 16-byte chunk XOR-swizzled by (row>>1)&7 -> conflict-free ds_read_b128.
// Round 6: the row-tile height is a template parameter.  BM = 256 (eight waves) is the form above; BM = 128 (four waves, one per
// SIMD, 32 rows x BN columns each) was built for the shapes whose 256-row tile grid cannot fill the chip: 8192 x 640 is 160 tiles of
// 256 x 128 on 256 CUs (0.625 of a round, every layer of the 16 x 16 transformer blocks) but exactly 256 tiles of 128 x 160;
// 2048 x 1280 is 160 tiles of 256 x 64 but 224 of 128 x 96.  Same ring, same epilogues, same bits per output element (the
// reduction order over k does not depend on the tile shape).  Measured: it LOSES on every shape of the step (gemm_dma_plan_us) --
// kept as a tested form behind MVD_BM128=1 / MVD_DENSE_BM=128, not planned by default.
#include "common.h"
#include "igemm_epilogue.h"

namespace {

constexpr int GBM = 256, GST = 3;

#ifdef MVD_TIMELINE
// investigation build only (make EXTRA=-DMVD_TIMELINE): per-workgroup phase timestamps of the last launch
__device__ unsigned long long mvd_tl[16 * 4096];
#define TL(i)                                                                                      \
  do {                                                                                             \
    if (tl_rep == MVD_TLREP - 1 && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096) \
      mvd_tl[blockIdx.x * 16 + (i)] = wall_clock64();                                               \
  } while (0)
#ifndef MVD_TLREP
#define MVD_TLREP 1
#endif
#else
#define TL(i)
#endif

[[maybe_unused]] __device__ __forceinline__ int swzg(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int N>
__device__ __forceinline__ void wait_vmg() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// MODE 0: the general kernel.  MODE 1 / 2 (plain row-major GEMM only: one tap, linear rows): the two passes of a
// GroupNorm folded into the GEMM that produces its input (engine_unet: do_cond, context projection).  The workgroup
// walks `nch` consecutive tiles of the row-major tile grid, so long-M / short-K shapes stream A through the ring;
// MODE 1 keeps only (sum, sumsq) of the accumulators per GroupNorm group and 256-row tile (nothing is stored),
// MODE 2 stores relu(acc * rowscale[b][n] + rowbias[b][n]) in fp16.  Both need rows-per-sample % 256 == 0.
// PLAIN (MODE 0 only): one centre tap, unit strides, linear input and output rows (every Linear layer and 1x1 conv): the
// A offsets are m * lda, so the set-up has no integer division and the first loads go out a few hundred cycles after the
// workgroup starts (a kernel's first pass over its code runs at instruction-fetch speed, ~0.7 us per KiB: measured
// with tools/gemm_timeline.py, the general set-up cost 1.8 us per workgroup before the first load was issued).
template <int BM, int BN, int MODE = 0, bool PLAIN = false>
__global__ __launch_bounds__(BM * 2, 1) void gemm_dma_kernel(const IGemm g) {
#if defined(__HIP_DEVICE_COMPILE__)
  [[maybe_unused]] constexpr int GBM = BM, GNT = BM * 2;
  constexpr int NWV = BM / 32;  // waves: each owns 32 rows x BN columns
  constexpr int FN = BN / 32;
  constexpr int A_BYTES = GBM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  constexpr int WGR = BN / 8;                       // 8-row weight groups per stage (one DMA instruction each)
  constexpr int NA = GBM / (8 * NWV);               // A instructions per wave and stage (4)
  constexpr int NW = (WGR + NWV - 1) / NWV, NW_MIN = WGR / NWV;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MVD_TIMELINE
#pragma unroll 1
  for (int tl_rep = 0; tl_rep < MVD_TLREP; ++tl_rep) {
  if (tl_rep) __syncthreads();
#endif
  TL(0);
  const int M = g.B * g.Z * g.Y * g.X, N = g.N, Cin = g.Cin;
  const int tiles_m = (M + GBM - 1) / GBM, tiles_n = (N + BN - 1) / BN;
  constexpr bool WALK = MODE != 0;  // tiles are walked in row-major order across row tiles too
  const int nch = g.nch > 0 ? g.nch : 1;
  const int groups_n = WALK ? 1 : (tiles_n + nch - 1) / nch;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap: consecutive workgroups (same rows, next column group) share an L2
    const int nwg = WALK ? (tiles_m * tiles_n + nch - 1) / nch : tiles_m * groups_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  // MODE 0: fixed row tile tm, column tiles [tn_beg, tn_end).  WALK: linear tile ids [tn_beg, tn_end), id = tm*tiles_n + tn
  int tm = WALK ? 0 : bid / groups_n, gn = WALK ? 0 : bid - tm * groups_n;
  if (!WALK && g.xcd_cols) {  // column-group-major walk: an XCD keeps a few column groups' weights against every row tile
    gn = bid / tiles_m;
    tm = bid - gn * tiles_m;
  }
  int m0 = tm * GBM;
  // PARITY WALK (g.par_walk, general MODE 0 kernel only; round 6): the workgroup owns ONE column tile and walks the npar parity
  // classes of a transposed / upsample convolution one after the other -- the "tile" index of the step stream below is then the
  // parity class, its k range par_ntaps[class] * cpt -- instead of one workgroup per class (blockIdx.z).  The level-0 ConvTranspose3d
  // of the frustum network is 3072 workgroups of 2 ... 16 k-steps that way (282 us at 154 TFLOP/s: fill, drain and a 64 KiB
  // epilogue per handful of steps); walked, the ring never drains between classes.
  const bool PWALK = MODE == 0 && !PLAIN && g.npar > 0 && g.par_walk != 0;
  const int col_tile = gn * nch;  // PWALK: the one column tile of this workgroup
  const int tn_beg = PWALK ? 0 : (WALK ? bid * nch : gn * nch);
  const int tn_end = PWALK ? g.npar : (WALK ? min(tiles_m * tiles_n, tn_beg + nch) : min(tiles_n, tn_beg + nch));

  // parity-batched launch: this workgroup's tap table and output offsets (PWALK: of the class being consumed, see par_of)
  const int par = (g.npar > 0 && !PWALK) ? (int)blockIdx.z : 0;
  const int ntaps = g.npar > 0 ? g.par_ntaps[par] : g.ntaps;
  const int cpt0 = (Cin + 63) / 64;
  const int ksteps_all = ntaps * cpt0;
  int kbeg = 0, kend = ksteps_all;
  if (g.splitk > 1) {
    const int per = (ksteps_all + g.splitk - 1) / g.splitk;
    kbeg = blockIdx.y * per;
    kend = min(ksteps_all, kbeg + per);
  }
  const int ksteps = kend - kbeg;
  // k range end of stream tile `t` (PWALK: of class t)
  auto kend_of = [&](int t) { return PWALK ? g.par_ntaps[t] * cpt0 : kend; };
  int nsteps = ksteps > 0 ? ksteps * (tn_end - tn_beg) : 0;
  if (PWALK) {
    nsteps = 0;
    for (int t = 0; t < g.npar; ++t) nsteps += g.par_ntaps[t] * cpt0;
  }

  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), (short)0, 0xFFFFFFFEu, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.w), (short)0, 0xFFFFFFFEu, 0x00020000);

  // per-lane source offsets: LDS position (row, pos = lane&7) receives source chunk pos ^ ((row>>1)&7)
  unsigned a_off[NA], w_off[NW];
  int a_ch[NA], w_ch[NW], w_row[NW];
  int ab[NA], az[NA], ay[NA], ax[NA];  // output pixel of each A row, pre-multiplied by the stride
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (wave + NWV * i) * 8 + (lane >> 3);
    a_ch[i] = (lane & 7) ^ ((row >> 1) & 7);
    const int m = m0 + row;
    if constexpr (PLAIN) {
      ab[i] = az[i] = ay[i] = ax[i] = 0;
      a_off[i] = (((unsigned)m * (unsigned)g.lda + a_ch[i] * 8) * 2) | (0u - (unsigned)(m >= M));
    } else if (m < M) {
      const int x = m % g.X;
      int t = m / g.X;
      const int y = t % g.Y;
      t /= g.Y;
      ab[i] = t / g.Z;
      az[i] = (t % g.Z) * g.sz;
      ay[i] = y * g.sy;
      ax[i] = x * g.sx;
    } else {
      ab[i] = -1;
      az[i] = ay[i] = ax[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int row = (wave + NWV * i) * 8 + (lane >> 3);
    w_ch[i] = (lane & 7) ^ ((row >> 1) & 7);
    w_row[i] = row;
    w_off[i] = ((unsigned)row * (unsigned)Cin + w_ch[i] * 8) * 2;
  }
  const int cpt = (Cin + 63) / 64;  // k-steps per tap
  [[maybe_unused]] const float inv_rps = 1.0f / (float)(g.Z * g.Y * g.X);
  unsigned w_slab = 0;              // byte offset of the current tap's weight slab
  auto set_tap = [&](int tap, int pr) {
    const int tu = __builtin_amdgcn_readfirstlane(tap), pu = __builtin_amdgcn_readfirstlane(pr);
    const int ti = g.npar > 0 ? g.par_tap[pu][tu] : g.tap[tu];  // one packed dword per tap, scalar load
    const int dz = (ti & 3) - 1, dy = ((ti >> 2) & 3) - 1, dx = ((ti >> 4) & 3) - 1;
    w_slab = (unsigned)(ti >> 8) * (unsigned)N * (unsigned)Cin * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int iz = az[i] + dz, iy = ay[i] + dy, ix = ax[i] + dx;
      const bool ok = ab[i] >= 0 && iz >= 0 && iz < g.IZ && iy >= 0 && iy < g.IY && ix >= 0 && ix < g.IX;
      const unsigned pix = (unsigned)(((ab[i] * g.PZ + (iz >> g.ups)) * g.PY + (iy >> g.ups)) * g.PX + (ix >> g.ups));
      a_off[i] = ((pix * (unsigned)g.lda + a_ch[i] * 8) * 2) | (0u - (unsigned)(!ok));
    }
  };

  // step stream: (column tile, k) with k = tap * cpt + cc; the ring treats the whole walk as one stream
  int set_for = -1;
  int d_kb = 0, d_n0 = 0;  // k offset, first column and weight base of the step being loaded
  unsigned d_wbase = 0;
  auto dma_prepare = [&](int tile, int ks) {
    int tn = tile, tap = 0, cc = ks;
    if constexpr (WALK) {  // linear rows, one tap: the A offsets follow the row tile
      const int tmw = tile / tiles_n;
      tn = tile - tmw * tiles_n;
      if (tmw != set_for) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int m = tmw * GBM + (wave + NWV * i) * 8 + (lane >> 3);
          a_off[i] = (((unsigned)m * (unsigned)g.lda + a_ch[i] * 8) * 2) | (0u - (unsigned)(m >= M));
        }
        set_for = tmw;
      }
    } else if constexpr (!PLAIN) {
      tap = ks / cpt;
      cc = ks - tap * cpt;
      const int pr = PWALK ? tile : par, key = pr * 64 + tap;
      if (PWALK) tn = col_tile;
      if (key != set_for) {  // nothing of this step is in flight yet: the scalar-load wait cannot drain a tile load
        set_tap(tap, pr);
        set_for = key;
      }
    }
    d_kb = cc * 64;
    d_n0 = tn * BN;
    d_wbase = w_slab + (unsigned)d_n0 * (unsigned)Cin * 2 + d_kb * 2;
  };
  // piece p of a stage: p < NA -> 8 A rows, else 8 W rows (one 1 KiB DMA instruction per wave)
  auto dma_piece = [&](int p, int stage) {
    char* sA = smem + stage * STAGE;
    char* sW = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (p == i) {
        const unsigned inval = (0u - (unsigned)(a_off[i] == 0xFFFFFFFFu)) | (0u - (unsigned)(d_kb + a_ch[i] * 8 >= Cin));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr)(sA + (wave + NWV * i) * 1024), 16, (a_off[i] + d_kb * 2) | inval,
                                                 0, 0, 0);
      }
#pragma unroll
    for (int i = 0; i < NW; ++i)
      if (p == NA + i) {
        const int grp = wave + NWV * i;
        if (grp < WGR) {
          const unsigned inval = (0u - (unsigned)(d_n0 + w_row[i] >= N)) | (0u - (unsigned)(d_kb + w_ch[i] * 8 >= Cin));
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr)(sW + grp * 1024), 16, (w_off[i] + d_wbase) | inval, 0, 0, 0);
        }
      }
  };
  auto dma_step = [&](int tile, int ks, int stage) {
    dma_prepare(tile, ks);
#pragma unroll
    for (int p = 0; p < NA + NW; ++p) dma_piece(p, stage);
  };

  f32x16 acc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // Single-tile plain GEMMs start their accumulators at bias[n] + per-sample bias[sample][n] (alpha = 1, the wave's 32-row
  // band inside one sample): in the C layout a lane owns ONE column of each fragment, so that is two loads per fragment,
  // issued before the first tile load (hence retired by the first counted wait) instead of 4 x FN dependent 16-byte loads
  // between the transposes of the epilogue (tools/lin_cold.py: bias + per-sample bias cost the 32768 x 320 x 320 layer
  // 11 us of 37).
  [[maybe_unused]] bool folded = false;
  [[maybe_unused]] float binit[FN], rinit[FN];  // raw loads: summed only after the first tile wait (a use would wait here)
  if constexpr (PLAIN && MODE == 0) {
    const int rps = g.Z * g.Y * g.X;
    folded = !g.geglu && g.splitk <= 1 && nch == 1 && g.alpha == 1.0f && (g.bias || g.rowbias) && tn_beg < tn_end &&
             (!g.rowbias || (rps & 31) == 0);
    if (folded) {
      const int bsu = (int)(((float)min(m0 + wave * 32, M - 1) + 0.5f) * inv_rps);
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = min(tn_beg * BN + fn * 32 + (lane & 31), N - 1);
        binit[fn] = g.bias ? g.bias[n] : 0.f;
        rinit[fn] = g.rowbias ? g.rowbias[(long)bsu * g.rb_ld + n] : 0.f;
      }
    }
  }

  // prologue: steps 0 and 1 in flight, step 0 landed
  int p_tn = tn_beg, p_ks = kbeg;  // (tile, k) of the next step to prefetch
  auto advance = [&](int& tn, int& ks) {
    if (++ks == kend_of(tn)) {
      ks = kbeg;
      ++tn;
    }
  };
  TL(6);
  if (nsteps > 0) {
    dma_step(p_tn, p_ks, 0);
    TL(7);
    advance(p_tn, p_ks);
    if (nsteps > 1) {
      dma_step(p_tn, p_ks, 1);
      advance(p_tn, p_ks);
      wait_vmg<NA + NW_MIN>();
    } else {
      wait_vmg<0>();
    }
  }
  __builtin_amdgcn_s_barrier();
  TL(1);
  if constexpr (PLAIN && MODE == 0) {
    if (folded) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fn][r] = binit[fn] + rinit[fn];
    }
  }

  int tn = tn_beg, ks = kbeg, stage = 0;
  h8 af[2], bf[2][FN];
  for (int s = 0; s < nsteps; ++s) {
    const bool pf = s + 2 < nsteps;  // prefetch of step s+2, issued inside the kk loop
    const char* sA = smem + stage * STAGE;
    const char* sW = sA + A_BYTES;
    auto read_frags = [&](int kk, h8& a, h8 (&b)[FN]) {
      const int ch = kk * 2 + (lane >> 5);
      a = *(const h8*)(sA + swzg(wave * 32 + (lane & 31), ch));
#pragma unroll
      for (int f = 0; f < FN; ++f) b[f] = *(const h8*)(sW + swzg(f * 32 + (lane & 31), ch));
    };
    read_frags(0, af[0], bf[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (s == 0 && kk == 1) TL(8);
      if (s == 0 && kk == 2) TL(9);
      if (s == 1 && kk == 1) TL(12);
      if (s == 1 && kk == 2) TL(13);
      if (kk < 3) read_frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[j] = MVD_MFMA_32x32x16(af[kk & 1], bf[kk & 1][j], acc[j], 0, 0, 0);
      // issue order within the kk block: one LDS read of kk+1 between consecutive MFMAs of kk (the eight waves run in
      // lockstep after the step barrier; a burst of 8 x (FN+1) reads would queue in the LDS while the MFMA pipe idles)
      if (kk < 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef MVD_DMA_SPREAD
      if (pf) {  // experiment: the step's DMA instructions spread over the four kk blocks
        int st2 = stage + 2;
        if (st2 >= GST) st2 -= GST;
        constexpr int NP = NA + NW, PER = (NP + 3) / 4;
        if (kk == 0) dma_prepare(p_tn, p_ks);
#pragma unroll
        for (int p = kk * PER; p < (kk + 1) * PER && p < NP; ++p) dma_piece(p, st2);
        if (kk == 3) advance(p_tn, p_ks);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      if (kk == 1 && pf) {
        // into the ring slot that was read at step s-1 (all waves passed the barrier that ended it); mid-step, so the
        // DMA address set-up does not delay the first fragment reads of the step (see k_conv3.hip)
        int st2 = stage + 2;
        if (st2 >= GST) st2 -= GST;
        dma_step(p_tn, p_ks, st2);
        advance(p_tn, p_ks);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    // step s+1 must have landed before anyone reads it; only this step's own prefetch may stay in flight
    if (s == 0) TL(10);
    if (s == 1) TL(14);
    if (pf) wait_vmg<NA + NW_MIN>();
    else wait_vmg<0>();
    if (s == 0) TL(11);
    if (s == 1) TL(15);
    __builtin_amdgcn_s_barrier();

    const bool tile_done = ks + 1 == kend_of(tn);
    if (s == 0) TL(2);
    if (s == 1) TL(3);
    if (tile_done) {
      TL(4);
      // ---- epilogue of column tile tn; scratch = the ring slot just consumed (free until the next prefetch) ----
      int n0 = (PWALK ? col_tile : tn) * BN;
      // output offsets of the class this tile belongs to (PWALK: the class just finished)
      const int opar = PWALK ? tn : par;
      const int ozo = g.npar > 0 ? g.par_oz[opar] : g.ozo, oyo = g.npar > 0 ? g.par_oy[opar] : g.oyo,
                oxo = g.npar > 0 ? g.par_ox[opar] : g.oxo;
      if constexpr (WALK) {
        const int tmw = tn / tiles_n;
        m0 = tmw * GBM;
        n0 = (tn - tmw * tiles_n) * BN;
      }
      float* scratch = (float*)(smem + stage * STAGE + wave * EPI_WAVE_BYTES);
      int rows4[4], bs4[4];
      long orow4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wave * 32 + (lane >> 3) + 8 * i;
        rows4[i] = m < M ? m : -1;
        if constexpr (PLAIN) {
          orow4[i] = m < M ? m : 0;
          // sample of row m: one float multiply (exact for m < 2^22; launch_gemm_dma checks), not four integer divisions
          bs4[i] = g.rowbias ? (int)(((float)m + 0.5f) * inv_rps) : 0;
        } else {
          orow4[i] = m < M ? out_row_off(g, m, ozo, oyo, oxo) : 0;
          bs4[i] = -1;
        }
      }
      if constexpr (MODE == 1) {
        // statistics-only pass: (sum, sumsq) of the accumulators per GroupNorm group of this 256-row tile; nothing is
        // stored.  Rows past M and columns past N are zero (hardware zero fill), so they do not disturb the sums.
        const int cpg = g.gn_cpg, ng = BN / cpg, ngroups = N / cpg;
        float* red = (float*)(smem + stage * STAGE);  // [8 waves][ng][2]
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          float sx = 0.f, sq = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[fn][r] * g.alpha;
            sx += v;
            sq += v * v;
          }
          sx += __shfl_xor(sx, 32);  // the two half-waves hold the other 16 rows of the same column
          sq += __shfl_xor(sq, 32);
          for (int o = 1; o < cpg; o <<= 1) {  // columns of one group are cpg adjacent lanes
            sx += __shfl_xor(sx, o);
            sq += __shfl_xor(sq, o);
          }
          if (lane < 32 && (lane & (cpg - 1)) == 0) {
            const int gi = (fn * 32 + lane) / cpg;
            red[(wave * ng + gi) * 2] = sx;
            red[(wave * ng + gi) * 2 + 1] = sq;
          }
        }
        __syncthreads();
        if (tid < ng && n0 / cpg + tid < ngroups) {
          float sx = 0.f, sq = 0.f;
          for (int w = 0; w < GNT / 64; ++w) {
            sx += red[(w * ng + tid) * 2];
            sq += red[(w * ng + tid) * 2 + 1];
          }
          float* p = g.gn_partial + ((long)(m0 / GBM) * ngroups + n0 / cpg + tid) * 2;
          p[0] = sx;
          p[1] = sq;
        }
      } else if constexpr (MODE == 2) {
        // relu(acc * scale[b][n] + shift[b][n]) -> fp16; the tile lies inside one sample (rows-per-sample % 256 == 0)
        const int bsmp = m0 / (g.Z * g.Y * g.X), col = lane & 31;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int n = n0 + fn * 32 + col;
          const bool okc = n < N;
          const float sc = okc ? g.rowscale[(long)bsmp * g.rs_ld + n] : 0.f, sh = okc ? g.rowbias[(long)bsmp * g.rb_ld + n] : 0.f;
          f32x16 v;
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = fmaxf(acc[fn][r] * sc + sh, 0.f);
          epilogue_frag_store_raw(g, v, scratch, lane, rows4, orow4, n0 + fn * 32, N);
        }
      } else if (g.geglu) {
        if constexpr (FN % 2 == 0) {
          // value / gate fragments share the C layout: pair them in registers, then one transposed store
          const int col = lane & 31;
#pragma unroll
          for (int p = 0; p < FN / 2; ++p) {
            const int nx = n0 + p * 64 + col;
            const bool okc = nx + 32 < N;
            const float bx = (g.bias && okc) ? g.bias[nx] : 0.f, bg = (g.bias && okc) ? g.bias[nx + 32] : 0.f;
            f32x16 v;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {  // packed fp32: two rows per instruction
              const f32x2 xv = f32x2{acc[2 * p][r], acc[2 * p][r + 1]} * g.alpha + bx;
              const f32x2 qv = f32x2{acc[2 * p + 1][r], acc[2 * p + 1][r + 1]} * g.alpha + bg;
              const f32x2 o = geglu_pair(xv, qv);
              v[r] = o.x;
              v[r + 1] = o.y;
            }
            if (PLAIN && epilogue8_ok(g, N >> 1))
              epilogue8_frag_store<true>(g, v, scratch, lane, m0 + wave * 32, M, (n0 >> 1) + p * 32, N >> 1, nullptr);
            else
              epilogue_frag_store_raw(g, v, scratch, lane, rows4, orow4, (n0 >> 1) + p * 32, N >> 1);
          }
        }
      } else {
        float* part = g.splitk > 1 ? g.partial + (long)blockIdx.y * M * N : nullptr;
        if (PLAIN && !part && epilogue8_ok(g, N)) {
          float pre[FN][2][8];
#pragma unroll
          for (int fn = 0; fn < FN; ++fn)
            epilogue8_prefetch(g, lane, m0 + wave * 32, M, n0 + fn * 32, inv_rps, pre[fn], folded);
#pragma unroll
          for (int fn = 0; fn < FN; ++fn)
            epilogue8_frag_store<false>(g, acc[fn], scratch, lane, m0 + wave * 32, M, n0 + fn * 32, N, pre[fn]);
        } else {  // (fp32 results: 16-byte stores already; prefetching their residual reads measured no gain, warm or cold)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn)
            epilogue_frag_store(g, acc[fn], scratch, lane, rows4, orow4, n0 + fn * 32, part, PLAIN ? bs4 : nullptr, folded);
        }
      }
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
      if (s + 1 < nsteps) __builtin_amdgcn_s_barrier();  // scratch is the next prefetch target
    }
    advance(tn, ks);
    if (++stage == GST) stage = 0;
  }
  TL(5);
#ifdef MVD_TIMELINE
  }
#endif
#endif
}

template <int BM, int BN, int MODE = 0, bool PLAIN = false>
int launch_gd(const IGemm& g, int M, hipStream_t s) {
  constexpr int LDS = GST * (BM * 128 + BN * 128);
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(BM == 256 || (BM == 128 && MODE == 0), "the 128-row form exists for the general / plain GEMM only");
  static bool attr_done[MVD_MAX_DEVICES] = {false};  // the attribute is per device
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(
        hipFuncSetAttribute((const void*)gemm_dma_kernel<BM, BN, MODE, PLAIN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int nch = g.nch > 0 ? g.nch : 1;
  const int gx = MODE ? cdiv(cdiv(M, BM) * cdiv(g.N, BN), nch) : cdiv(M, BM) * cdiv(cdiv(g.N, BN), nch);
  const bool pwalk = MODE == 0 && !PLAIN && g.npar > 0 && g.par_walk != 0;
  dim3 grid(gx, g.splitk > 1 ? g.splitk : 1, (g.npar > 0 && !pwalk) ? g.npar : 1);
  IGemm gl = g;
  gl.xcd_cols = 0;
  if (MODE == 0) {
    static const bool no_cols = getenv("MVD_NO_XCD_COLS") != nullptr;
    const int ntaps = g.npar > 0 ? g.par_ntaps[0] : g.ntaps;
    gl.xcd_cols = !no_cols && xcd_prefers_cols(cdiv(M, BM), cdiv(cdiv(g.N, BN), nch),
                                               (double)g.B * g.PZ * g.PY * g.PX * g.Cin * (g.a_f32 ? 4 : 2), (double)ntaps * g.N * g.Cin * 2);
  }
  hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, MODE, PLAIN>), grid, dim3(BM * 2), LDS, s, gl);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

#ifdef MVD_TIMELINE
extern "C" int mvd_debug_timeline(unsigned long long* host_out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mvd_tl), (size_t)n * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(mvd_tl)) != hipSuccess) return -1;
  return hipMemset(p, 0, sizeof(mvd_tl)) == hipSuccess ? 0 : -1;
}
#endif

// eligibility: fp16 activations with 16-byte aligned rows and a vectorisable epilogue
bool gemm_dma_eligible(const IGemm& g) {
  if (g.a_f32) return false;
  if ((g.lda & 7) || (g.Cin & 7) || ((uintptr_t)g.a & 15) || ((uintptr_t)g.w & 15)) return false;
  if ((g.N & 3) || (g.ldc & 3)) return false;
  if (g.geglu) return (g.N & 63) == 0;
  if (g.resid && (g.ldr & 3)) return false;
  if (g.rowbias && (g.rb_ld & 3)) return false;
  return true;
}

// Tile walk: `nch` column tiles per workgroup, or split-K when the tile count cannot fill the chip.
// Cost model in k-steps: waves * (steps per workgroup + ~4 for fill and epilogue).
void gemm_dma_plan(int M, int N, int ksteps, int bn, int geglu, int* nch_out, int* splitk_out) {
  const int tiles_m = cdiv(M, GBM), tiles_n = cdiv(N, bn);
  const int CUS = 256;
  int best_nch = 1, best_sk = 1;
  long best = -1;
  for (int nch = 1; nch <= tiles_n; ++nch) {
    const int blocks = tiles_m * cdiv(tiles_n, nch);
    const long cost = (long)cdiv(blocks, CUS) * (nch * ksteps + 4 + 2 * nch);
    if (best < 0 || cost < best) {
      best = cost;
      best_nch = nch;
    }
  }
  if (!geglu && tiles_m * tiles_n < CUS) {
    for (int sk = 2; sk <= 16 && sk * 2 <= ksteps; ++sk) {
      const int blocks = tiles_m * tiles_n * sk;
      const long cost = (long)cdiv(blocks, CUS) * (cdiv(ksteps, sk) + 6) + 1;  // + the reduce pass (swept 0..20: 0-1 best)
      if (cost < best) {
        best = cost;
        best_nch = 1;
        best_sk = sk;
      }
    }
  }
  *nch_out = best_nch;
  *splitk_out = best_sk;
}

// Column-tile width, column walk and split-K of a plain dense launch from a time model in microseconds, fitted to
// tools/gemm_timeline.py / tools/gemm_plan_sweep.py (MI355X, one 256-row workgroup per CU):
//   workgroup = 2.8 (first loads + the cold first k-step) + k-steps * t_step(bn) + tiles * t_epi,
//   t_epi     = the round's result bytes at ~6 TB/s (every workgroup of a round stores at the same time), >= 2.5,
//   split-K   = fp32 slabs written by the GEMM and read back by the reduce launch (its own ~5 us of launch and latency).
// The k-step model above undervalued the slab traffic: e.g. 8192 x 2560 x 640 ran 63 us as 3 splits and 49 us unsplit.
// out_b / res_b: bytes per result element stored and per residual element read (GEGLU: per paired column).
// bm_out (optional): also consider 128-row tiles (four waves) with 96 / 128 / 160 columns -- for the shapes whose 256-row tile grid
// leaves CUs idle.  k-step times of the 128-row forms fitted to tools/gemm_plan_sweep.py on the same box (round 6).
void gemm_dma_plan_us(int M, int N, int ksteps, int geglu, int out_b, int res_b, int* bn_io, int* nch_out, int* splitk_out, int* bm_out) {
  const int CUS = 256;
  double best = 1e30;
  int best_bn = *bn_io, best_nch = 1, best_sk = 1, best_bm = 256;
  // MEASURED, and it lost (tools/gemm_bm_sweep.py, profiles/r06_b_bm_sweep.txt): a k-step of the four-wave tile takes 0.70 / 0.79 /
  // 1.03 us at 96 / 128 / 160 columns -- as long as the eight-wave tile of twice the rows (0.70 / 1.05 / 1.25 at 64 / 128 / 160): one
  // wave per SIMD pays every DMA issue, LDS wait and barrier of the step itself, so what a k-step costs is its dependent chain, not
  // its MFMAs.  8192 x 640 x 640: 27.0 us as 256 tiles of 128 x 160 against 24.0 as 160 tiles of 256 x 128; no shape of the step won.
  // The planner therefore only offers these tiles when MVD_BM128=1 (A/B reproduction); with the measured step times it would not pick them.
  static const bool no_bm128 = getenv("MVD_BM128") == nullptr || getenv("MVD_NO_BM128") != nullptr;
  struct Cand {
    int bm, bn;
    double t_step;
  };
  const Cand cand[6] = {{256, 64, 0.70}, {256, 128, 1.05}, {256, 160, 1.25}, {128, 96, 0.70}, {128, 128, 0.79}, {128, 160, 1.03}};
  for (int ci = 0; ci < 6; ++ci) {
    const int bm = cand[ci].bm, bn = cand[ci].bn;
    if (bm == 128 && (!bm_out || no_bm128 || geglu)) continue;
    if (bm == 256 && (geglu ? bn != 128 : (*bn_io < 0 && bn != -*bn_io))) continue;  // bn_io < 0: the caller fixes the width
    if (bm == 128 && *bn_io < 0) continue;
    const double t_step = cand[ci].t_step;
    const int tiles_m = cdiv(M, bm), tiles_n = cdiv(N, bn);
    const double tile_cols = geglu ? bn / 2 : bn;
    for (int nch = 1; nch <= tiles_n; ++nch) {
      const int wgs = tiles_m * cdiv(tiles_n, nch), rounds = cdiv(wgs, CUS);
      const int in_round = wgs < CUS ? wgs : CUS;
      double t_epi = in_round * (double)bm * tile_cols * (out_b + res_b) / 6e6;
      if (t_epi < 2.5) t_epi = 2.5;
      const double t = rounds * (2.8 + (double)nch * ksteps * t_step + nch * t_epi) + 2.0;
      if (t < best) {
        best = t;
        best_bn = bn;
        best_bm = bm;
        best_nch = nch;
        best_sk = 1;
      }
    }
    if (geglu) continue;
    // Reductions of more than 512 k-steps (K > 32768: only the weight-gradient GEMMs of a training step, whose K is the row
    // count of a layer) may split further than 16 ways: a [128 x 64] result over K = 1.18 M ran 690 us as 16 workgroups.
    const int sk_max = ksteps > 512 ? 256 : 16;
    for (int sk = 2; sk <= sk_max && sk * 2 <= ksteps; sk += (sk < 16 ? 1 : (sk < 64 ? 4 : 16))) {
      const int per = cdiv(ksteps, sk), sk_eff = cdiv(ksteps, per);  // no empty split
      if (sk_eff != sk && sk <= 16) continue;
      const int wgs = tiles_m * tiles_n * sk_eff, rounds = cdiv(wgs, CUS);
      const int in_round = wgs < CUS ? wgs : CUS;
      double t_epi = in_round * (double)bm * bn * 4.0 / 6e6;
      if (t_epi < 2.5) t_epi = 2.5;
      const double mn = (double)M * N;
      const double t = rounds * (2.8 + per * t_step + t_epi) + 2.0 + 5.0 + (sk_eff * mn * 4.0 + mn * (out_b + res_b)) / 6e6;
      if (t < best) {
        best = t;
        best_bn = bn;
        best_bm = bm;
        best_nch = 1;
        best_sk = sk_eff;
      }
    }
  }
  *bn_io = best_bn;
  *nch_out = best_nch;
  *splitk_out = best_sk;
  if (bm_out) *bm_out = best_bm;
}

// one centre tap, unit strides, linear input and output rows: the PLAIN instantiations (every Linear layer and 1 x 1 conv)
bool gemm_dma_is_plain(const IGemm& g) {
  static const bool no_plain = getenv("MVD_NO_PLAIN") != nullptr;
  const long M = (long)g.B * g.Z * g.Y * g.X;
  return !no_plain && g.npar == 0 && g.ntaps == 1 && g.tap[0] == igemm_tap(0, 0, 0, 0) && g.out_linear && g.ups == 0 && g.sz == 1 &&
         g.sy == 1 && g.sx == 1 && g.PZ == g.Z && g.PY == g.Y && g.PX == g.X && g.IZ == g.Z && g.IY == g.Y && g.IX == g.X &&
         M < (1 << 22) && !g.gn_partial && !g.rowscale;
}

int launch_gemm_dma(const IGemm& g, hipStream_t s) {
  const int M = g.B * g.Z * g.Y * g.X;
  if (M <= 0 || g.N <= 0) return 0;
  if (g.splitk > 1 && !g.partial) return mvd_fail("gemm_dma: split-K without a partial buffer");
  if (g.npar > 0 && (g.npar > 8 || g.splitk > 1)) return mvd_fail("gemm_dma: parity batch needs npar <= 8 and no split-K");
  // 32-bit buffer offsets: the weight operand spans (highest slab index used + 1) slabs of N x Cin halfs
  int slabs = 1;
  if (g.npar > 0) {
    for (int p = 0; p < g.npar; ++p)
      for (int t = 0; t < g.par_ntaps[p]; ++t) slabs = (g.par_tap[p][t] >> 8) + 1 > slabs ? (g.par_tap[p][t] >> 8) + 1 : slabs;
  } else {
    for (int t = 0; t < g.ntaps; ++t) slabs = (g.tap[t] >> 8) + 1 > slabs ? (g.tap[t] >> 8) + 1 : slabs;
  }
  if ((long)g.B * g.PZ * g.PY * g.PX * g.lda * 2 >= 0xFFFFFF00L || (long)slabs * g.N * g.Cin * 2 >= 0xFFFFFF00L)
    return mvd_fail("gemm_dma: operand exceeds 4 GiB buffer addressing");
  if (g.gn_partial || g.rowscale) {  // folded-GroupNorm passes: plain GEMM, 64 / 128 wide tiles
    if (g.ntaps != 1 || g.splitk > 1 || g.npar > 0 || g.geglu || !g.out_linear || (g.bn != 64 && g.bn != 128))
      return mvd_fail("gemm_dma: the folded-GroupNorm passes need a plain GEMM");
    if (g.bm == 128) return mvd_fail("gemm_dma: the folded-GroupNorm passes use 256-row tiles");
    if (g.gn_partial) return g.bn == 64 ? launch_gd<256, 64, 1>(g, M, s) : launch_gd<256, 128, 1>(g, M, s);
    if (!g.rowbias || g.out_f32) return mvd_fail("gemm_dma: the apply pass needs scale, shift and an fp16 output");
    return g.bn == 64 ? launch_gd<256, 64, 2>(g, M, s) : launch_gd<256, 128, 2>(g, M, s);
  }
  // split-K: the caller (igemm_go) runs launch_splitk_reduce
  const bool plain = gemm_dma_is_plain(g);
  if (g.bm == 128) {  // four-wave row tiles (the plan picks them for plain dense layers only)
    if (!plain || g.geglu) return mvd_fail("gemm_dma: 128-row tiles are for plain (one tap, linear rows, no GEGLU) GEMMs");
    if (g.bn == 96) return launch_gd<128, 96, 0, true>(g, M, s);
    if (g.bn == 128) return launch_gd<128, 128, 0, true>(g, M, s);
    if (g.bn == 160) return launch_gd<128, 160, 0, true>(g, M, s);
    return mvd_fail("gemm_dma: 128-row tiles come 96, 128 or 160 columns wide");
  }
  if (plain) return g.bn == 160 ? launch_gd<256, 160, 0, true>(g, M, s) : (g.bn == 64 ? launch_gd<256, 64, 0, true>(g, M, s) : launch_gd<256, 128, 0, true>(g, M, s));
  return g.bn == 160 ? launch_gd<256, 160>(g, M, s) : (g.bn == 64 ? launch_gd<256, 64>(g, M, s) : launch_gd<256, 128>(g, M, s));
}
