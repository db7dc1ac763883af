// Row-chain kernel: the row-local tail of a SpatialTransformer block in one launch (layout and scope: rowchain.h).
//
//   t2 = to_out(attn) + b_o + attn2 + t0          (optional, AO)     modules/attention.py:196-200, :266-267
//   x  = LayerNorm3(t2)                                               :268
//   t3 = t2 + FF2(GEGLU(FF1(x)))                                      :37-73, :268
//   y  = proj_out(t3) + b_p + x_in                (optional, PO)      :333-336
//
// Why this shape.  As two GEMM launches FF1 / FF2 at 32 x 32 resolution write and re-read the 32768 x 1280 GEGLU product
// (84 MB each way per block) and the fp32 residual stream between every pair of layers; the dominant GEMM family sat at
// 0.22 of the MFMA roof with 2.1 x its algorithmic bytes.  Every layer above maps a pixel row to a pixel row, so here the
// rows never leave the registers:
//
//   * TRANSPOSED products.  All GEMMs are computed as  out^T[feature][pixel] = W[feature][k] . act^T[k][pixel]:  the weights
//     are the MFMA A operand, the activations the B operand (lane = pixel, 8 consecutive k per lane).  The 32 x 32 x 16
//     MFMA leaves D with lane = pixel and registers = features, i.e. ALREADY in B-operand form for the next layer: eight
//     consecutive accumulator registers, converted to fp16, are one B fragment whose k slots are the features
//     4h + (e & 3) + 8 (e >> 2) of a 16-group (h = lane >> 5).  The weight stream is packed with that permutation of k, so
//     no shuffle, no LDS round trip and no transposition exists between two layers.  LayerNorm, GEGLU, the residual adds
//     are elementwise in that layout (LayerNorm: one cross-half shuffle per row statistic).
//   * ONE wave owns 32 pixel rows for the whole chain and runs alone on its SIMD (4 waves = 128 rows per workgroup, up to
//     512 registers per lane): 10 fp32 accumulator fragments of the residual stream (C = 320), 21 fp16 B fragments of the
//     normalised row, two sets of FF1 accumulators (the GEGLU of unit u is evaluated while the matrix cores run FF1 of
//     unit u + 1).  The 4C hidden activations exist 32 at a time, in registers.
//   * WEIGHTS ONLY through LDS.  The pre-packed stream (rowchain_pack) is a sequence of 1 KiB A fragments in consumption
//     order, fragment-major (lane l reads its 16 bytes at 16 l: conflict-free ds_read_b128, no swizzle arithmetic).  It
//     flows L2 -> LDS by buffer_load ... lds into a ring of four 32-fragment half-bodies; a half-body boundary is
//     {s_waitcnt vmcnt(8); s_barrier; issue the DMA of the half-body three ahead}: two half-bodies (64 KiB) are in
//     flight, the next one has always landed, so the 4-deep register prefetch of fragments never stops at a boundary.
//     All 256 workgroups stream the same 3 MB, which stays in every XCD's L2.
//   * Biases without VALU or VMEM in the loop: FF1's bias (with LayerNorm's shift folded through FF1) and FF2's bias ride
//     as one extra k step against a constant B fragment {1, 1, 0, ...} (weights: bias hi, bias lo in fp16).
//     LayerNorm's gain is folded into FF1's columns.
// HBM traffic per block at 32 x 32, C = 320: 42 MB in (t0 or t2) + 21 MB (attention output) + 42 MB residual + 42 MB out,
// instead of 10 tensors of 21-168 MB.  Inference only (the training step keeps the layered path and its tape).
#include <type_traits>

#include "common.h"
#include "igemm_epilogue.h"
#include "rowchain.h"

namespace {

constexpr int RC_RING_BYTES = 128 * 1024;
#ifndef RC_D
#define RC_D 4
#endif
constexpr int RC_PF = RC_D;  // fragments read ahead of the MFMA that consumes them (register window)
// bodies per iteration of the unit loop: whole ring cycles, alternating accumulator sets, and the register window keeps its phase
__host__ __device__ constexpr int rc_ub(const RcLayout& L) {
  int ub = L.UNR;
  while ((ub * L.BODY_RAW) % RC_PF) ub += L.UNR;
  return ub;
}
constexpr int RC_LDS_BYTES = RC_RING_BYTES + 4 * EPI_WAVE_BYTES;

// position of the (k, h, e) operand slot inside a row of the source matrix (see the file comment)
__host__ __device__ constexpr int rc_perm(int kk, int h, int e) { return 16 * kk + 4 * h + (e & 3) + 8 * (e >> 2); }
// position (relative to its body sequence) of the r-th used fragment of the loop
__host__ __device__ constexpr int rc_loopq(const RcLayout& L, int r) { return (r / L.BODY_RAW) * L.BODY + r % L.BODY_RAW; }
// position of the i-th used fragment of region R (0 prologue, 1 one loop iteration, 2 tail) relative to the region's base;
// indices past the region continue into the next one (the prefetch window crosses region ends)
__host__ __device__ constexpr int rc_qrel(const RcLayout& L, int R, int i) {
  if (R == 0) return i < L.PRO_REAL ? L.PRO_PAD + i : L.PRO + rc_loopq(L, i - L.PRO_REAL);
  if (R == 1) return i < rc_ub(L) * L.BODY_RAW ? rc_loopq(L, i) : rc_ub(L) * L.BODY + (i - rc_ub(L) * L.BODY_RAW);
  return i;
}

template <int C, bool AO, int PO>
__global__ __launch_bounds__(256) void rowchain_kernel(const RowChain p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr RcLayout L = rc_layout(C, AO, PO);
  constexpr int F = L.F, KC = L.KC, K1 = L.K1;
  static_assert(L.BODY == 32 || L.BODY == 64, "body must be one or two half-bodies");
  constexpr int UB = rc_ub(L);
  static_assert(L.NU % UB == 0 && UB % 2 == 0, "whole loop iterations, alternating accumulator sets");
  static_assert((UB * L.BODY_RAW) % RC_PF == 0, "the prefetch window keeps its phase across loop iterations");
  static_assert(L.BODY - L.BODY_RAW + RC_PF < 32 && L.TAIL_REAL >= 2 && L.BODY_RAW % 32 >= 8, "the prefetch window reaches at most one half-body ahead");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128 + wave * 32;  // the wave's first pixel row
  const long pix = m0 + pl;

  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.stream), (short)0, 0xFFFFFFFEu, 0x00020000);
  const unsigned voff = (unsigned)(wave * 8 * 1024 + lane * 16);
  // half-body jg of the stream -> ring quarter s4: this wave's 8 of its 32 fragments, piece i (one 1 KiB DMA instruction)
  auto dma_piece = [&](int jg, int s4, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + ((s4 * 32 + wave * 8 + i) << 10)), 16, voff + (i << 10), jg << 15, 0,
                                             0);
  };
  auto dma_hb = [&](int jg, int s4) {
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(jg, s4, i);
  };
  constexpr int HB0 = L.PRO_PAD / 32;  // first half-body that holds a used fragment
  dma_hb(HB0, HB0 & 3);
  dma_hb(HB0 + 1, (HB0 + 1) & 3);
  dma_hb(HB0 + 2, (HB0 + 2) & 3);

  // ---- the wave's rows: residual stream in the D layout (lane = pixel, register r of block f = channel
  //      32 f + (r & 3) + 8 (r >> 2) + 4 h), B fragments of the current layer's input
  f32x16 acc[F];
  h8 X[K1];
  {
    const float* xr = p.xin + pix * p.ld_x + 4 * h;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = *(const float4*)(xr + 32 * f + 8 * j);
        acc[f][4 * j] = v.x; acc[f][4 * j + 1] = v.y; acc[f][4 * j + 2] = v.z; acc[f][4 * j + 3] = v.w;
      }
    if constexpr (AO) {
      const half_t* ar = p.ao + pix * p.ld_ao + 8 * h;
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) X[kk] = *(const h8*)(ar + 16 * kk);
      const int smp = m0 / p.T;  // T % 32 == 0: the wave's rows belong to one sample
      const bool has_rb = p.rowbias != nullptr;  // a kernel argument: wave-uniform
      const float* rb = has_rb ? p.rowbias + (long)smp * p.rb_ld + 4 * h : p.b_ao;
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 b = *(const float4*)(p.b_ao + 4 * h + 32 * f + 8 * j);
          if (has_rb) {
            const float4 r = *(const float4*)(rb + 32 * f + 8 * j);
            b.x += r.x; b.y += r.y; b.z += r.z; b.w += r.w;
          }
          acc[f][4 * j] += b.x; acc[f][4 * j + 1] += b.y; acc[f][4 * j + 2] += b.z; acc[f][4 * j + 3] += b.w;
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  dma_hb(HB0 + 3, (HB0 + 3) & 3);

  auto rd = [&](int slot) -> h8 { return *(const h8*)(smem + (slot << 10) + lane * 16); };
  h8 w[RC_PF];
#pragma unroll
  for (int k = 0; k < RC_PF; ++k) w[k] = rd(rc_qrel(L, 0, k) & 127);

  int hb_base = 0;  // half-body index of the current region's base
  // one MFMA: the i-th used fragment of region R times B fragment b, into c.  Ring protocol: entering half-body j the wave waits
  // until at most its 8 newest DMA instructions are in flight (half-body j + 1 has landed, j + 2 may still fly) and meets the
  // others; the first 8 steps of half-body j each issue one piece of half-body j + 3 into the quarter j - 1 just left.
  auto step = [&](int R, int i, const h8& b, f32x16& c, const h8* b2 = nullptr) {
    const int q = rc_qrel(L, R, i);
    const bool bnd = i == 0 ? R != 0 : q / 32 != rc_qrel(L, R, i - 1) / 32;
#ifndef RC_EXP_NOBAR
    if (bnd) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#endif
#ifdef RC_EXP_NODMA
    if (R != 1)
#endif
    if ((q & 31) < 8 && !(R == 0 && q / 32 == HB0)) dma_piece(hb_base + q / 32 + 3, (q / 32 + 3) & 3, q & 31);
    const int wi = (i + (R == 0 ? 0 : L.PRO_REAL)) % RC_PF;
    const h8 a = w[wi];
    const int nreal = R == 0 ? L.PRO_REAL : (R == 1 ? UB * L.BODY_RAW : L.TAIL_REAL);
#ifdef RC_EXP_NOLDS
    if (R != 1)
#endif
    if (R != 2 || i + RC_PF < nreal) w[wi] = rd(rc_qrel(L, R, i + RC_PF) & 127);
    c = MVD_MFMA_32x32x16(a, b, c, 0, 0, 0);
    if (b2) c = MVD_MFMA_32x32x16(a, *b2, c, 0, 0, 0);  // the same weight fragment against a second activation fragment
  };

  // ---- to_out projection onto t0 + biases -> t2 (order (k step, row block): consecutive MFMAs hit different accumulators)
  if constexpr (AO) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
      for (int f = 0; f < F; ++f) step(0, kk * F + f, X[kk], acc[f]);
  }

  // ---- LayerNorm3 over the row: lanes l and l ^ 32 hold its two halves.  Gain and shift live in the packed FF1 weights.
  {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[f][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (float)C);
    float v = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[f][r] - mean;
        v += d * d;
      }
    v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / (float)C) + 1e-5f);
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) X[2 * f + g][e] = (half_t)((acc[f][8 * g + e] - mean) * rstd);
#pragma unroll
    for (int e = 0; e < 8; ++e) X[KC][e] = (half_t)((h == 0 && e < 2) ? 1.0f : 0.0f);  // the bias step's activations
  }

  f32x16 hvA, hgA, hvB, hgB;  // FF1 accumulators (value, gate) of two consecutive hidden units
  h8 H0[2], H1[2];            // GEGLU products of two consecutive units as FF2 B fragments
  auto zero = [](f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
  };
  // GEGLU of one unit, cut into items that are issued a few at a time between the MFMAs of the neighbouring units (one wave per
  // SIMD: the matrix pipe only stays busy if VALU work is fed in slices).  value * gelu(gate) with the exact-erf GELU
  // (modules/attention.py:44) in the form  gelu(q) = max(q, 0) - |q| h(|q|),  h(a sqrt 2) = erfc(a) / 2 = (1 + a1 a + ... + a6 a^6)^-16 / 2
  // (Abramowitz & Stegun 7.1.28, |error| <= 3e-7; igemm_epilogue.h): 16 scalar fp32 instructions per element, no compare / select.
  // Items are stage-major (item k = stage k / 16 of element k % 16): neighbours are independent, dependent ones 16 items apart.
  // One wave per SIMD issues everything itself: per MFMA (32 cycles) there are ~8 issue slots of 4 cycles, a transcendental takes
  // four of them.  Scalar fp32 the GEGLU costs 76 cycles per element (rcp form: 15 + 1 transcendental; the exp2 / log2 form trades
  // five plain instructions for one more transcendental: the same 76) -- 1216 of a unit's 1984 MFMA cycles, with the MFMA and
  // LDS issue on top the port is full.  PACKED fp32 (v_pk_*_f32: two elements per 4-cycle slot, |x| as max(x, -x) since the
  // packed forms have no abs modifier) brings it to 48 cycles per element.  Items are stage-major over the 8 element pairs, so
  // dependent packed instructions are 8 items apart (no wait states), and a few of them go out with every MFMA.
  constexpr int NST = 18;
  constexpr int NITEM = NST * 8;
  f32x2 ga[8], gp[8], go[8];
  auto geglu_items = [&](int k0, int k1, const f32x16& hv, const f32x16& hg, h8 (&Ho)[2]) {
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int st = k >> 3, e = k & 7;
      const f32x2 q = f32x2{hg[2 * e], hg[2 * e + 1]};
#ifdef RC_NO_GEGLU
      if (st == NST - 2) go[e] = f32x2{hv[2 * e], hv[2 * e + 1]} + q;
      if (st == NST - 1) {
        Ho[e >> 2][2 * (e & 3)] = (half_t)go[e].x;
        Ho[e >> 2][2 * (e & 3) + 1] = (half_t)go[e].y;
      }
#else
      switch (st) {
        case 0: ga[e] = q * 0.70710678118654752f; break;
        case 1: ga[e] = __builtin_elementwise_max(ga[e], -ga[e]); break;  // |q| / sqrt 2
        case 2: gp[e] = ga[e] * 0.0000430638f + 0.0002765672f; break;
        case 3: gp[e] = gp[e] * ga[e] + 0.0001520143f; break;
        case 4: gp[e] = gp[e] * ga[e] + 0.0092705272f; break;
        case 5: gp[e] = gp[e] * ga[e] + 0.0422820123f; break;
        case 6: gp[e] = gp[e] * ga[e] + 0.0705230784f; break;
        case 7: gp[e] = gp[e] * ga[e] + 1.0f; break;
        case 8: case 9: case 10: case 11: gp[e] = gp[e] * gp[e]; break;
        case 12: gp[e] = f32x2{__builtin_amdgcn_rcpf(gp[e].x), __builtin_amdgcn_rcpf(gp[e].y)}; break;  // 2 h
        case 13: ga[e] = ga[e] * 0.70710678118654752f; break;                                            // |q| / 2
        case 14: go[e] = __builtin_elementwise_max(q, f32x2{0.f, 0.f}); break;
        case 15: go[e] = go[e] - ga[e] * gp[e]; break;  // gelu(q) = max(q, 0) - |q| h
        case 16: go[e] = f32x2{hv[2 * e], hv[2 * e + 1]} * go[e]; break;
        default:
          Ho[e >> 2][2 * (e & 3)] = (half_t)go[e].x;
          Ho[e >> 2][2 * (e & 3) + 1] = (half_t)go[e].y;
          break;
      }
#endif
    }
  };
  // FF1 of one unit (into nv, ng) [+ FF2 of the unit two back, from Hi] with the GEGLU of the unit between them (cv, cg -> Ho)
  auto pipe = [&](int R, int ibase, auto NS, f32x16& nv, f32x16& ng, const f32x16& cv, const f32x16& cg, h8 (&Ho)[2], const h8 (&Hi)[2]) {
    constexpr int nslot = decltype(NS)::value;
    zero(nv);
    zero(ng);
#pragma unroll
    for (int m = 0; m < nslot; ++m) {
      // slot order: (value, gate, FF2) per k step -- three MFMAs apart on any one accumulator (two FF1 chains alone run the matrix
      // pipe at about half rate: a 32 x 32 x 16 MFMA's result is not back within two issue intervals)
      constexpr bool with_g2 = nslot > L.G1;
      const int kk = with_g2 ? (m < 3 * KC ? m / 3 : KC) : m >> 1;
      const int t = with_g2 ? (m < 3 * KC ? m % 3 : m - 3 * KC) : m & 1;
      if (t == 2) step(R, ibase + m, Hi[kk / F], acc[kk % F]);
      else step(R, ibase + m, X[kk], t ? ng : nv);
      geglu_items(m * NITEM / nslot, (m + 1) * NITEM / nslot, cv, cg, Ho);
#ifndef RC_NO_SCHEDBAR
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };

  // FF1 of unit 0, then FF1 of unit 1 beside the GEGLU of unit 0
  zero(hvA);
  zero(hgA);
#pragma unroll
  for (int m = 0; m < L.G1; ++m) step(0, L.AO_N + m, X[m >> 1], (m & 1) ? hgA : hvA);
  pipe(0, L.AO_N + L.G1, std::integral_constant<int, L.G1>{}, hvB, hgB, hvA, hgA, H0, H1);

#ifdef RC_EXP_NOLOOP
  if (p.rows < 0)
#endif
#pragma unroll 1
  for (int it = 0; it < L.NU / UB; ++it) {
    hb_base = L.PRO / 32 + it * (UB * L.BODY / 32);
#pragma unroll
    for (int b = 0; b < UB; ++b) {  // body j = it * UNR + b: FF1 of unit j + 2, GEGLU of unit j + 1, FF2 of unit j
      if (b & 1) pipe(1, b * L.BODY_RAW, std::integral_constant<int, L.BODY_RAW>{}, hvB, hgB, hvA, hgA, H0, H1);
      else pipe(1, b * L.BODY_RAW, std::integral_constant<int, L.BODY_RAW>{}, hvA, hgA, hvB, hgB, H1, H0);
    }
  }

  // ---- tail: FF2 bias (augmented k step), then proj_out
  hb_base = (L.PRO + L.LOOP) / 32;
#pragma unroll
  for (int f = 0; f < F; ++f) step(2, f, X[KC], acc[f]);
  if constexpr (PO == 1) {
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) X[2 * f + g][e] = (half_t)acc[f][8 * g + e];
#pragma unroll
    for (int f = 0; f < F; ++f) zero(acc[f]);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
      for (int f = 0; f < F; ++f) step(2, F + kk * F + f, X[kk], acc[f]);
  }
  if constexpr (PO == 2) {
    // extended precision (ConvW::xp: a_hi w_hi + a_lo w_hi + a_hi w_lo): t3 splits into fp16 hi + lo in registers, every weight
    // position holds a hi and a lo fragment; the hi fragment multiplies both activation parts
    h8 XL[KC];
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const half_t hi = (half_t)acc[f][8 * g + e];
          X[2 * f + g][e] = hi;
          XL[2 * f + g][e] = (half_t)(acc[f][8 * g + e] - (float)hi);
        }
#pragma unroll
    for (int f = 0; f < F; ++f) zero(acc[f]);
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        step(2, F + 2 * (kk * F + f), X[kk], acc[f], &XL[kk]);
        step(2, F + 2 * (kk * F + f) + 1, X[kk], acc[f]);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's run-ahead loads (into the slack of the stream)

  // ---- store: each 32 x 32 block through a wave-private LDS scratch -> full 128-byte row segments
  float* sc = (float*)(smem + RC_RING_BYTES + wave * EPI_WAVE_BYTES);
#ifndef RC_NO_RESID_PREFETCH
  // the block input (the residual of proj_out) for the whole 32 x C tile, requested before the first transpose: F x 4 loads of 16
  // bytes in flight per lane instead of four behind each LDS round trip (as conv3x does since round 6)
  [[maybe_unused]] float4 rres[PO != 0 ? F : 1][4];
  if constexpr (PO != 0) {
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rres[f][i] = *(const float4*)(p.resid + (long)(m0 + (lane >> 3) + 8 * i) * p.ld_r + 32 * f + (lane & 7) * 4);
  }
#endif
#pragma unroll
  for (int f = 0; f < F; ++f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(sc + pl * EPI_LD + 8 * j + 4 * h) = make_float4(acc[f][4 * j], acc[f][4 * j + 1], acc[f][4 * j + 2], acc[f][4 * j + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (PO != 0) {
      const int cq = (lane & 7) * 4, n = 32 * f + cq;
      const float4 bp = *(const float4*)(p.b_po + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fr = (lane >> 3) + 8 * i;
        const long row = m0 + fr;
        float4 v = *(const float4*)(sc + fr * EPI_LD + cq);
#ifndef RC_NO_RESID_PREFETCH
        const float4 r = rres[f][i];
#else
        const float4 r = *(const float4*)(p.resid + row * p.ld_r + n);
#endif
        v.x += bp.x + r.x; v.y += bp.y + r.y; v.z += bp.z + r.z; v.w += bp.w + r.w;
        *(float4*)((float*)p.out + row * p.ld_o + n) = v;
      }
    } else {
      const int c8 = (lane & 3) * 8, n = 32 * f + c8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int fr = (lane >> 2) + 16 * i;
        const long row = m0 + fr;
        const float4 a = *(const float4*)(sc + fr * EPI_LD + c8), b = *(const float4*)(sc + fr * EPI_LD + c8 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        h8 hv;
#pragma unroll
        for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
        half_t* o = (half_t*)p.out + row * p.ld_o + n;
        *(h8*)o = hv;
        if (p.out_split) {
          h8 lo;
#pragma unroll
          for (int e = 0; e < 8; ++e) lo[e] = (half_t)(v[e] - (float)hv[e]);
          *(h8*)(o + p.out_split) = lo;
          *(h8*)(o + 2 * p.out_split) = hv;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the scratch is rewritten by the next block
  }
#endif
}

// ---------------------------------------------------------------------------------------------------- packing
// FF1 bias with LayerNorm's shift folded through: b'[r] = b1[r] + sum_c w1[r][c] * ln_b[c]
__global__ void rowchain_bias_fold_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ ln_b,
                                          int rows, int C, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  double s = b1[r];
  for (int c = 0; c < C; ++c) s += (double)w1[(long)r * C + c] * (double)ln_b[c];
  out[r] = (float)s;
}

// one thread per (fragment, lane): its 8 halfs.  Mirrors the consumption order of rowchain_kernel exactly.
__global__ void rowchain_pack_kernel(const RcWeights w, const RcLayout L, const float* __restrict__ b1f, half_t* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)L.NT_ALLOC * 64) return;
  const int q = (int)(gid >> 6), l = (int)(gid & 63), h = l >> 5, r32 = l & 31;
  const int C = L.C, F = L.F, KC = L.KC;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto bias_pair = [&](float b) {
    if (h == 0) {
      const half_t hi = (half_t)b;
      v[0] = (float)hi;
      v[1] = b - (float)hi;
    }
  };
  auto ff1 = [&](int u, int r) {  // r-th fragment of unit u's FF1 part: (k step, value | gate)
    if (u >= L.NU) return;
    const int kk = r >> 1, gate = r & 1;
    const long row = (gate ? 4 * C : 0) + 32 * u + r32;
    if (kk < KC) {
      for (int e = 0; e < 8; ++e) {
        const int c = rc_perm(kk, h, e);
        v[e] = w.w1[row * C + c] * w.ln_g[c];
      }
    } else {
      bias_pair(b1f[row]);
    }
  };
  if (q < L.PRO) {
    int r = q - L.PRO_PAD;
    if (r >= 0) {
      if (r < L.AO_N) {  // to_out: natural k order (its B operand is loaded from memory)
        const int kk = r / F, f = r % F;
        for (int e = 0; e < 8; ++e) v[e] = w.w_ao[(long)(32 * f + r32) * C + 16 * kk + 8 * h + e];
      } else {
        ff1((r - L.AO_N) / L.G1, (r - L.AO_N) % L.G1);
      }
    }
  } else if (q < L.PRO + L.LOOP) {
    const int u = (q - L.PRO) / L.BODY;
    int r = (q - L.PRO) % L.BODY;
    if (r < 3 * KC) {  // (value, gate, FF2) per k step
      const int kk = r / 3, t = r % 3;
      if (t < 2) {
        ff1(u + 2, 2 * kk + t);
      } else {
        const int g = kk / F, f = kk % F;
        for (int e = 0; e < 8; ++e) v[e] = w.w2[(long)(32 * f + r32) * (4 * C) + 32 * u + rc_perm(g, h, e)];
      }
    } else if (r < L.BODY_RAW) {
      ff1(u + 2, 2 * KC + (r - 3 * KC));
    }
  } else if (q < L.NT) {
    int r = q - L.PRO - L.LOOP;
    if (r < F) {
      bias_pair(w.b2[32 * r + r32]);
    } else if (L.po == 1 && r < F + KC * F) {
      r -= F;
      const int kk = r / F, f = r % F;
      for (int e = 0; e < 8; ++e) v[e] = w.w_po[(long)(32 * f + r32) * C + rc_perm(kk, h, e)];
    } else if (L.po == 2 && r < F + 2 * KC * F) {  // (hi, lo) per (k step, row block)
      r -= F;
      const int lo = r & 1, kk = (r >> 1) / F, f = (r >> 1) % F;
      for (int e = 0; e < 8; ++e) {
        const float x = w.w_po[(long)(32 * f + r32) * C + rc_perm(kk, h, e)];
        v[e] = lo ? x - (float)(half_t)x : x;
      }
    }
  }
  h8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
  *(h8*)(out + gid * 8) = o;
}


// ---------------------------------------------------------------------------------------------------- row head
// proj_in -> t0, LayerNorm1, q | k | v in one launch (rowchain.h).  Same ring, same transposed products; the q|k|v product
// runs in groups of three 32-feature row blocks whose results are converted, transposed through a wave-private scratch and stored
// while the next group multiplies (the stores are issued right behind a half-body boundary: the boundary's vmcnt(8) also
// waits for stores, so they get a whole half-body to be acknowledged).
constexpr int RH_SCR_LD = 104;                           // halfs per scratch row (96 + pad: 208 bytes, 16-byte aligned)
constexpr int RH_SCR_WAVE = 32 * RH_SCR_LD * 2;          // 6656 bytes per wave (>= EPI_WAVE_BYTES: the t0 store uses it too)
constexpr int RH_LDS_BYTES = RC_RING_BYTES + 4 * RH_SCR_WAVE;
static_assert(RH_SCR_WAVE >= EPI_WAVE_BYTES, "scratch");

template <bool XP>
__global__ __launch_bounds__(256) void rowhead_kernel(const RowHead p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int C = RH_C, F = RH_F, KC = RH_KC, K1 = RH_K1;
  constexpr int RH_PI = rh_pi(XP), RH_PRO = rh_pro(XP), RH_PAD = rh_pad(XP);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128 + wave * 32;
  const long pix = m0 + pl;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.stream), (short)0, 0xFFFFFFFEu, 0x00020000);
  const unsigned voff = (unsigned)(wave * 8 * 1024 + lane * 16);
  auto dma_piece = [&](int jg, int s4, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + ((s4 * 32 + wave * 8 + i) << 10)), 16, voff + (i << 10), jg << 15, 0,
                                             0);
  };
  constexpr int HB0 = RH_PAD / 32;
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(HB0 + j, (HB0 + j) & 3, i);

  f32x16 acc[F];
  h8 X[K1];
  [[maybe_unused]] h8 XL[XP ? KC : 1];
  {
    const half_t* ar = p.n0 + pix * p.ld_n0 + 8 * h;
#pragma unroll
    for (int kk = 0; kk < KC; ++kk) X[kk] = *(const h8*)(ar + 16 * kk);
    if constexpr (XP) {  // [hi | lo | hi] rows: the lo part
#pragma unroll
      for (int kk = 0; kk < KC; ++kk) XL[kk] = *(const h8*)(ar + C + 16 * kk);
    }
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = *(const float4*)(p.b_pi + 4 * h + 32 * f + 8 * j);
        acc[f][4 * j] = b.x; acc[f][4 * j + 1] = b.y; acc[f][4 * j + 2] = b.z; acc[f][4 * j + 3] = b.w;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) dma_piece(HB0 + 3, (HB0 + 3) & 3, i);

  auto rd = [&](int slot) -> h8 { return *(const h8*)(smem + (slot << 10) + lane * 16); };
  // stream position of the i-th used fragment of region R (0: proj_in, 1: one loop iteration = two groups)
  auto qrel = [](int R, int i) {
    constexpr int PI = rh_pi(XP), PAD = rh_pad(XP), PRO = rh_pro(XP);
    return R == 0 ? (i < PI ? PAD + i : PRO + (i - PI)) : i;
  };
  h8 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = rd(qrel(0, k) & 127);
  int hb_base = 0;
  // as rowchain_kernel's step; used = false: the position exists in the stream (the 64th of a group) but carries no MFMA
  auto step = [&](int R, int i, const h8& b, f32x16& c, bool used, const h8* b2 = nullptr) {
    const int q = qrel(R, i);
    const bool bnd = i == 0 ? R != 0 : q / 32 != qrel(R, i - 1) / 32;
    if (bnd) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if ((q & 31) < 8 && !(R == 0 && q / 32 == HB0)) dma_piece(hb_base + q / 32 + 3, (q / 32 + 3) & 3, q & 31);
    const int wi = (i + (R == 0 ? 0 : RH_PI)) & 3;
    const h8 a = w[wi];
    w[wi] = rd(qrel(R, i + 4) & 127);
    if (used) c = MVD_MFMA_32x32x16(a, b, c, 0, 0, 0);
    if (b2) c = MVD_MFMA_32x32x16(a, *b2, c, 0, 0, 0);
  };

  // ---- proj_in: t0^T = W n0^T + b (extended precision: w_hi (n_hi + n_lo) + w_lo n_hi)
#pragma unroll
  for (int kk = 0; kk < KC; ++kk)
#pragma unroll
    for (int f = 0; f < F; ++f) {
      if constexpr (XP) {
        step(0, 2 * (kk * F + f), X[kk], acc[f], true, &XL[kk]);
        step(0, 2 * (kk * F + f) + 1, X[kk], acc[f], true);
      } else {
        step(0, kk * F + f, X[kk], acc[f], true);
      }
    }

  // ---- t0 -> memory (the residual of the block's second half): 32 x 32 blocks through the scratch, 128-byte row segments
  float* sc = (float*)(smem + RC_RING_BYTES + wave * RH_SCR_WAVE);
#pragma unroll
  for (int f = 0; f < F; ++f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(sc + pl * EPI_LD + 8 * j + 4 * h) = make_float4(acc[f][4 * j], acc[f][4 * j + 1], acc[f][4 * j + 2], acc[f][4 * j + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int cq = (lane & 7) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int fr = (lane >> 3) + 8 * i;
      *(float4*)(p.t0 + (long)(m0 + fr) * p.ld_t0 + 32 * f + cq) = *(const float4*)(sc + fr * EPI_LD + cq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  // ---- LayerNorm1 (gain and shift live in the packed q|k|v weights)
  {
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[f][r];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / (float)C);
    float v = 0.f;
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[f][r] - mean;
        v += d * d;
      }
    v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / (float)C) + 1e-5f);
#pragma unroll
    for (int f = 0; f < F; ++f)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 8; ++e) X[2 * f + g][e] = (half_t)((acc[f][8 * g + e] - mean) * rstd);
#pragma unroll
    for (int e = 0; e < 8; ++e) X[KC][e] = (half_t)((h == 0 && e < 2) ? 1.0f : 0.0f);
  }

  // ---- q | k | v in ten groups of three row blocks; the result of group g leaves while group g + 1 multiplies
  half_t* hs = (half_t*)(smem + RC_RING_BYTES + wave * RH_SCR_WAVE);  // [32 rows][RH_SCR_LD]
  f32x16 rA[3], rB[3];
  auto zero = [](f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
  };
  // store work of one finished group, in items: 0..11 convert + scratch write of (row block j = k / 4, register quad k % 4), 12..17
  // one 16-byte piece each from the scratch to memory
  auto store_items = [&](int k0, int k1, const f32x16 (&r)[3], int g) {
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      if (k < 12) {
        const int j = k >> 2, qd = k & 3;
        h4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (half_t)r[j][4 * qd + e];
        *(h4*)(hs + pl * RH_SCR_LD + 32 * j + 8 * qd + 4 * h) = v;
      } else if (k < 18) {
        const int id = lane + 64 * (k - 12), row = id / 12, col = id - row * 12;
        *(h8*)(p.qkv + (long)(m0 + row) * p.ld_qkv + 96 * g + 8 * col) = *(const h8*)(hs + row * RH_SCR_LD + 8 * col);
      }
    }
  };
  auto group = [&](int ibase, f32x16 (&cur)[3], const f32x16 (&prev)[3], int gprev, bool has_prev) {
    zero(cur[0]);
    zero(cur[1]);
    zero(cur[2]);
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      if (m < 63) step(1, ibase + m, X[m / 3], cur[m % 3], true);
      else step(1, ibase + m, X[0], cur[0], false);
      // the previous group's 18 store items: conversions and scratch writes over the first slots, the memory stores right
      // behind the next half-body boundary (slot 32 of a group) so that they are acknowledged before the one after it
      if (has_prev) {
        if (m < 12) store_items(m, m + 1, prev, gprev);
        if (m >= 33 && m < 39) store_items(12 + (m - 33), 12 + (m - 33) + 1, prev, gprev);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int it = 0; it < RH_NG / 2; ++it) {
    hb_base = RH_PRO / 32 + it * 4;
    group(0, rA, rB, 2 * it - 1, it > 0);
    group(64, rB, rA, 2 * it, true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_items(0, 18, rB, RH_NG - 1);
#endif
}

__global__ void rowhead_bias_fold_kernel(const float* __restrict__ wq, const float* __restrict__ wk, const float* __restrict__ wv,
                                         const float* __restrict__ ln_b, float* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= 3 * RH_C) return;
  const float* w = (r < RH_C ? wq : (r < 2 * RH_C ? wk : wv)) + (long)(r % RH_C) * RH_C;
  double s = 0.0;
  for (int c = 0; c < RH_C; ++c) s += (double)w[c] * (double)ln_b[c];
  out[r] = (float)s;
}

__global__ void rowhead_pack_kernel(const RhWeights w, const int xp, const float* __restrict__ bf, half_t* __restrict__ out) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int RH_PAD = rh_pad(xp), RH_PRO = rh_pro(xp), RH_NT = rh_nt(xp);
  if (gid >= (long)rh_nt_alloc(xp) * 64) return;
  const int q = (int)(gid >> 6), l = (int)(gid & 63), h = l >> 5, r32 = l & 31;
  constexpr int C = RH_C, F = RH_F, KC = RH_KC;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (q >= RH_PAD && q < RH_PRO) {  // proj_in, natural k order; extended precision: (hi, lo) per (k step, row block)
    const int r0 = q - RH_PAD, lo = xp ? (r0 & 1) : 0, r = xp ? (r0 >> 1) : r0, kk = r / F, f = r % F;
    for (int e = 0; e < 8; ++e) {
      const float x = w.w_pi[(long)(32 * f + r32) * C + 16 * kk + 8 * h + e];
      v[e] = lo ? x - (float)(half_t)x : x;
    }
  } else if (q >= RH_PRO && q < RH_NT) {
    const int g = (q - RH_PRO) / RH_BODY, r = (q - RH_PRO) % RH_BODY;
    if (r < 63) {
      const int kk = r / 3, j = r % 3;
      const int row = 32 * (3 * g + j) + r32;  // row of the stacked q | k | v matrix
      const float* src = (row < C ? w.w_q : (row < 2 * C ? w.w_k : w.w_v)) + (long)(row % C) * C;
      if (kk < KC) {
        for (int e = 0; e < 8; ++e) {
          const int c = rc_perm(kk, h, e);
          v[e] = src[c] * w.ln_g[c];
        }
      } else if (h == 0) {
        const float b = bf[row];
        const half_t hi = (half_t)b;
        v[0] = (float)hi;
        v[1] = b - (float)hi;
      }
    }
  }
  h8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
  *(h8*)(out + gid * 8) = o;
}

template <int C, bool AO, int PO>
int launch_rc(const RowChain& p, hipStream_t s) {
  static bool attr_done[MVD_MAX_DEVICES] = {false};
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)rowchain_kernel<C, AO, PO>, hipFuncAttributeMaxDynamicSharedMemorySize, RC_LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL((rowchain_kernel<C, AO, PO>), dim3(p.rows / 128), dim3(256), RC_LDS_BYTES, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// instantiated forms: the engine uses (to_out, proj_out) = (1, 1) and (1, 2) (extended-precision proj_out); (1, 0) writes x + ff(x);
// (0, 0) is the feed-forward part alone (tests, tools/rowchain_bench.py).  rc_form_instantiated() below lists the same keys: the
// engine packs and plans against it, so a width without the (1, 2) form runs (1, 0) + the separate extended-precision proj_out GEMM.
constexpr bool rc_form_key_ok(int key) {
  return key == 64 * 8 + 5 || key == 64 * 8 + 6 || key == 64 * 8 + 4 || key == 64 * 8 + 0 || key == 128 * 8 + 5 || key == 128 * 8 + 4 ||
         key == 256 * 8 + 5 || key == 256 * 8 + 4 || key == 320 * 8 + 5 || key == 320 * 8 + 6 || key == 320 * 8 + 4 || key == 320 * 8 + 0;
}
int launch_rc_any(const RowChain& p, int C, int ao, int po, hipStream_t s) {
  const int key = C * 8 + (ao ? 4 : 0) + po;
  switch (key) {
    case 64 * 8 + 5: return launch_rc<64, true, 1>(p, s);
    case 64 * 8 + 6: return launch_rc<64, true, 2>(p, s);
    case 64 * 8 + 4: return launch_rc<64, true, 0>(p, s);
    case 64 * 8 + 0: return launch_rc<64, false, 0>(p, s);
    case 128 * 8 + 5: return launch_rc<128, true, 1>(p, s);
    case 128 * 8 + 4: return launch_rc<128, true, 0>(p, s);
    case 256 * 8 + 5: return launch_rc<256, true, 1>(p, s);
    case 256 * 8 + 4: return launch_rc<256, true, 0>(p, s);
    case 320 * 8 + 5: return launch_rc<320, true, 1>(p, s);
    case 320 * 8 + 6: return launch_rc<320, true, 2>(p, s);
    case 320 * 8 + 4: return launch_rc<320, true, 0>(p, s);
    case 320 * 8 + 0: return launch_rc<320, false, 0>(p, s);
    default: return mvd_fail("rowchain: this (width, to_out, proj_out) form is not instantiated");
  }
}

}  // namespace

size_t rowhead_stream_halfs(int xp) { return (size_t)rh_nt_alloc(xp != 0) * 512; }

int rowhead_pack(const RhWeights& w, int xp, float* tmp, half_t* stream, hipStream_t s) {
  hipLaunchKernelGGL(rowhead_bias_fold_kernel, dim3(cdiv(3 * RH_C, 128)), dim3(128), 0, s, w.w_q, w.w_k, w.w_v, w.ln_b, tmp);
  const long n = (long)rh_nt_alloc(xp != 0) * 64;
  hipLaunchKernelGGL(rowhead_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, xp ? 1 : 0, tmp, stream);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <bool XP>
static int launch_rh(const RowHead& p, hipStream_t s) {
  static bool attr_done[MVD_MAX_DEVICES] = {false};
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)rowhead_kernel<XP>, hipFuncAttributeMaxDynamicSharedMemorySize, RH_LDS_BYTES));
    attr_set = true;
  }
  hipLaunchKernelGGL(rowhead_kernel<XP>, dim3(p.rows / 128), dim3(256), RH_LDS_BYTES, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_rowhead(const RowHead& p, int xp, hipStream_t s) {
  if (p.rows <= 0 || p.rows % 128 || (p.ld_n0 & 7) || (p.ld_t0 & 3) || (p.ld_qkv & 7) || (xp && p.ld_n0 < 3 * RH_C))
    return mvd_fail("rowhead: rows % 128, 16-byte aligned row strides ([hi | lo | hi] input rows for the extended-precision form)");
  return xp ? launch_rh<true>(p, s) : launch_rh<false>(p, s);
}

bool rowchain_form_instantiated(int C, int ao, int po) { return rc_form_key_ok(C * 8 + (ao ? 4 : 0) + po); }

bool rowchain_takes(int C, int rows, int T) { return rc_supported_c(C) && rows > 0 && rows % 128 == 0 && T % 32 == 0; }

size_t rowchain_stream_halfs(int C, int ao, int po) { return (size_t)rc_layout(C, ao != 0, po).NT_ALLOC * 512; }

int rowchain_pack(const RcWeights& w, int C, int ao, int po, float* tmp, half_t* stream, hipStream_t s) {
  if (!rc_supported_c(C)) return mvd_fail("rowchain_pack: unsupported width");
  const RcLayout L = rc_layout(C, ao != 0, po);
  hipLaunchKernelGGL(rowchain_bias_fold_kernel, dim3(cdiv(8 * C, 128)), dim3(128), 0, s, w.w1, w.b1, w.ln_b, 8 * C, C, tmp);
  const long n = (long)L.NT_ALLOC * 64;
  hipLaunchKernelGGL(rowchain_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, L, tmp, stream);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_rowchain(const RowChain& p, int C, int ao, int po, hipStream_t s) {
  if (!rowchain_takes(C, p.rows, p.T)) return mvd_fail("rowchain: shape not supported (C in {64,128,256,320}, rows % 128, T % 32)");
  if (po && p.out_split) return mvd_fail("rowchain: the split output belongs to the fp16 form");
  if ((p.ld_x & 3) || (p.ld_o & (po ? 3 : 7)) || (ao && (p.ld_ao & 7)) || (po && (p.ld_r & 3)) || (p.out_split & 7))
    return mvd_fail("rowchain: row strides must keep 16-byte alignment");
  return launch_rc_any(p, C, ao, po, s);
}
