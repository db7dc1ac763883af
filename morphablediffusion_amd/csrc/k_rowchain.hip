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
#include "common.h"
#include "igemm_epilogue.h"
#include "rowchain.h"

namespace {

constexpr int RC_RING_BYTES = 128 * 1024;
constexpr int RC_LDS_BYTES = RC_RING_BYTES + 4 * EPI_WAVE_BYTES;

// position of the (k, h, e) operand slot inside a row of the source matrix (see the file comment)
__host__ __device__ constexpr int rc_perm(int kk, int h, int e) { return 16 * kk + 4 * h + (e & 3) + 8 * (e >> 2); }
// position (relative to its body sequence) of the r-th used fragment of the loop
__host__ __device__ constexpr int rc_loopq(const RcLayout& L, int r) { return (r / L.BODY_RAW) * L.BODY + r % L.BODY_RAW; }
// position of the i-th used fragment of region R (0 prologue, 1 one loop iteration, 2 tail) relative to the region's base;
// indices past the region continue into the next one (the prefetch window crosses region ends)
__host__ __device__ constexpr int rc_qrel(const RcLayout& L, int R, int i) {
  if (R == 0) return i < L.PRO_REAL ? L.PRO_PAD + i : L.PRO + rc_loopq(L, i - L.PRO_REAL);
  if (R == 1) return i < L.UNR * L.BODY_RAW ? rc_loopq(L, i) : 128 + (i - L.UNR * L.BODY_RAW);
  return i;
}

template <int C, bool AO, bool PO>
__global__ __launch_bounds__(256) void rowchain_kernel(const RowChain p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr RcLayout L = rc_layout(C, AO, PO);
  constexpr int F = L.F, KC = L.KC, K1 = L.K1;
  static_assert(L.BODY == 32 || L.BODY == 64, "body must be one or two half-bodies");
  static_assert(L.NU % L.UNR == 0 && L.UNR % 2 == 0, "whole loop iterations, alternating accumulator sets");
  static_assert((L.UNR * L.BODY_RAW) % 4 == 0, "the prefetch window keeps its phase across loop iterations");
  static_assert(L.BODY - L.BODY_RAW + 4 < 32 && L.TAIL_REAL >= 2, "the prefetch window reaches at most one half-body ahead");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, pl = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * 128 + wave * 32;  // the wave's first pixel row
  const long pix = m0 + pl;

  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(p.stream), (short)0, 0xFFFFFFFEu, 0x00020000);
  const unsigned voff = (unsigned)(wave * 8 * 1024 + lane * 16);
  // half-body jg of the stream -> ring quarter s4: this wave's 8 of its 32 fragments
  auto dma_hb = [&](int jg, int s4) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(smem + ((s4 * 32 + wave * 8 + i) << 10)), 16, voff + (i << 10),
                                               jg << 15, 0, 0);
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
  h8 w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = rd(rc_qrel(L, 0, k) & 127);

  int hb_base = 0;  // half-body index of the current region's base
  // one MFMA: the i-th used fragment of region R times B fragment b, into c
  auto step = [&](int R, int i, const h8& b, f32x16& c) {
    const int q = rc_qrel(L, R, i);
    const bool bnd = i == 0 ? R != 0 : q / 32 != rc_qrel(L, R, i - 1) / 32;
    if (bnd) {  // entering half-body q / 32: the one after it has landed once at most this wave's last 8 loads are in flight
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      dma_hb(hb_base + q / 32 + 3, (q / 32 + 3) & 3);
    }
    const int wi = (i + (R == 0 ? 0 : L.PRO_REAL)) & 3;
    const h8 a = w[wi];
    const int nreal = R == 0 ? L.PRO_REAL : (R == 1 ? L.UNR * L.BODY_RAW : L.TAIL_REAL);
    if (R != 2 || i + 4 < nreal) w[wi] = rd(rc_qrel(L, R, i + 4) & 127);
    c = MVD_MFMA_32x32x16(a, b, c, 0, 0, 0);
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
  auto zero = [](f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
  };
  auto gemm1 = [&](int R, int ibase, f32x16& hv, f32x16& hg) {
    zero(hv);
    zero(hg);
#pragma unroll
    for (int kk = 0; kk < K1; ++kk) {
      step(R, ibase + 2 * kk, X[kk], hv);
      step(R, ibase + 2 * kk + 1, X[kk], hg);
    }
  };
  h8 H[2];
  auto geglu = [&](const f32x16& hv, const f32x16& hg) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2 o = geglu_pair(f32x2{hv[r], hv[r + 1]}, f32x2{hg[r], hg[r + 1]});
      H[r >> 3][r & 7] = (half_t)o.x;
      H[r >> 3][(r & 7) + 1] = (half_t)o.y;
    }
  };
  auto gemm2 = [&](int R, int ibase) {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int f = 0; f < F; ++f) step(R, ibase + g * F + f, H[g], acc[f]);
  };

  gemm1(0, L.AO_N, hvA, hgA);  // FF1 of unit 0

#pragma unroll 1
  for (int it = 0; it < L.NU / L.UNR; ++it) {
    hb_base = L.PRO / 32 + it * 4;
#pragma unroll
    for (int b = 0; b < L.UNR; ++b) {  // unit u = it * UNR + b: FF1 of unit u + 1 beside the GEGLU of unit u, then FF2 of unit u
      if (b & 1) {
        gemm1(1, b * L.BODY_RAW, hvA, hgA);
        geglu(hvB, hgB);
      } else {
        gemm1(1, b * L.BODY_RAW, hvB, hgB);
        geglu(hvA, hgA);
      }
      gemm2(1, b * L.BODY_RAW + L.G1);
    }
  }

  // ---- tail: FF2 bias (augmented k step), then proj_out
  hb_base = (L.PRO + L.LOOP) / 32;
#pragma unroll
  for (int f = 0; f < F; ++f) step(2, f, X[KC], acc[f]);
  if constexpr (PO) {
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the ring's run-ahead loads (into the slack of the stream)

  // ---- store: each 32 x 32 block through a wave-private LDS scratch -> full 128-byte row segments
  float* sc = (float*)(smem + RC_RING_BYTES + wave * EPI_WAVE_BYTES);
#pragma unroll
  for (int f = 0; f < F; ++f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *(float4*)(sc + pl * EPI_LD + 8 * j + 4 * h) = make_float4(acc[f][4 * j], acc[f][4 * j + 1], acc[f][4 * j + 2], acc[f][4 * j + 3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (PO) {
      const int cq = (lane & 7) * 4, n = 32 * f + cq;
      const float4 bp = *(const float4*)(p.b_po + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int fr = (lane >> 3) + 8 * i;
        const long row = m0 + fr;
        float4 v = *(const float4*)(sc + fr * EPI_LD + cq);
        const float4 r = *(const float4*)(p.resid + row * p.ld_r + n);
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
        ff1(0, r - L.AO_N);
      }
    }
  } else if (q < L.PRO + L.LOOP) {
    const int u = (q - L.PRO) / L.BODY;
    int r = (q - L.PRO) % L.BODY;
    if (r < L.G1) {
      ff1(u + 1, r);
    } else if (r < L.G1 + L.G2) {
      r -= L.G1;
      const int g = r / F, f = r % F;
      for (int e = 0; e < 8; ++e) v[e] = w.w2[(long)(32 * f + r32) * (4 * C) + 32 * u + rc_perm(g, h, e)];
    }
  } else if (q < L.NT) {
    int r = q - L.PRO - L.LOOP;
    if (r < F) {
      bias_pair(w.b2[32 * r + r32]);
    } else if (L.po && r < F + KC * F) {
      r -= F;
      const int kk = r / F, f = r % F;
      for (int e = 0; e < 8; ++e) v[e] = w.w_po[(long)(32 * f + r32) * C + rc_perm(kk, h, e)];
    }
  }
  h8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
  *(h8*)(out + gid * 8) = o;
}

template <int C, bool AO, bool PO>
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

template <int C>
int launch_rc_c(const RowChain& p, int ao, int po, hipStream_t s) {
  if (ao) return po ? launch_rc<C, true, true>(p, s) : launch_rc<C, true, false>(p, s);
  return po ? launch_rc<C, false, true>(p, s) : launch_rc<C, false, false>(p, s);
}

}  // namespace

bool rowchain_takes(int C, int rows, int T) { return rc_supported_c(C) && rows > 0 && rows % 128 == 0 && T % 32 == 0; }

size_t rowchain_stream_halfs(int C, int ao, int po) { return (size_t)rc_layout(C, ao != 0, po != 0).NT_ALLOC * 512; }

int rowchain_pack(const RcWeights& w, int C, int ao, int po, float* tmp, half_t* stream, hipStream_t s) {
  if (!rc_supported_c(C)) return mvd_fail("rowchain_pack: unsupported width");
  const RcLayout L = rc_layout(C, ao != 0, po != 0);
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
  switch (C) {
    case 64: return launch_rc_c<64>(p, ao, po, s);
    case 128: return launch_rc_c<128>(p, ao, po, s);
    case 256: return launch_rc_c<256>(p, ao, po, s);
    default: return launch_rc_c<320>(p, ao, po, s);
  }
}
