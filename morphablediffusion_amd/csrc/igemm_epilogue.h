// Fused epilogue shared by the implicit GEMM and the halo 3x3 conv kernel.
#pragma once
#include "common.h"

// Exact-erf GELU (modules/attention.py:44) for the GEGLU epilogue of the FF1 GEMMs, which evaluates it 370 M times per step and
// is VALU-bound there.  erf through Abramowitz & Stegun 7.1.28, erfc(a) = (1 + a1 a + ... + a6 a^6)^-16 (|error| <= 3e-7; the
// product value * gelu(gate) is within 8e-7 absolute of the erff form, three orders below the fp16 rounding of the result): ONE
// transcendental (rcp) per element instead of the rcp + exp of 7.1.26, and the polynomial / squarings / products on packed
// fp32 (v_pk_fma_f32: two elements per lane and instruction) -- 11 VALU instructions per element against 17 for 7.1.26 and
// 38 for libm's erff.  -DMVD_GELU_LIBM restores erff (A/B builds).  An overflowing power (|gate| > ~17) gives rcp(inf) = 0.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef MVD_GELU_LIBM
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
#else
  const float a = fabsf(x) * 0.70710678118654752f;
  float p = __builtin_fmaf(0.0000430638f, a, 0.0002765672f);
  p = __builtin_fmaf(p, a, 0.0001520143f);
  p = __builtin_fmaf(p, a, 0.0092705272f);
  p = __builtin_fmaf(p, a, 0.0422820123f);
  p = __builtin_fmaf(p, a, 0.0705230784f);
  p = __builtin_fmaf(p, a, 1.0f);
  p *= p; p *= p; p *= p; p *= p;
  const float h = 0.5f * __builtin_amdgcn_rcpf(p);  // erfc(|x| / sqrt 2) / 2
  return x * (x >= 0.f ? 1.0f - h : h);
#endif
}
// value * gelu(gate) for two elements at once
__device__ __forceinline__ f32x2 geglu_pair(f32x2 x, f32x2 q) {
#ifdef MVD_GELU_LIBM
  f32x2 o;
  o.x = x.x * gelu_erf(q.x);
  o.y = x.y * gelu_erf(q.y);
  return o;
#else
  const f32x2 a = __builtin_elementwise_abs(q) * 0.70710678118654752f;
  f32x2 p = a * 0.0000430638f + 0.0002765672f;
  p = p * a + 0.0001520143f;
  p = p * a + 0.0092705272f;
  p = p * a + 0.0422820123f;
  p = p * a + 0.0705230784f;
  p = p * a + 1.0f;
  p = p * p; p = p * p; p = p * p; p = p * p;
  f32x2 h;
  h.x = __builtin_amdgcn_rcpf(p.x);
  h.y = __builtin_amdgcn_rcpf(p.y);
  const f32x2 lo = h * 0.5f, hi = 1.0f - lo;
  f32x2 phi;
  phi.x = q.x >= 0.f ? hi.x : lo.x;
  phi.y = q.y >= 0.f ? hi.y : lo.y;
  return x * q * phi;
#endif
}

__device__ __forceinline__ long out_row_off(const IGemm& g, int m, int ozo, int oyo, int oxo) {
  if (g.out_linear) return m;
  int x = m % g.X;
  int t = m / g.X;
  int y = t % g.Y;
  t /= g.Y;
  int z = t % g.Z;
  int b = t / g.Z;
  return ((long)(b * g.OZ + z * g.ozm + ozo) * g.OY + (y * g.oym + oyo)) * g.OX + (x * g.oxm + oxo);
}
__device__ __forceinline__ long out_row(const IGemm& g, int m) { return out_row_off(g, m, g.ozo, g.oyo, g.oxo); }

// v: accumulator for column n (and `gate` for column n+32 when geglu)
__device__ __forceinline__ void igemm_epilogue_store(const IGemm& g, int m, long orow, int n, float v, float gate) {
  v *= g.alpha;
  if (g.bias) v += g.bias[n];
  int ncol = n;
  if (g.geglu) {
    gate *= g.alpha;
    if (g.bias) gate += g.bias[n + 32];
    v = v * gelu_erf(gate);
    ncol = (n >> 6) * 32 + (n & 31);
  } else {
    if (g.rowbias) {
      int b = m / (g.Z * g.Y * g.X);
      v += g.rowbias[(long)b * g.rb_ld + n];
    }
    if (g.resid) {
      if (g.resid_f32) v += ((const float*)g.resid)[orow * g.ldr + n];
      else v += (float)((const half_t*)g.resid)[orow * g.ldr + n];
    }
    if (g.act == ACT_SILU) v = v / (1.0f + __expf(-v));
  }
  if (g.out_f32) ((float*)g.out)[orow * g.ldc + ncol] = v;
  else {
    const half_t hi = (half_t)v;
    ((half_t*)g.out)[orow * g.ldc + ncol] = hi;
    if (g.out_split) {
      ((half_t*)g.out)[orow * g.ldc + g.out_split + ncol] = (half_t)(v - (float)hi);
      ((half_t*)g.out)[orow * g.ldc + 2 * g.out_split + ncol] = hi;
    }
  }
}

// fp16 store of 4 consecutive columns; with out_split also the rounding residual and a second copy (IGemm::out_split)
__device__ __forceinline__ void store_h4_split(const IGemm& g, long o, int n, float4 v) {
  h4 hv;
  hv[0] = (half_t)v.x; hv[1] = (half_t)v.y; hv[2] = (half_t)v.z; hv[3] = (half_t)v.w;
  half_t* p = (half_t*)g.out + o * g.ldc + n;
  *(h4*)p = hv;
  if (g.out_split) {
    h4 lo;
    lo[0] = (half_t)(v.x - (float)hv[0]); lo[1] = (half_t)(v.y - (float)hv[1]);
    lo[2] = (half_t)(v.z - (float)hv[2]); lo[3] = (half_t)(v.w - (float)hv[3]);
    *(h4*)(p + g.out_split) = lo;
    *(h4*)(p + 2 * g.out_split) = hv;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Vectorised tile epilogue.  In the MFMA C layout a lane owns ONE column and 16 rows of a 32x32 fragment, so
// a direct store is 16 x 4-byte writes per fragment per lane (issue-bound, 128-byte segments).  Instead each
// wave transposes the fragment through a private LDS scratch ([32][36] floats) and every lane then owns
// 4 x (one row, 4 consecutive columns): bias / per-sample bias / residual are 16-byte loads and the result is
// one 16-byte (fp32) or 8-byte (fp16) store, 8 lanes covering 128 contiguous bytes of an output row.
// Requires N % 4 == 0, ldc % 4 == 0, ldr % 4 == 0 and no GEGLU (checked once per launch: igemm_fast_epi).
// ---------------------------------------------------------------------------------------------------------
constexpr int EPI_LD = 36;                          // floats per scratch row (16-byte aligned, conflict-light)
constexpr int EPI_WAVE_BYTES = 32 * EPI_LD * 4;     // 4608 B per wave

__device__ __forceinline__ bool igemm_fast_epi(const IGemm& g) {
  return !g.geglu && (g.N & 3) == 0 && (g.ldc & 3) == 0 && (!g.resid || (g.ldr & 3) == 0) &&
         (!g.rowbias || (g.rb_ld & 3) == 0);
}

// bias / per-sample bias / residual / SiLU / store for 4 consecutive columns n..n+3 of GEMM row m
// bs >= 0: the sample index of row m (for the per-sample bias), already known to the caller
// skip_bias: bias and per-sample bias are already in the accumulator (gemm_dma_kernel pre-loads them)
__device__ __forceinline__ void epilogue_vec4(const IGemm& g, int m, long o, int n, float4 v, int bs_known = -1,
                                              bool skip_bias = false) {
    v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
    if (g.bias && !skip_bias) {
      const float4 b = *(const float4*)(g.bias + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (g.rowbias && !skip_bias) {
      const int bs = bs_known >= 0 ? bs_known : m / (g.Z * g.Y * g.X);
      const float4 b = *(const float4*)(g.rowbias + (long)bs * g.rb_ld + n);
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (g.resid) {
      if (g.resid_f32) {
        const float4 q = *(const float4*)((const float*)g.resid + o * g.ldr + n);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
      } else {
        const h4 q = *(const h4*)((const half_t*)g.resid + o * g.ldr + n);
        v.x += (float)q[0]; v.y += (float)q[1]; v.z += (float)q[2]; v.w += (float)q[3];
      }
    }
    if (g.act == ACT_SILU) {
      v.x = v.x / (1.0f + __expf(-v.x)); v.y = v.y / (1.0f + __expf(-v.y));
      v.z = v.z / (1.0f + __expf(-v.z)); v.w = v.w / (1.0f + __expf(-v.w));
    }
    if (g.out_f32) {
      *(float4*)((float*)g.out + o * g.ldc + n) = v;
    } else {
      store_h4_split(g, o, n, v);
    }
}

// rows4[i] / orow4[i]: GEMM row index m and output row of fragment row (lane>>3) + 8 i  (m < 0: skip)
__device__ __forceinline__ void epilogue_frag_store(const IGemm& g, const f32x16& acc, float* scratch, int lane,
                                                    const int (&rows4)[4], const long (&orow4)[4], int n_base,
                                                    float* partial, const int* bs4 = nullptr, bool skip_bias = false) {
  // C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31)] = acc[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes are visible to its own reads
  const int cq = (lane & 7) * 4;
  const int n = n_base + cq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int fr = (lane >> 3) + 8 * i;
    const int m = rows4[i];
    if (m < 0 || n >= g.N) continue;
    float4 v = *(const float4*)(scratch + fr * EPI_LD + cq);
    if (partial) {
      *(float4*)(partial + (long)m * g.N + n) = v;
      continue;
    }
    epilogue_vec4(g, m, orow4[i], n, v, bs4 ? bs4[i] : -1, skip_bias);
  }
}

// Two-phase form of the fragment epilogue: epilogue_prefetch issues every global load the epilogue of one fragment
// needs (bias, per-sample bias, residual) and returns their sum; a kernel calls it for ALL its fragments before the
// first LDS transpose, so the residual reads of a 42 MB tensor are in flight together instead of one fragment at a time
// behind each transpose (the in-situ conv ran ~20 us above its isolated time mostly for that).
__device__ __forceinline__ void epilogue_prefetch(const IGemm& g, int lane, const int (&rows4)[4], const long (&orow4)[4],
                                                  int n_base, float4 (&pre)[4], const int* bs4 = nullptr,
                                                  bool skip_bias = false) {
  const int n = n_base + (lane & 7) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    const int m = rows4[i];
    if (m >= 0 && n < g.N) {
      if (g.bias && !skip_bias) {
        const float4 b = *(const float4*)(g.bias + n);
        p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w;
      }
      if (g.rowbias && !skip_bias) {
        const int bs = bs4 ? bs4[i] : m / (g.Z * g.Y * g.X);
        const float4 b = *(const float4*)(g.rowbias + (long)bs * g.rb_ld + n);
        p.x += b.x; p.y += b.y; p.z += b.z; p.w += b.w;
      }
      if (g.resid) {
        if (g.resid_f32) {
          const float4 q = *(const float4*)((const float*)g.resid + orow4[i] * g.ldr + n);
          p.x += q.x; p.y += q.y; p.z += q.z; p.w += q.w;
        } else {
          const h4 q = *(const h4*)((const half_t*)g.resid + orow4[i] * g.ldr + n);
          p.x += (float)q[0]; p.y += (float)q[1]; p.z += (float)q[2]; p.w += (float)q[3];
        }
      }
    }
    pre[i] = p;
  }
}

__device__ __forceinline__ void epilogue_frag_store_pre(const IGemm& g, const f32x16& acc, float* scratch, int lane,
                                                        const int (&rows4)[4], const long (&orow4)[4], int n_base,
                                                        const float4 (&pre)[4]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31)] = acc[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int cq = (lane & 7) * 4;
  const int n = n_base + cq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int fr = (lane >> 3) + 8 * i;
    if (rows4[i] < 0 || n >= g.N) continue;
    float4 v = *(const float4*)(scratch + fr * EPI_LD + cq);
    v.x = v.x * g.alpha + pre[i].x; v.y = v.y * g.alpha + pre[i].y;
    v.z = v.z * g.alpha + pre[i].z; v.w = v.w * g.alpha + pre[i].w;
    if (g.act == ACT_SILU) {
      v.x = v.x / (1.0f + __expf(-v.x)); v.y = v.y / (1.0f + __expf(-v.y));
      v.z = v.z / (1.0f + __expf(-v.z)); v.w = v.w / (1.0f + __expf(-v.w));
    }
    if (g.out_f32) {
      *(float4*)((float*)g.out + orow4[i] * g.ldc + n) = v;
    } else {
      store_h4_split(g, orow4[i], n, v);
    }
  }
}

// transposed store of an already-final fragment (no bias / residual): columns ncol_base + [0, 32) of g.out, < ncols
__device__ __forceinline__ void epilogue_frag_store_raw(const IGemm& g, const f32x16& v, float* scratch, int lane,
                                                        const int (&rows4)[4], const long (&orow4)[4], int ncol_base,
                                                        int ncols) {
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31)] = v[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int cq = (lane & 7) * 4;
  const int n = ncol_base + cq;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int fr = (lane >> 3) + 8 * i;
    if (rows4[i] < 0 || n >= ncols) continue;
    const float4 x = *(const float4*)(scratch + fr * EPI_LD + cq);
    if (g.out_f32) {
      *(float4*)((float*)g.out + orow4[i] * g.ldc + n) = x;
    } else {
      h4 hv;
      hv[0] = (half_t)x.x; hv[1] = (half_t)x.y; hv[2] = (half_t)x.z; hv[3] = (half_t)x.w;
      *(h4*)((half_t*)g.out + orow4[i] * g.ldc + n) = hv;
    }
  }
}

// ---- fp16 output, linear rows, 8 columns per lane -------------------------------------------------------------
// The store path is issue-bound (a wave's 8-byte stores move half the bytes per cycle of its 16-byte stores; measured
// with tools/gemm_timeline.py: 4.7 us for the fp16 epilogue of a 256 x 160 tile, the same as for its fp32 epilogue), so
// fp16 results leave as 16-byte stores: lane -> rows (lane >> 2) + 16 i (i = 0, 1), columns 8 (lane & 3) .. + 8.
// m_w: GEMM row of the wave's fragment row 0; rows are output rows (out_linear), the sample of a row is
// floor((m + 0.5) * inv_rps).  Needs ldc, ldr, rb_ld, out_split and the column count to be multiples of 8.
__device__ __forceinline__ bool epilogue8_ok(const IGemm& g, int ncols) {
  return !g.out_f32 && g.out_linear && !((g.ldc | ncols | g.out_split) & 7) && (!g.resid || !(g.ldr & 7)) &&
         (!g.rowbias || !(g.rb_ld & 7));
}

__device__ __forceinline__ void frag_to_scratch(const f32x16& acc, float* scratch, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) scratch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31)] = acc[r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own writes are visible to its own reads
}

// every global load of one fragment's epilogue (bias + per-sample bias + residual), summed: issued for all fragments
// of a tile before the first transpose, so they are in flight together
__device__ __forceinline__ void epilogue8_prefetch(const IGemm& g, int lane, int m_w, int M, int n_base, float inv_rps,
                                                   float (&pre)[2][8], bool skip_bias = false) {
  const int n = n_base + (lane & 3) * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_w + (lane >> 2) + 16 * i;
#pragma unroll
    for (int e = 0; e < 8; ++e) pre[i][e] = 0.f;
    if (m >= M || n >= g.N) continue;
    if (g.bias && !skip_bias) {
      const float4 b0 = *(const float4*)(g.bias + n), b1 = *(const float4*)(g.bias + n + 4);
      pre[i][0] += b0.x; pre[i][1] += b0.y; pre[i][2] += b0.z; pre[i][3] += b0.w;
      pre[i][4] += b1.x; pre[i][5] += b1.y; pre[i][6] += b1.z; pre[i][7] += b1.w;
    }
    if (g.rowbias && !skip_bias) {
      const float* rb = g.rowbias + (long)(int)(((float)m + 0.5f) * inv_rps) * g.rb_ld + n;
      const float4 b0 = *(const float4*)rb, b1 = *(const float4*)(rb + 4);
      pre[i][0] += b0.x; pre[i][1] += b0.y; pre[i][2] += b0.z; pre[i][3] += b0.w;
      pre[i][4] += b1.x; pre[i][5] += b1.y; pre[i][6] += b1.z; pre[i][7] += b1.w;
    }
    if (g.resid) {
      if (g.resid_f32) {
        const float* rp = (const float*)g.resid + (long)m * g.ldr + n;
        const float4 q0 = *(const float4*)rp, q1 = *(const float4*)(rp + 4);
        pre[i][0] += q0.x; pre[i][1] += q0.y; pre[i][2] += q0.z; pre[i][3] += q0.w;
        pre[i][4] += q1.x; pre[i][5] += q1.y; pre[i][6] += q1.z; pre[i][7] += q1.w;
      } else {
        const h8 q = *(const h8*)((const half_t*)g.resid + (long)m * g.ldr + n);
#pragma unroll
        for (int e = 0; e < 8; ++e) pre[i][e] += (float)q[e];
      }
    }
  }
}

// RAW: v is final (GEGLU product, folded GroupNorm): plain transposed store into columns ncol_base + [0, 32) < ncols.
// Otherwise out = act(acc * alpha + pre) with the optional [hi | lo | hi] split.
template <bool RAW>
__device__ __forceinline__ void epilogue8_frag_store(const IGemm& g, const f32x16& acc, float* scratch, int lane, int m_w,
                                                     int M, int ncol_base, int ncols, const float (*pre)[8]) {
  frag_to_scratch(acc, scratch, lane);
  const int c8 = (lane & 3) * 8, n = ncol_base + c8;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int fr = (lane >> 2) + 16 * i, m = m_w + fr;
    if (m >= M || n >= ncols) continue;
    const float4 a = *(const float4*)(scratch + fr * EPI_LD + c8), b = *(const float4*)(scratch + fr * EPI_LD + c8 + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if constexpr (!RAW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = v[e] * g.alpha + pre[i][e];
        if (g.act == ACT_SILU) v[e] = v[e] / (1.0f + __expf(-v[e]));
      }
    }
    h8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) hv[e] = (half_t)v[e];
    half_t* p = (half_t*)g.out + (long)m * g.ldc + n;
    *(h8*)p = hv;
    if (!RAW && g.out_split) {
      h8 lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) lo[e] = (half_t)(v[e] - (float)hv[e]);
      *(h8*)(p + g.out_split) = lo;
      *(h8*)(p + 2 * g.out_split) = hv;
    }
  }
}

// GEGLU variant: fragment 0 holds 32 value columns, fragment 1 the matching 32 gate columns (weights are packed
// in alternating 32-row blocks).  out[.., (n>>6)*32 + (n&31)] = (x + b_x) * gelu_erf(gate + b_g), 4 columns per lane.
__device__ __forceinline__ void epilogue_geglu_frag_store(const IGemm& g, const f32x16& ax, const f32x16& ag, float* sx,
                                                          float* sg, int lane, const int (&rows4)[4],
                                                          const long (&orow4)[4], int n_base) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EPI_LD + (lane & 31);
    sx[o] = ax[r];
    sg[o] = ag[r];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int cq = (lane & 7) * 4;
  const int n = n_base + cq;  // value column; gate column = n + 32
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int fr = (lane >> 3) + 8 * i;
    if (rows4[i] < 0 || n + 32 >= g.N) continue;
    float4 x = *(const float4*)(sx + fr * EPI_LD + cq);
    float4 q = *(const float4*)(sg + fr * EPI_LD + cq);
    x.x *= g.alpha; x.y *= g.alpha; x.z *= g.alpha; x.w *= g.alpha;
    q.x *= g.alpha; q.y *= g.alpha; q.z *= g.alpha; q.w *= g.alpha;
    if (g.bias) {
      const float4 bx = *(const float4*)(g.bias + n), bg = *(const float4*)(g.bias + n + 32);
      x.x += bx.x; x.y += bx.y; x.z += bx.z; x.w += bx.w;
      q.x += bg.x; q.y += bg.y; q.z += bg.z; q.w += bg.w;
    }
    float4 v;
    {
      const f32x2 v0 = geglu_pair(f32x2{x.x, x.y}, f32x2{q.x, q.y}), v1 = geglu_pair(f32x2{x.z, x.w}, f32x2{q.z, q.w});
      v.x = v0.x; v.y = v0.y; v.z = v1.x; v.w = v1.y;
    }
    const int ncol = (n >> 6) * 32 + (n & 31);
    if (g.out_f32) {
      *(float4*)((float*)g.out + orow4[i] * g.ldc + ncol) = v;
    } else {
      h4 hv;
      hv[0] = (half_t)v.x; hv[1] = (half_t)v.y; hv[2] = (half_t)v.z; hv[3] = (half_t)v.w;
      *(h4*)((half_t*)g.out + orow4[i] * g.ldc + ncol) = hv;
    }
  }
}
