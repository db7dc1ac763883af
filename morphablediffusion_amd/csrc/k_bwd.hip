// Backward-pass kernels of the training step (reference training_step morphable_diffusion.py:520-549 +
// loss.backward()).  The matrix products of the backward pass run on the SAME MFMA kernels as the forward pass:
//   dgrad  = the forward implicit GEMM on transposed (and, for 3x3, tap-flipped) packed weights (pack_dgrad_weight),
//   wgrad  = a plain GEMM  dW[Cout][Cin*taps] = dY^T [Cout][R] x col(X)^T [Cin*taps][R]^T  whose two operands are made
//            K-contiguous (K = pixel rows) by the transposing casts below (tcast / im2colT).
// What is left for this file is the bandwidth-bound part: transposing casts, GroupNorm / LayerNorm / GEGLU backward, the
// reductions for bias / gain gradients, the flash-style self-attention backward (MFMA, same transposed formulation as
// k_attn.hip), nearest-upsample / strided-conv scatter-free adjoints, AdamW.
// Everything is deterministic: fixed summation orders, no atomics.  Activation gradients are fp32 in HBM (carrying the
// caller's loss scale), MFMA operands fp16.
#include "common.h"

namespace {

inline int gridn(size_t n, int cap = 8192) {
  size_t g = (n + 255) / 256;
  return (int)(g > (size_t)cap ? (size_t)cap : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const half_t* p) { return (float)*p; }

// ---------------------------------------------------------------------------------------------------------------------
// dst[c][r] = fp16(src[r][c]) for c < C, r < Rp (zero for r >= R): the K-contiguous operand of a wgrad GEMM.
// 64 x 64 tiles through LDS; reads are coalesced along c, writes along r.
// split != 0 (fp32 sources): every dst row holds three Rp-long segments -- split 1: [hi | lo | hi], split 2: [hi | hi | lo]
// with hi = fp16(x), lo = fp16(x - hi) -- the two operand images of an extended-precision GEMM (a_hi b_hi + a_lo b_hi +
// a_hi b_lo by concatenation along K, the forward pass's ConvW::xp trick).
// rows16 (optional): the row-major fp16 image [R][Cp] of the same tile as well (columns C..Cp zero) -- the A operand of the
// dgrad GEMM that accompanies the wgrad GEMM of a layer: one read of the fp32 gradient instead of two kernels
template <typename T>
__global__ __launch_bounds__(256) void tcast_kernel(const T* __restrict__ src, long ld, int R, int C, half_t* __restrict__ dst, int Rp,
                                                    int split, half_t* __restrict__ rows16, int Cp) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    const float v = (r < R && c < C) ? ldf(src + (long)r * ld + c) : 0.f;
    tile[rr][cc] = v;
    if (rows16 && r < R && c < Cp) rows16[(long)r * Cp + c] = (half_t)v;
  }
  __syncthreads();
  const long rowlen = split ? 3L * Rp : Rp;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cc = i >> 6, rr = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    if (c < C && r < Rp) {
      const float v = tile[rr][cc];
      const half_t hi = (half_t)v;
      half_t* d = dst + (long)c * rowlen + r;
      d[0] = hi;
      if (split) {
        const half_t lo = (half_t)(v - (float)hi);
        d[Rp] = split == 1 ? lo : hi;
        d[2L * Rp] = split == 1 ? hi : lo;
      }
    }
  }
}

// The same with 16-byte accesses (round 4): a thread loads four consecutive columns (float4 / four halfs), writes the row image as
// four halfs, and the transposed image as EIGHT consecutive rows of one column (one 16-byte store; eight lanes cover 128
// contiguous bytes of a dst row).  The element-per-thread form above moved 2 bytes per lane and store instruction and was the
// largest item of the training step's operand staging (401 launches, 6.4 ms at 8 samples).  Same values.
// Needs C % 4 == 0, ld % 4 == 0, Rp % 64 == 0, Cp % 4 == 0 and aligned pointers (bwd_tcast checks; otherwise the scalar kernel).
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  const float4 q = *(const float4*)p;
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void ld4(const half_t* p, float (&v)[4]) {
  const h4 q = *(const h4*)p;
  v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
}
template <typename T>
__global__ __launch_bounds__(256) void tcast_vec_kernel(const T* __restrict__ src, long ld, int R, int C, half_t* __restrict__ dst, int Rp,
                                                        int split, half_t* __restrict__ rows16, int Cp) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int rr = i >> 4, cc = (i & 15) * 4;
    const int r = r0 + rr, c = c0 + cc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R && c < C) ld4(src + (long)r * ld + c, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rr][cc + e] = v[e];
    if (rows16 && r < R && c < Cp) {
      h4 hv;
#pragma unroll
      for (int e = 0; e < 4; ++e) hv[e] = (half_t)v[e];
      *(h4*)(rows16 + (long)r * Cp + c) = hv;
    }
  }
  __syncthreads();
  const long rowlen = split ? 3L * Rp : Rp;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int cc = i >> 3, g = i & 7;
    const int c = c0 + cc, r = r0 + g * 8;
    if (c >= C) continue;
    h8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = tile[g * 8 + k][cc];
      hi[k] = (half_t)v;
      lo[k] = (half_t)(v - (float)hi[k]);
    }
    half_t* d = dst + (long)c * rowlen + r;
    *(h8*)d = hi;
    if (split) {
      *(h8*)(d + Rp) = split == 1 ? lo : hi;
      *(h8*)(d + 2L * Rp) = split == 1 ? hi : lo;
    }
  }
}

// dst[(ci * 9 + tap)][r] = fp16(X[b, (yo*s + ky - 1) >> ups, (xo*s + kx - 1) >> ups, ci])  (zero outside the virtual image of
// (H << ups) x (W << ups) and for r >= R); r = (b*Ho + yo)*Wo + xo.  The row order ci*9 + tap makes the wgrad GEMM's
// output [Cout][Cin*9] the reference's conv weight layout [Cout][Cin][3][3] itself.
template <typename T>
__global__ __launch_bounds__(256) void im2colT_kernel(const T* __restrict__ src, long ld, int B, int H, int W, int C, int stride, int ups,
                                                      int Ho, int Wo, half_t* __restrict__ dst, int Rp) {
  __shared__ half_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int ky = tap / 3 - 1, kx = tap % 3 - 1;
  const int R = B * Ho * Wo, IY = H << ups, IX = W << ups;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    half_t v = (half_t)0;
    if (r < R && c < C) {
      const int xo = r % Wo, yo = (r / Wo) % Ho, b = r / (Wo * Ho);
      const int y = yo * stride + ky, x = xo * stride + kx;
      if (y >= 0 && y < IY && x >= 0 && x < IX) v = (half_t)ldf(src + (((long)b * H + (y >> ups)) * W + (x >> ups)) * ld + c);
    }
    tile[rr][cc] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cc = i >> 6, rr = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    if (c < C && r < Rp) dst[((long)c * 9 + tap) * Rp + r] = tile[rr][cc];
  }
}

// im2colT with 8- / 16-byte accesses (round 4): a thread gathers four consecutive channels of one output row (one set of index
// divisions per four elements instead of per element) and stores eight consecutive rows of one (channel, tap) line at once.
// Needs C % 4 == 0, ld % 4 == 0, Rp % 64 == 0 and aligned pointers; same values as im2colT_kernel.
template <typename T>
__global__ __launch_bounds__(256) void im2colT_vec_kernel(const T* __restrict__ src, long ld, int B, int H, int W, int C, int stride, int ups,
                                                          int Ho, int Wo, half_t* __restrict__ dst, int Rp) {
  __shared__ half_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int ky = tap / 3 - 1, kx = tap % 3 - 1;
  const int R = B * Ho * Wo, IY = H << ups, IX = W << ups;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int rr = i >> 4, cc = (i & 15) * 4;
    const int r = r0 + rr, c = c0 + cc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R && c < C) {
      const int xo = r % Wo, yo = (r / Wo) % Ho, b = r / (Wo * Ho);
      const int y = yo * stride + ky, x = xo * stride + kx;
      if (y >= 0 && y < IY && x >= 0 && x < IX) ld4(src + (((long)b * H + (y >> ups)) * W + (x >> ups)) * ld + c, v);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rr][cc + e] = (half_t)v[e];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int cc = i >> 3, g = i & 7;
    const int c = c0 + cc;
    if (c >= C) continue;
    h8 hv;
#pragma unroll
    for (int k = 0; k < 8; ++k) hv[k] = tile[g * 8 + k][cc];
    *(h8*)(dst + ((long)c * 9 + tap) * Rp + r0 + g * 8) = hv;
  }
}

// fp32 rows (stride ld) -> dense fp16 rows of width Cp >= C (zero padded); split: rows of 3 Cp halfs, see tcast_kernel
__global__ void cast_rows_kernel(const float* __restrict__ src, long ld, long rows, int C, int Cp, half_t* __restrict__ dst, int split) {
  const long total = rows * Cp;
  const long rowlen = split ? 3L * Cp : Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const long r = i / Cp;
    const float v = c < C ? src[r * ld + c] : 0.f;
    const half_t hi = (half_t)v;
    half_t* d = dst + r * rowlen + c;
    d[0] = hi;
    if (split) {
      const half_t lo = (half_t)(v - (float)hi);
      d[Cp] = split == 1 ? lo : hi;
      d[2L * Cp] = split == 1 ? hi : lo;
    }
  }
}

// four columns per thread (float4 in, four halfs out); needs C % 4 == 0, Cp % 4 == 0, ld % 4 == 0, aligned pointers
__global__ void cast_rows_vec_kernel(const float* __restrict__ src, long ld, long rows, int C, int Cp, half_t* __restrict__ dst, int split) {
  const int q = Cp >> 2;
  const long total = rows * q;
  const long rowlen = split ? 3L * Cp : Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    const long r = i / q;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) v = *(const float4*)(src + r * ld + c);
    h4 hi;
    hi[0] = (half_t)v.x; hi[1] = (half_t)v.y; hi[2] = (half_t)v.z; hi[3] = (half_t)v.w;
    half_t* d = dst + r * rowlen + c;
    *(h4*)d = hi;
    if (split) {
      h4 lo;
      lo[0] = (half_t)(v.x - (float)hi[0]); lo[1] = (half_t)(v.y - (float)hi[1]);
      lo[2] = (half_t)(v.z - (float)hi[2]); lo[3] = (half_t)(v.w - (float)hi[3]);
      *(h4*)(d + Cp) = split == 1 ? lo : hi;
      *(h4*)(d + 2L * Cp) = split == 1 ? hi : lo;
    }
  }
}

// packed forward weight fp16 [taps][N][ldw] (first Cl columns of each row are w_hi) -> dgrad weight [taps][Cl][Np]:
// wT[taps-1-t][ci][co] = w[t][co][ci]  (the tap flip turns the forward cross-correlation into its adjoint; Np >= N zero padded)
// flip = 0: the taps keep their index (the adjoint of a strided conv / of a transposed conv is launched as the OTHER kind, whose
// tap tables already carry the index relation o = 2 i - 1 + k)
// 64 x 64 tiles through LDS: 128-byte row segments on both sides (the element-per-thread form read 2 bytes per cache line)
__global__ __launch_bounds__(256) void pack_dgrad_kernel(const half_t* __restrict__ w, int taps, int N, int ldw, int Cl, int Np,
                                                         half_t* __restrict__ wT, int flip) {
  __shared__ half_t tile[64][66];
  const int t2 = blockIdx.z, t = flip ? taps - 1 - t2 : t2;
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = co0 + ty + 4 * i, ci = ci0 + tx;
    tile[ty + 4 * i][tx] = (co < N && ci < Cl) ? w[((long)t * N + co) * ldw + ci] : (half_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int ci = ci0 + ty + 4 * i, co = co0 + tx;
    if (ci < Cl && co < Np) wT[((long)t2 * Cl + ci) * Np + co] = tile[tx][ty + 4 * i];
  }
}

// the same with 16-byte accesses on both sides (needs ldw % 8 == 0, Np % 8 == 0, Cl % 8 == 0, aligned pointers)
__global__ __launch_bounds__(256) void pack_dgrad_vec_kernel(const half_t* __restrict__ w, int taps, int N, int ldw, int Cl, int Np,
                                                             half_t* __restrict__ wT, int flip) {
  __shared__ half_t tile[64][72];  // [co][ci]
  const int t2 = blockIdx.z, t = flip ? taps - 1 - t2 : t2;
  const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int col = i >> 3, c8 = (i & 7) * 8;
    const int co = co0 + col, ci = ci0 + c8;
    h8 v = (h8)(half_t)0;
    if (co < N && ci < Cl) v = *(const h8*)(w + ((long)t * N + co) * ldw + ci);
    *(h8*)(&tile[col][c8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int cil = i >> 3, g = i & 7;
    const int ci = ci0 + cil, co = co0 + g * 8;
    if (ci >= Cl || co >= Np) continue;
    h8 hv;
#pragma unroll
    for (int k = 0; k < 8; ++k) hv[k] = tile[g * 8 + k][cil];
    *(h8*)(wT + ((long)t2 * Cl + ci) * Np + co) = hv;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_f(float u, int act) { return act == ACT_SILU ? u / (1.f + __expf(-u)) : (act == ACT_RELU ? fmaxf(u, 0.f) : u); }
__device__ __forceinline__ float act_d(float u, int act) {
  if (act == ACT_SILU) {
    const float s = 1.f / (1.f + __expf(-u));
    return s * (1.f + u * (1.f - s));
  }
  return act == ACT_RELU ? (u > 0.f ? 1.f : 0.f) : 1.f;
}

__device__ float block_sum256(float v, float* s_red) {
  const int t = threadIdx.x;
  s_red[t] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  const float r = s_red[0];
  __syncthreads();
  return r;
}

// GroupNorm (+ activation) backward, one workgroup per (sample, group).  x [B][rows][ld] fp32 is the norm's INPUT (before the
// optional per-sample pre-add `pre` [B][pld]); dy the gradient w.r.t. act(gamma xhat + beta).
//   du = dy act'(u),  dxhat = du gamma,  dx = rstd (dxhat - mean_g(dxhat) - xhat mean_g(dxhat xhat))
// dx is written (accum = 0) or added.  Per-sample channel sums of du xhat / du / dx go to dg_part / db_part / dpre_part
// [B][C] (any may be null) -- summed over the samples by sum_rows_add.  Thread t owns channel t % cpg of the group and the rows
// t / cpg, + rpi, ...: every reduction runs in a fixed order.
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ x, long ld, const float* __restrict__ pre, int pld,
                                                     const float* __restrict__ dy, long ldy, int rows, int C, int G,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int act,
                                                     float* __restrict__ dx, long lddx, int accum, float* __restrict__ dg_part,
                                                     float* __restrict__ db_part, float* __restrict__ dpre_part) {
  __shared__ float s_red[256], s_a[256], s_b[256];
  const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G;
  const int rpi = 256 / cpg, t = threadIdx.x;
  const bool active = t < rpi * cpg;
  const int c = t % cpg, r0 = t / cpg, ch = g * cpg + c;
  const float n = (float)rows * (float)cpg;
  const float* xb = x + (long)b * rows * ld + ch;
  const float* dyb = dy + (long)b * rows * ldy + ch;
  const float pv = (pre && active) ? pre[(long)b * pld + ch] : 0.f;
  float a = 0.f;
  if (active)
    for (int r = r0; r < rows; r += rpi) a += xb[(long)r * ld] + pv;
  const float mean = block_sum256(a, s_red) / n;
  float q = 0.f;
  if (active)
    for (int r = r0; r < rows; r += rpi) {
      const float d = xb[(long)r * ld] + pv - mean;
      q += d * d;
    }
  const float rstd = rsqrtf(block_sum256(q, s_red) / n + eps);
  const float gm = active ? gamma[ch] : 0.f, bt = active ? beta[ch] : 0.f;
  float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
  if (active)
    for (int r = r0; r < rows; r += rpi) {
      const float xh = (xb[(long)r * ld] + pv - mean) * rstd;
      const float du = dyb[(long)r * ldy] * act_d(gm * xh + bt, act);
      const float dxh = du * gm;
      s1 += dxh;
      s2 += dxh * xh;
      dg += du * xh;
      db += du;
    }
  const float m1 = block_sum256(s1, s_red) / n, m2 = block_sum256(s2, s_red) / n;
  float dsum = 0.f;
  if (active) {
    float* dxb = dx + (long)b * rows * lddx + ch;
    for (int r = r0; r < rows; r += rpi) {
      const float xh = (xb[(long)r * ld] + pv - mean) * rstd;
      const float du = dyb[(long)r * ldy] * act_d(gm * xh + bt, act);
      const float v = rstd * (du * gm - m1 - xh * m2);
      dsum += v;
      dxb[(long)r * lddx] = accum ? dxb[(long)r * lddx] + v : v;
    }
  }
  // per-channel sums over the row lanes, in row-lane order
  s_a[t] = dg;
  s_b[t] = db;
  s_red[t] = dsum;
  __syncthreads();
  if (t < cpg) {
    float ag = 0.f, ab = 0.f, ad = 0.f;
    for (int k = 0; k < rpi; ++k) {
      ag += s_a[k * cpg + t];
      ab += s_b[k * cpg + t];
      ad += s_red[k * cpg + t];
    }
    const long o = (long)b * C + g * cpg + t;
    if (dg_part) dg_part[o] = ag;
    if (db_part) db_part[o] = ab;
    if (dpre_part) dpre_part[o] = ad;
  }
}

// ---- slabbed GroupNorm forward / backward: one workgroup per (sample, group, row slab), so that the few (sample, group)
// pairs of a small batch still fill the chip (B * G * S workgroups instead of B * G).  Three passes over the data:
//   gn_slab_stats:  per-slab (sum, sumsq) of x (+ pre)                     -> part [B*G][S][2]
//   gn_slab_sums:   mean / rstd from the S partials (fp64), then per-slab (sum dxhat, sum dxhat xhat) -> part2 [B*G][S][2]
//                   and per-(sample, slab) channel sums of du xhat / du   -> dg_part / db_part [B*S][C]
//   gn_slab_apply:  m1 / m2 from the S partials, dx for the slab's rows
// The forward (fp32 out) is gn_slab_stats + gn_slab_fwd.  Same thread mapping as gn_bwd_kernel (fixed summation orders).
__device__ __forceinline__ void gn_slab_meanrstd(const float* __restrict__ part, int S, float n, float eps, float* mean, float* rstd) {
  double sx = 0.0, sq = 0.0;
  for (int k = 0; k < S; ++k) {
    sx += (double)part[2 * k];
    sq += (double)part[2 * k + 1];
  }
  const double m = sx / (double)n;
  double var = sq / (double)n - m * m;
  if (var < 0.0) var = 0.0;
  *mean = (float)m;
  *rstd = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void gn_slab_stats_kernel(const float* __restrict__ x, long ld, const float* __restrict__ pre, int pld,
                                                            int rows, int C, int G, int S, int rps, float* __restrict__ part) {
  __shared__ float s_red[256];
  const int bg = blockIdx.x, sl = blockIdx.y, b = bg / G, g = bg % G, cpg = C / G;
  const int rpi = 256 / cpg, t = threadIdx.x;
  const bool active = t < rpi * cpg;
  const int c = t % cpg, r0 = t / cpg, ch = g * cpg + c;
  const int rbeg = sl * rps, rend = min(rows, rbeg + rps);
  const float* xb = x + (long)b * rows * ld + ch;
  const float pv = (pre && active) ? pre[(long)b * pld + ch] : 0.f;
  float a = 0.f, q = 0.f;
  if (active)
    for (int r = rbeg + r0; r < rend; r += rpi) {
      const float v = xb[(long)r * ld] + pv;
      a += v;
      q += v * v;
    }
  const float sa = block_sum256(a, s_red), sq = block_sum256(q, s_red);
  if (t == 0) {
    part[((long)bg * S + sl) * 2] = sa;
    part[((long)bg * S + sl) * 2 + 1] = sq;
  }
}

__global__ __launch_bounds__(256) void gn_slab_fwd_kernel(const float* __restrict__ x, long ld, const float* __restrict__ part, int rows,
                                                          int C, int G, int S, int rps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int act, float* __restrict__ y,
                                                          long ldy) {
  const int bg = blockIdx.x, sl = blockIdx.y, b = bg / G, g = bg % G, cpg = C / G;
  const int rpi = 256 / cpg, t = threadIdx.x;
  if (t >= rpi * cpg) return;
  const int c = t % cpg, r0 = t / cpg, ch = g * cpg + c;
  float mean, rstd;
  gn_slab_meanrstd(part + (long)bg * S * 2, S, (float)rows * (float)cpg, eps, &mean, &rstd);
  const float gm = gamma[ch] * rstd, bt = beta[ch] - mean * rstd * gamma[ch];
  const int rbeg = sl * rps, rend = min(rows, rbeg + rps);
  for (int r = rbeg + r0; r < rend; r += rpi) {
    const long o = ((long)b * rows + r);
    y[o * ldy + ch] = act_f(x[o * ld + ch] * gm + bt, act);
  }
}

__global__ __launch_bounds__(256) void gn_slab_sums_kernel(const float* __restrict__ x, long ld, const float* __restrict__ pre, int pld,
                                                           const float* __restrict__ dy, long ldy, const float* __restrict__ part,
                                                           int rows, int C, int G, int S, int rps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int act, float* __restrict__ part2,
                                                           float* __restrict__ dg_part, float* __restrict__ db_part) {
  __shared__ float s_red[256], s_a[256], s_b[256];
  const int bg = blockIdx.x, sl = blockIdx.y, b = bg / G, g = bg % G, cpg = C / G;
  const int rpi = 256 / cpg, t = threadIdx.x;
  const bool active = t < rpi * cpg;
  const int c = t % cpg, r0 = t / cpg, ch = g * cpg + c;
  float mean, rstd;
  gn_slab_meanrstd(part + (long)bg * S * 2, S, (float)rows * (float)cpg, eps, &mean, &rstd);
  const int rbeg = sl * rps, rend = min(rows, rbeg + rps);
  const float* xb = x + (long)b * rows * ld + ch;
  const float* dyb = dy + (long)b * rows * ldy + ch;
  const float pv = (pre && active) ? pre[(long)b * pld + ch] : 0.f;
  const float gm = active ? gamma[ch] : 0.f, bt = active ? beta[ch] : 0.f;
  float s1 = 0.f, s2 = 0.f, dg = 0.f, db = 0.f;
  if (active)
    for (int r = rbeg + r0; r < rend; r += rpi) {
      const float xh = (xb[(long)r * ld] + pv - mean) * rstd;
      const float du = dyb[(long)r * ldy] * act_d(gm * xh + bt, act);
      const float dxh = du * gm;
      s1 += dxh;
      s2 += dxh * xh;
      dg += du * xh;
      db += du;
    }
  const float t1 = block_sum256(s1, s_red), t2 = block_sum256(s2, s_red);
  if (t == 0) {
    part2[((long)bg * S + sl) * 2] = t1;
    part2[((long)bg * S + sl) * 2 + 1] = t2;
  }
  s_a[t] = dg;
  s_b[t] = db;
  __syncthreads();
  if (t < cpg) {
    float ag = 0.f, ab = 0.f;
    for (int k = 0; k < rpi; ++k) {
      ag += s_a[k * cpg + t];
      ab += s_b[k * cpg + t];
    }
    const long o = ((long)b * S + sl) * C + g * cpg + t;
    if (dg_part) dg_part[o] = ag;
    if (db_part) db_part[o] = ab;
  }
}

__global__ __launch_bounds__(256) void gn_slab_apply_kernel(const float* __restrict__ x, long ld, const float* __restrict__ pre, int pld,
                                                            const float* __restrict__ dy, long ldy, const float* __restrict__ part,
                                                            const float* __restrict__ part2, int rows, int C, int G, int S, int rps,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int act, float* __restrict__ dx, long lddx, int accum,
                                                            float* __restrict__ dpre_part) {
  __shared__ float s_red[256];
  const int bg = blockIdx.x, sl = blockIdx.y, b = bg / G, g = bg % G, cpg = C / G;
  const int rpi = 256 / cpg, t = threadIdx.x;
  const bool active = t < rpi * cpg;
  const int c = t % cpg, r0 = t / cpg, ch = g * cpg + c;
  const float n = (float)rows * (float)cpg;
  float mean, rstd;
  gn_slab_meanrstd(part + (long)bg * S * 2, S, n, eps, &mean, &rstd);
  double a1 = 0.0, a2 = 0.0;
  for (int k = 0; k < S; ++k) {
    a1 += (double)part2[((long)bg * S + k) * 2];
    a2 += (double)part2[((long)bg * S + k) * 2 + 1];
  }
  const float m1 = (float)(a1 / (double)n), m2 = (float)(a2 / (double)n);
  const int rbeg = sl * rps, rend = min(rows, rbeg + rps);
  const float pv = (pre && active) ? pre[(long)b * pld + ch] : 0.f;
  const float gm = active ? gamma[ch] : 0.f, bt = active ? beta[ch] : 0.f;
  float dsum = 0.f;
  if (active)
    for (int r = rbeg + r0; r < rend; r += rpi) {
      const long o = (long)b * rows + r;
      const float xh = (x[o * ld + ch] + pv - mean) * rstd;
      const float du = dy[o * ldy + ch] * act_d(gm * xh + bt, act);
      const float v = rstd * (du * gm - m1 - xh * m2);
      dsum += v;
      dx[o * lddx + ch] = accum ? dx[o * lddx + ch] + v : v;
    }
  if (dpre_part) {
    s_red[t] = dsum;
    __syncthreads();
    if (t < cpg) {
      float ad = 0.f;
      for (int k = 0; k < rpi; ++k) ad += s_red[k * cpg + t];
      dpre_part[((long)b * S + sl) * C + g * cpg + t] = ad;
    }
  }
}

// C[m][n] += sum_{k < K} a[k][m] * b[k][n] (fp32, exact summation order): the weight gradient of a Linear layer applied to a
// handful of rows (K = samples: time-embedding / attn2 projections), where a GEMM launch would be all overhead
__global__ void outer_add_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb, float* __restrict__ C,
                                 int M, int N, int K) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % N), m = (int)(i / N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += a[(long)k * lda + m] * b[(long)k * ldb + n];
    C[i] += acc;
  }
}

// out[c] (+)= sum_r part[r][c]: a workgroup owns 8 columns, 32 row lanes each, combined through LDS in lane order (fixed order
// for a given R)
struct SumJobs {  // up to four independent column sums in one launch (blockIdx.y picks the job)
  const float* part[4];
  float* out[4];
  long ldp[4];
  int R[4], C[4], accum[4];
};
__global__ __launch_bounds__(256) void sum_rows_add_kernel(const SumJobs j) {
  __shared__ float s_p[32][9];
  const int k = blockIdx.y, C = j.C[k], R = j.R[k];
  const int cc = threadIdx.x & 7, rl = threadIdx.x >> 3, c = blockIdx.x * 8 + cc;
  if (blockIdx.x * 8 >= C) return;
  const float* __restrict__ part = j.part[k];
  const long ldp = j.ldp[k];
  float a = 0.f;
  if (c < C)
    for (int r = rl; r < R; r += 32) a += part[(long)r * ldp + c];
  s_p[rl][cc] = a;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
    for (int i = 0; i < 32; ++i) t += s_p[i][cc];
    float* out = j.out[k];
    out[c] = j.accum[k] ? out[c] + t : t;
  }
}

// out[b][c] = sum over the rows of sample b of v[b*rows + r][c] (ld): one workgroup per (sample, 64-channel slab); 16 row
// lanes x 64 channels with four loads in flight per lane, fixed order
template <typename T>
__global__ __launch_bounds__(1024) void colsum_samples_kernel(const T* __restrict__ v, long ld, int rows, int C, float* __restrict__ out,
                                                              long ldo) {
  __shared__ float s[16][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, rl = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (c < C) {
    const T* p = v + (long)b * rows * ld + c;
    int r = rl;
    for (; r + 48 < rows; r += 64) {
      a0 += ldf(p + (long)r * ld);
      a1 += ldf(p + (long)(r + 16) * ld);
      a2 += ldf(p + (long)(r + 32) * ld);
      a3 += ldf(p + (long)(r + 48) * ld);
    }
    for (; r < rows; r += 16) a0 += ldf(p + (long)r * ld);
  }
  s[rl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += s[i][cl];
    out[(long)b * ldo + c] = t;
  }
}

// LayerNorm backward over rows of C <= 64 * LN_NC channels: one wave per row, lanes own columns lane + 64 k.
// dx[r][c] (+)= rstd (dy gamma - mean(dy gamma) - xhat mean(dy gamma xhat)); per-workgroup partial sums of dy xhat / dy go
// to part[blockIdx][2][C] (summed by sum_rows_add).
constexpr int LN_NC = 20;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, long ld, const float* __restrict__ dy, long ldy, int rows,
                                                     int C, const float* __restrict__ gamma, float eps, float* __restrict__ dx, long lddx,
                                                     int accum, float* __restrict__ part) {
  __shared__ float s_g[4][64 * LN_NC / 4];  // reused per quarter below (see the reduction)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dg[LN_NC], db[LN_NC];
#pragma unroll
  for (int k = 0; k < LN_NC; ++k) dg[k] = db[k] = 0.f;
  const float invC = 1.f / (float)C;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const float* xr = x + (long)r * ld;
    const float* dr = dy + (long)r * ldy;
    float xv[LN_NC], dv[LN_NC];
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NC; ++k) {
      const int c = lane + 64 * k;
      xv[k] = c < C ? xr[c] : 0.f;
      dv[k] = c < C ? dr[c] : 0.f;
      a += xv[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    const float mean = a * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NC; ++k) {
      const int c = lane + 64 * k;
      const float d = c < C ? xv[k] - mean : 0.f;
      q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q * invC + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NC; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float xh = (xv[k] - mean) * rstd, dxh = dv[k] * gamma[c];
        s1 += dxh;
        s2 += dxh * xh;
        dg[k] += dv[k] * xh;
        db[k] += dv[k];
        xv[k] = xh;
        dv[k] = dxh;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    const float m1 = s1 * invC, m2 = s2 * invC;
    float* dxr = dx + (long)r * lddx;
#pragma unroll
    for (int k = 0; k < LN_NC; ++k) {
      const int c = lane + 64 * k;
      if (c < C) {
        const float v = rstd * (dv[k] - m1 - xv[k] * m2);
        dxr[c] = accum ? dxr[c] + v : v;
      }
    }
  }
  // the four waves' partials, added in wave order, one quarter of the columns at a time (LDS budget)
  float* pg = part + (long)blockIdx.x * 2 * C;
  for (int which = 0; which < 2; ++which) {
#pragma unroll
    for (int k0 = 0; k0 < LN_NC; k0 += LN_NC / 4) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < LN_NC / 4; ++k) s_g[wave][k * 64 + lane] = which ? db[k0 + k] : dg[k0 + k];
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int k = 0; k < LN_NC / 4; ++k) {
          const int c = lane + 64 * (k0 + k);
          if (c < C) pg[(long)which * C + c] = (s_g[0][k * 64 + lane] + s_g[1][k * 64 + lane]) + (s_g[2][k * 64 + lane] + s_g[3][k * 64 + lane]);
        }
      }
    }
  }
}

// GEGLU backward (modules/attention.py:37-45, exact-erf GELU).  pre [rows][N] fp16 = the FF1 pre-activations in the PACKED
// column order of the forward GEMM (64-column blocks: 32 value columns, then their 32 gate columns); dgg [rows][N/2] fp32 =
// dL/d(value * gelu(gate)).  dpre [rows][N] fp16 in the same packed order.
__global__ void geglu_bwd_kernel(const half_t* __restrict__ pre, const float* __restrict__ dgg, long ldg, long rows, int N,
                                 half_t* __restrict__ dpre) {
  const int Nh = N / 2;
  const long total = rows * Nh;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Nh);
    const long r = i / Nh;
    const int blk = j >> 5, wi = j & 31;
    const long pv = r * N + blk * 64 + wi, pg = pv + 32;
    const float v = (float)pre[pv], g = (float)pre[pg], d = dgg[r * ldg + j];
    const float cdf = 0.5f * (1.f + erff(g * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * g * g);
    dpre[pv] = (half_t)(d * g * cdf);
    dpre[pg] = (half_t)(d * v * (cdf + g * pdf));
  }
}

// dst[perm(p)][c] += src[p][c]: un-does the GEGLU row interleave for the FF1 weight ([N][C]) / bias ([N][1]) gradients
__global__ void geglu_unpermute_add_kernel(const float* __restrict__ src, int N, int C, float* __restrict__ dst) {
  const long total = (long)N * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C), nd = (int)(i / C);
    const int j = nd >> 6, wi = nd & 63;
    const int n = wi < 32 ? 32 * j + wi : N / 2 + 32 * j + (wi - 32);
    dst[(long)n * C + c] += src[i];
  }
}

// out[r][c] = (accum ? out : 0) + a[r][c] (+ b[r][c]); row strides in elements
__global__ void add_views_kernel(float* __restrict__ out, long ldo, const float* __restrict__ a, long lda, const float* __restrict__ b,
                                 long ldb, long rows, int C, int accum) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    float v = a[r * lda + c];
    if (b) v += b[r * ldb + c];
    if (accum) v += out[r * ldo + c];
    out[r * ldo + c] = v;
  }
}

// adjoint of nearest x2 upsampling: dx[b,y,x,c] (+)= sum of the 2x2 block of dup [B,2H,2W,C]
__global__ void upsample2_bwd_kernel(const float* __restrict__ dup, int B, int H, int W, int C, float* __restrict__ dx, long lddx, int accum) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    const float* q = dup + (((long)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c;
    const float v = (q[0] + q[C]) + (q[(long)2 * W * C] + q[(long)2 * W * C + C]);
    dx[p * lddx + c] = accum ? dx[p * lddx + c] + v : v;
  }
}

// adjoint of the 3x3 / stride s / pad 1 gather: dcol [R_out][9 * C] (column block t2 holds the contribution of weight tap
// 8 - t2: the dgrad weights are stored tap-flipped) -> dx[b,y,x,c] (+)= sum over taps (ky,kx) and outputs with yo*s + ky - 1 == y
__global__ void col2im3_bwd_kernel(const float* __restrict__ dcol, int B, int H, int W, int C, int stride, int Ho, int Wo,
                                   float* __restrict__ dx, long lddx, int accum) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long p = i / C;
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    float a = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int ny = y + 1 - tap / 3, nx = x + 1 - tap % 3;
      if (ny < 0 || nx < 0 || ny % stride || nx % stride) continue;
      const int yo = ny / stride, xo = nx / stride;
      if (yo >= Ho || xo >= Wo) continue;
      a += dcol[(((long)b * Ho + yo) * Wo + xo) * 9 * C + (8 - tap) * C + c];
    }
    dx[p * lddx + c] = accum ? dx[p * lddx + c] + a : a;
  }
}

// silu'(u) applied to a gradient in place: g *= silu'(u)
__global__ void silu_bwd_kernel(float* __restrict__ g, const float* __restrict__ u, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] *= act_d(u[i], ACT_SILU);
}

__global__ void silu_fwd_kernel(const float* __restrict__ u, float* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = act_f(u[i], ACT_SILU);
}

// d timestep_embedding is not needed (the time steps are data); the sinusoid itself is recomputed by the forward kernels.

// ---------------------------------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, the reference's optimiser: morphable_diffusion.py:627-646) on one contiguous range of
// the flat parameter arena.  g carries the loss scale: inv_scale un-does it.  `skip` (device flag, non-zero when a gradient
// was non-finite) leaves everything untouched.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                             float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2_sqrt, float inv_scale,
                             const int* __restrict__ skip) {
  if (skip && *skip) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * inv_scale;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}

// flag[0] = 1 if any element is inf / nan (flag must be zeroed first); sumsq[blockIdx] = partial sum of squares (may be null)
__global__ __launch_bounds__(256) void finite_check_kernel(const float* __restrict__ g, size_t n, int* __restrict__ flag) {
  int bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float x = g[i];
    if (!(fabsf(x) <= 3.0e38f)) bad = 1;
  }
  if (__builtin_amdgcn_ballot_w64(bad) && (threadIdx.x & 63) == 0) *flag = 1;  // benign race: every writer stores 1
}

// ---------------------------------------------------------------------------------------------------------------------
// Self-attention backward (CrossAttention.forward modules/attention.py:170-203, self-attention case), flash style on the
// matrix cores, in the transposed formulation of k_attn.hip (one query -- or, in the dK/dV kernel, one key -- per lane):
//   kernel 1 (per 128 queries):  lse_q, delta_q = dO_q . O_q;   dQ^T = K^T dS^T  with  dS^T = P^T o (V dO^T - delta)
//   kernel 2 (per 128 keys):     dV^T = dO^T P,   dK^T = Q^T dS   with  P = exp2(S scale - lse),  dS = P o (dO V^T - delta)
// qkv: [token][q | k | v] fp16 (row stride ld3, each block C = heads * D wide); o / dO: [token][C] fp16.
// dqkv: [token][dq | dk | dv] fp16.  lse / delta: [B][heads][T] fp32 scratch written by kernel 1, read by kernel 2.
template <int D>
struct AttnBwdCfg {
  static constexpr int KS = (D + 15) / 16, DK = KS * 16, DVF = (D + 31) / 32, DVP = DVF * 32;
  static constexpr int KLD = DK + 8, VLD = 64 + 8;
  static constexpr int ROWS_BYTES = 64 * KLD * 2, IMG_BYTES = DVP * VLD * 2;
};

// stage a 64-row tile of [token][D] (row stride ld, column offset col) as rows [64][KLD] and, optionally, as the transposed
// image [DVP][VLD]; rows >= T and columns >= D are zero
template <int D>
__device__ __forceinline__ void stage_tile(const half_t* __restrict__ base, long ld, int col, long tok0, int r0, int T, half_t* rowsL,
                                           half_t* imgL) {
  using Cf = AttnBwdCfg<D>;
  const int tid = threadIdx.x;
  constexpr int CH = Cf::DK / 8;
  for (int idx = tid; idx < 64 * CH; idx += 256) {
    const int key = idx / CH, ch = idx - key * CH;
    h8 v = (h8)(half_t)0;
    if (r0 + key < T && ch * 8 < D) v = *(const h8*)(base + (tok0 + r0 + key) * ld + col + ch * 8);
    if (rowsL) *(h8*)(rowsL + key * Cf::KLD + ch * 8) = v;
    if (imgL && ch * 8 < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) imgL[(ch * 8 + e) * Cf::VLD + key] = v[e];
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const half_t* __restrict__ qkv, int ld3, const half_t* __restrict__ o,
                                                          const half_t* __restrict__ dO, int ldo, half_t* __restrict__ dqkv, int ldd,
                                                          float* __restrict__ lse, float* __restrict__ delta, int T, int heads,
                                                          float scale_l2, float scale_nat) {
  using Cf = AttnBwdCfg<D>;
  constexpr int KS = Cf::KS, DVF = Cf::DVF, KLD = Cf::KLD, VLD = Cf::VLD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* sK = (half_t*)smem;
  half_t* sV = (half_t*)(smem + Cf::ROWS_BYTES);
  half_t* sKT = (half_t*)(smem + 2 * Cf::ROWS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, lq = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int C = heads * D;
  const int q_row = blockIdx.x * 128 + wave * 32 + lq;
  const long tok0 = (long)b * T;
  h8 qf[KS], dof[KS];
  float dl = 0.f;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = ks * 16 + hh * 8;
    qf[ks] = (h8)(half_t)0;
    dof[ks] = (h8)(half_t)0;
    if (q_row < T && d0 < D) {
      qf[ks] = *(const h8*)(qkv + (tok0 + q_row) * ld3 + head * D + d0);
      dof[ks] = *(const h8*)(dO + (tok0 + q_row) * ldo + head * D + d0);
      const h8 ov = *(const h8*)(o + (tok0 + q_row) * ldo + head * D + d0);
#pragma unroll
      for (int e = 0; e < 8; ++e) dl += (float)dof[ks][e] * (float)ov[e];
    }
  }
  dl += __shfl_xor(dl, 32);
  // pass A: softmax statistics of the query's row (online max / sum over the key tiles)
  float m_run = -INFINITY, l_run = 0.f;
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();
    stage_tile<D>(qkv, ld3, C + head * D, tok0, k0, T, sK, nullptr);
    __syncthreads();
    float mx = -INFINITY;
    f32x16 s[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h8 kf = *(const h8*)(sK + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        s[f] = MVD_MFMA_32x32x16(kf, qf[ks], s[f], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= T) s[f][r] = -INFINITY;
        mx = fmaxf(mx, s[f][r]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * scale_l2);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) psum += __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], scale_l2, -m_new));
    psum += __shfl_xor(psum, 32);
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + psum;
    m_run = m_new;
  }
  const float lse_q = m_run + __builtin_amdgcn_logf(l_run);  // v_log_f32 is log2
  if (q_row < T && hh == 0) {
    lse[((long)b * heads + head) * T + q_row] = lse_q;
    delta[((long)b * heads + head) * T + q_row] = dl;
  }
  // pass B
  f32x16 acc[DVF];
#pragma unroll
  for (int f = 0; f < DVF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
  for (int idx = tid; idx < (Cf::DVP - D) * 64; idx += 256) sKT[(D + idx / 64) * VLD + (idx & 63)] = (half_t)0;
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();
    stage_tile<D>(qkv, ld3, C + head * D, tok0, k0, T, sK, sKT);
    stage_tile<D>(qkv, ld3, 2 * C + head * D, tok0, k0, T, sV, nullptr);
    __syncthreads();
    f32x16 s[2], dp[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = dp[f][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h8 kf = *(const h8*)(sK + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        s[f] = MVD_MFMA_32x32x16(kf, qf[ks], s[f], 0, 0, 0);
        const h8 vf = *(const h8*)(sV + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        dp[f] = MVD_MFMA_32x32x16(vf, dof[ks], dp[f], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float p = key < T ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], scale_l2, -lse_q)) : 0.f;
        s[f][r] = p * (dp[f][r] - dl);  // dS^T
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      h8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (half_t)s[kk >> 1][8 * (kk & 1) + j];
#pragma unroll
      for (int f = 0; f < DVF; ++f) {
        const half_t* row = sKT + (f * 32 + lq) * VLD + kk * 16 + 4 * hh;
        const h4 v0 = *(const h4*)(row);
        const h4 v1 = *(const h4*)(row + 8);
        h8 va;
        va[0] = v0[0]; va[1] = v0[1]; va[2] = v0[2]; va[3] = v0[3];
        va[4] = v1[0]; va[5] = v1[1]; va[6] = v1[2]; va[7] = v1[3];
        acc[f] = MVD_MFMA_32x32x16(va, pb, acc[f], 0, 0, 0);
      }
    }
  }
  if (q_row < T) {
    half_t* orow = dqkv + (tok0 + q_row) * ldd + head * D;
#pragma unroll
    for (int f = 0; f < DVF; ++f)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int dv = f * 32 + 8 * rg + 4 * hh;
        if (dv < D) {
          h4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (half_t)(acc[f][rg * 4 + j] * scale_nat);
          *(h4*)(orow + dv) = v;
        }
      }
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const half_t* __restrict__ qkv, int ld3, const half_t* __restrict__ dO, int ldo,
                                                           half_t* __restrict__ dqkv, int ldd, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, int T, int heads, float scale_l2,
                                                           float scale_nat) {
  using Cf = AttnBwdCfg<D>;
  constexpr int KS = Cf::KS, DVF = Cf::DVF, KLD = Cf::KLD, VLD = Cf::VLD;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* sQ = (half_t*)smem;
  half_t* sdO = (half_t*)(smem + Cf::ROWS_BYTES);
  half_t* sQT = (half_t*)(smem + 2 * Cf::ROWS_BYTES);
  half_t* sdOT = (half_t*)(smem + 2 * Cf::ROWS_BYTES + Cf::IMG_BYTES);
  float* s_lse = (float*)(smem + 2 * Cf::ROWS_BYTES + 2 * Cf::IMG_BYTES);
  float* s_dl = s_lse + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, lq = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int C = heads * D;
  const int key = blockIdx.x * 128 + wave * 32 + lq;
  const long tok0 = (long)b * T;
  h8 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = ks * 16 + hh * 8;
    kf[ks] = (h8)(half_t)0;
    vf[ks] = (h8)(half_t)0;
    if (key < T && d0 < D) {
      kf[ks] = *(const h8*)(qkv + (tok0 + key) * ld3 + C + head * D + d0);
      vf[ks] = *(const h8*)(qkv + (tok0 + key) * ld3 + 2 * C + head * D + d0);
    }
  }
  f32x16 dk[DVF], dv[DVF];
#pragma unroll
  for (int f = 0; f < DVF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[f][r] = dv[f][r] = 0.f;
  for (int idx = tid; idx < (Cf::DVP - D) * 64; idx += 256) {
    sQT[(D + idx / 64) * VLD + (idx & 63)] = (half_t)0;
    sdOT[(D + idx / 64) * VLD + (idx & 63)] = (half_t)0;
  }
  const float* lse_b = lse + ((long)b * heads + head) * T;
  const float* dl_b = delta + ((long)b * heads + head) * T;
  for (int q0 = 0; q0 < T; q0 += 64) {
    __syncthreads();
    stage_tile<D>(qkv, ld3, head * D, tok0, q0, T, sQ, sQT);
    stage_tile<D>(dO, ldo, head * D, tok0, q0, T, sdO, sdOT);
    if (tid < 64) {
      s_lse[tid] = q0 + tid < T ? lse_b[q0 + tid] : 0.f;
      s_dl[tid] = q0 + tid < T ? dl_b[q0 + tid] : 0.f;
    }
    __syncthreads();
    f32x16 s[2], dp[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = dp[f][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h8 qa = *(const h8*)(sQ + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        s[f] = MVD_MFMA_32x32x16(qa, kf[ks], s[f], 0, 0, 0);
        const h8 da = *(const h8*)(sdO + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        dp[f] = MVD_MFMA_32x32x16(da, vf[ks], dp[f], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qi = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        const float p = q0 + qi < T ? __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], scale_l2, -s_lse[qi])) : 0.f;
        s[f][r] = p;                              // P
        dp[f][r] = p * (dp[f][r] - s_dl[qi]);     // dS
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      h8 pb, sb;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[j] = (half_t)s[kk >> 1][8 * (kk & 1) + j];
        sb[j] = (half_t)dp[kk >> 1][8 * (kk & 1) + j];
      }
#pragma unroll
      for (int f = 0; f < DVF; ++f) {
        const half_t* r1 = sdOT + (f * 32 + lq) * VLD + kk * 16 + 4 * hh;
        const h4 a0 = *(const h4*)(r1);
        const h4 a1 = *(const h4*)(r1 + 8);
        h8 va;
        va[0] = a0[0]; va[1] = a0[1]; va[2] = a0[2]; va[3] = a0[3];
        va[4] = a1[0]; va[5] = a1[1]; va[6] = a1[2]; va[7] = a1[3];
        dv[f] = MVD_MFMA_32x32x16(va, pb, dv[f], 0, 0, 0);
        const half_t* r2 = sQT + (f * 32 + lq) * VLD + kk * 16 + 4 * hh;
        const h4 b0 = *(const h4*)(r2);
        const h4 b1 = *(const h4*)(r2 + 8);
        h8 vb;
        vb[0] = b0[0]; vb[1] = b0[1]; vb[2] = b0[2]; vb[3] = b0[3];
        vb[4] = b1[0]; vb[5] = b1[1]; vb[6] = b1[2]; vb[7] = b1[3];
        dk[f] = MVD_MFMA_32x32x16(vb, sb, dk[f], 0, 0, 0);
      }
    }
  }
  if (key < T) {
    half_t* krow = dqkv + (tok0 + key) * ldd + C + head * D;
    half_t* vrow = dqkv + (tok0 + key) * ldd + 2 * C + head * D;
#pragma unroll
    for (int f = 0; f < DVF; ++f)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int d0 = f * 32 + 8 * rg + 4 * hh;
        if (d0 < D) {
          h4 a, c2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a[j] = (half_t)(dk[f][rg * 4 + j] * scale_nat);
            c2[j] = (half_t)dv[f][rg * 4 + j];
          }
          *(h4*)(krow + d0) = a;
          *(h4*)(vrow + d0) = c2;
        }
      }
  }
}

template <int D>
int launch_attn_bwd_t(const half_t* qkv, int ld3, const half_t* o, const half_t* dO, int ldo, half_t* dqkv, int ldd, float* lse,
                      float* delta, int B, int T, int heads, hipStream_t s) {
  using Cf = AttnBwdCfg<D>;
  constexpr int LDS1 = 2 * Cf::ROWS_BYTES + Cf::IMG_BYTES, LDS2 = 2 * Cf::ROWS_BYTES + 2 * Cf::IMG_BYTES + 512;
  static_assert(LDS2 <= 160 * 1024, "LDS budget");
  static bool attr_done[MVD_MAX_DEVICES] = {false};
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2));
    attr_set = true;
  }
  const float scale_nat = 1.0f / sqrtf((float)D), scale_l2 = 1.4426950408889634f * scale_nat;
  dim3 grid(cdiv(T, 128), heads, B);
  hipLaunchKernelGGL(attn_bwd_dq_kernel<D>, grid, dim3(256), LDS1, s, qkv, ld3, o, dO, ldo, dqkv, ldd, lse, delta, T, heads, scale_l2,
                     scale_nat);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel<D>, grid, dim3(256), LDS2, s, qkv, ld3, dO, ldo, dqkv, ldd, lse, delta, T, heads, scale_l2,
                     scale_nat);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// ---- launchers -------------------------------------------------------------------------------------------------------
int bwd_tcast(const void* src, int src_f32, long ld, int R, int C, half_t* dst, int Rp, hipStream_t s, int split, half_t* rows16, int Cp) {
  if (R <= 0 || C <= 0 || Rp < R) return mvd_fail("bwd_tcast: bad shape");
  if (split && !src_f32) return mvd_fail("bwd_tcast: the split layout needs an fp32 source");
  if (rows16 && (Cp < C || Cp > 64 * cdiv(C, 64))) return mvd_fail("bwd_tcast: bad row-image width");
  dim3 grid(cdiv(Rp, 64), cdiv(C, 64));
  static const bool scalar_only = getenv("MVD_STAGE_SCALAR") != nullptr;  // A/B switch: the element-per-thread staging kernels
  const bool vec = !scalar_only && !(C & 3) && !(ld & 3) && !(Rp & 63) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15) &&
                   (!rows16 || (!(Cp & 3) && !((uintptr_t)rows16 & 7)));
  if (vec) {
    if (src_f32) hipLaunchKernelGGL(tcast_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, R, C, dst, Rp, split, rows16, Cp);
    else hipLaunchKernelGGL(tcast_vec_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, R, C, dst, Rp, 0, rows16, Cp);
  } else if (src_f32) hipLaunchKernelGGL(tcast_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, R, C, dst, Rp, split, rows16, Cp);
  else hipLaunchKernelGGL(tcast_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, R, C, dst, Rp, 0, rows16, Cp);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_im2colT(const void* src, int src_f32, long ld, int B, int H, int W, int C, int stride, int ups, half_t* dst, int Rp, hipStream_t s) {
  const int Ho = ((H << ups) - 1) / stride + 1, Wo = ((W << ups) - 1) / stride + 1;
  if (Rp < B * Ho * Wo) return mvd_fail("bwd_im2colT: bad shape");
  dim3 grid(cdiv(Rp, 64), cdiv(C, 64), 9);
  static const bool scalar_only = getenv("MVD_STAGE_SCALAR") != nullptr;
  if (!scalar_only && !(C & 3) && !(ld & 3) && !(Rp & 63) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15)) {
    if (src_f32) hipLaunchKernelGGL(im2colT_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, B, H, W, C, stride, ups, Ho, Wo, dst, Rp);
    else hipLaunchKernelGGL(im2colT_vec_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, B, H, W, C, stride, ups, Ho, Wo, dst, Rp);
  } else if (src_f32) hipLaunchKernelGGL(im2colT_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, B, H, W, C, stride, ups, Ho, Wo, dst, Rp);
  else hipLaunchKernelGGL(im2colT_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, B, H, W, C, stride, ups, Ho, Wo, dst, Rp);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_cast_rows(const float* src, long ld, long rows, int C, int Cp, half_t* dst, hipStream_t s, int split) {
  static const bool scalar_only = getenv("MVD_STAGE_SCALAR") != nullptr;
  if (!scalar_only && !(C & 3) && !(Cp & 3) && !(ld & 3) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 7))
    hipLaunchKernelGGL(cast_rows_vec_kernel, dim3(gridn((size_t)rows * (Cp >> 2))), dim3(256), 0, s, src, ld, rows, C, Cp, dst, split);
  else
    hipLaunchKernelGGL(cast_rows_kernel, dim3(gridn((size_t)rows * Cp)), dim3(256), 0, s, src, ld, rows, C, Cp, dst, split);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_pack_dgrad(const half_t* w, int taps, int N, int ldw, int Cl, int Np, half_t* wT, hipStream_t s, int flip) {
  static const bool scalar_only = getenv("MVD_STAGE_SCALAR") != nullptr;
  if (!scalar_only && !(ldw & 7) && !(Np & 7) && !(Cl & 7) && !((uintptr_t)w & 15) && !((uintptr_t)wT & 15))
    hipLaunchKernelGGL(pack_dgrad_vec_kernel, dim3(cdiv(Np, 64), cdiv(Cl, 64), taps), dim3(256), 0, s, w, taps, N, ldw, Cl, Np, wT, flip);
  else
    hipLaunchKernelGGL(pack_dgrad_kernel, dim3(cdiv(Np, 64), cdiv(Cl, 64), taps), dim3(256), 0, s, w, taps, N, ldw, Cl, Np, wT, flip);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_group_norm(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                   const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* dg_part,
                   float* db_part, float* dpre_part, hipStream_t s) {
  if (C % G || C / G > 256) return mvd_fail("bwd_group_norm: channels per group must divide C and be <= 256");
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * G), dim3(256), 0, s, x, ld, pre, pld, dy, ldy, rows, C, G, gamma, beta, eps, act, dx, lddx,
                     accum, dg_part, db_part, dpre_part);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// slab plan: S row slabs per (sample, group) so that B * G * S is about 1024 workgroups, at least 8 rows per slab
int bwd_gn_slabs(int B, int G, int rows) {
  int S = 1024 / (B * G);
  if (S > rows / 8) S = rows / 8;
  if (S > 64) S = 64;
  return S < 1 ? 1 : S;
}
// GroupNorm forward in fp32 (x [B][rows][ld] -> y [B][rows][ldy]); part: [B*G*S*2] floats of scratch
int bwd_group_norm_fwd(const float* x, long ld, int B, int rows, int C, int G, const float* gamma, const float* beta, float eps, int act,
                       float* y, long ldy, float* part, int S, hipStream_t s) {
  if (C % G || C / G > 256) return mvd_fail("bwd_group_norm_fwd: channels per group must divide C and be <= 256");
  const int rps = cdiv(rows, S);
  hipLaunchKernelGGL(gn_slab_stats_kernel, dim3(B * G, S), dim3(256), 0, s, x, ld, (const float*)nullptr, 0, rows, C, G, S, rps, part);
  hipLaunchKernelGGL(gn_slab_fwd_kernel, dim3(B * G, S), dim3(256), 0, s, x, ld, part, rows, C, G, S, rps, gamma, beta, eps, act, y, ldy);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// slabbed backward: part / part2 [B*G*S*2], dg_part / db_part / dpre_part [B*S][C] (sum over the B*S rows with bwd_sum_rows_add)
int bwd_group_norm_slab(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                        const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* part,
                        float* part2, float* dg_part, float* db_part, float* dpre_part, int S, hipStream_t s) {
  if (C % G || C / G > 256) return mvd_fail("bwd_group_norm: channels per group must divide C and be <= 256");
  const int rps = cdiv(rows, S);
  dim3 grid(B * G, S);
  hipLaunchKernelGGL(gn_slab_stats_kernel, grid, dim3(256), 0, s, x, ld, pre, pld, rows, C, G, S, rps, part);
  hipLaunchKernelGGL(gn_slab_sums_kernel, grid, dim3(256), 0, s, x, ld, pre, pld, dy, ldy, part, rows, C, G, S, rps, gamma, beta, eps, act,
                     part2, dg_part, db_part);
  hipLaunchKernelGGL(gn_slab_apply_kernel, grid, dim3(256), 0, s, x, ld, pre, pld, dy, ldy, part, part2, rows, C, G, S, rps, gamma, beta,
                     eps, act, dx, lddx, accum, dpre_part);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_outer_add(const float* a, long lda, const float* b, long ldb, float* C, int M, int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(outer_add_kernel, dim3(gridn((size_t)M * N)), dim3(256), 0, s, a, lda, b, ldb, C, M, N, K);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_sum_rows_add(const float* part, int R, int C, long ldp, float* out, int accum, hipStream_t s) {
  SumJobs j{};
  j.part[0] = part, j.out[0] = out, j.ldp[0] = ldp, j.R[0] = R, j.C[0] = C, j.accum[0] = accum;
  hipLaunchKernelGGL(sum_rows_add_kernel, dim3(cdiv(C, 8), 1), dim3(256), 0, s, j);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// n <= 4 column sums in one launch: out[k][c] (+)= sum_r part[k][r * ldp[k] + c]
int bwd_sum_rows_multi(int n, const float* const* part, const int* R, const int* C, const long* ldp, float* const* out, const int* accum,
                       hipStream_t s) {
  if (n < 1 || n > 4) return mvd_fail("bwd_sum_rows_multi: 1..4 jobs");
  SumJobs j{};
  int cmax = 0;
  for (int k = 0; k < n; ++k) {
    j.part[k] = part[k], j.out[k] = out[k], j.ldp[k] = ldp[k], j.R[k] = R[k], j.C[k] = C[k], j.accum[k] = accum[k];
    cmax = C[k] > cmax ? C[k] : cmax;
  }
  hipLaunchKernelGGL(sum_rows_add_kernel, dim3(cdiv(cmax, 8), n), dim3(256), 0, s, j);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_colsum_samples(const void* v, int v_f32, long ld, int B, int rows, int C, float* out, long ldo, hipStream_t s) {
  if (v_f32) hipLaunchKernelGGL(colsum_samples_kernel<float>, dim3(cdiv(C, 64), B), dim3(1024), 0, s, (const float*)v, ld, rows, C, out, ldo);
  else hipLaunchKernelGGL(colsum_samples_kernel<half_t>, dim3(cdiv(C, 64), B), dim3(1024), 0, s, (const half_t*)v, ld, rows, C, out, ldo);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_ln_max_blocks() { return 256; }
// part: [min(256, ceil(rows/4))][2][C] floats; returns the number of partial rows through *nblk
int bwd_layer_norm(const float* x, long ld, const float* dy, long ldy, int rows, int C, const float* gamma, float eps, float* dx,
                   long lddx, int accum, float* part, int* nblk, hipStream_t s) {
  if (C > 64 * LN_NC) return mvd_fail("bwd_layer_norm: C too large");
  int nb = cdiv(rows, 4);
  if (nb > bwd_ln_max_blocks()) nb = bwd_ln_max_blocks();
  *nblk = nb;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(256), 0, s, x, ld, dy, ldy, rows, C, gamma, eps, dx, lddx, accum, part);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_geglu(const half_t* pre, const float* dgg, long ldg, long rows, int N, half_t* dpre, hipStream_t s) {
  if (N % 64) return mvd_fail("bwd_geglu: N must be a multiple of 64");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(gridn((size_t)rows * N / 2)), dim3(256), 0, s, pre, dgg, ldg, rows, N, dpre);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_geglu_unpermute_add(const float* src, int N, int C, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(geglu_unpermute_add_kernel, dim3(gridn((size_t)N * C)), dim3(256), 0, s, src, N, C, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_add_views(float* out, long ldo, const float* a, long lda, const float* b, long ldb, long rows, int C, int accum, hipStream_t s) {
  hipLaunchKernelGGL(add_views_kernel, dim3(gridn((size_t)rows * C)), dim3(256), 0, s, out, ldo, a, lda, b, ldb, rows, C, accum);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_upsample2(const float* dup, int B, int H, int W, int C, float* dx, long lddx, int accum, hipStream_t s) {
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(gridn((size_t)B * H * W * C)), dim3(256), 0, s, dup, B, H, W, C, dx, lddx, accum);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_col2im3(const float* dcol, int B, int H, int W, int C, int stride, float* dx, long lddx, int accum, hipStream_t s) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  hipLaunchKernelGGL(col2im3_bwd_kernel, dim3(gridn((size_t)B * H * W * C)), dim3(256), 0, s, dcol, B, H, W, C, stride, Ho, Wo, dx, lddx, accum);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_silu_inplace(float* g, const float* u, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(gridn(n)), dim3(256), 0, s, g, u, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_silu_fwd(const float* u, float* out, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(silu_fwd_kernel, dim3(gridn(n)), dim3(256), 0, s, u, out, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd, int step,
              float inv_scale, const int* skip, hipStream_t s) {
  if (!n) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(gridn(n, 16384)), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2),
                     inv_scale, skip);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int bwd_finite_check(const float* g, size_t n, int* flag, hipStream_t s) {
  if (!n) return 0;
  hipLaunchKernelGGL(finite_check_kernel, dim3(gridn(n, 4096)), dim3(256), 0, s, g, n, flag);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// qkv [B*T][3C] (ld3), o / dO [B*T][C] (ldo) fp16 -> dqkv [B*T][3C] (ldd) fp16; lse / delta: [B*heads*T] floats scratch each
int bwd_attention(const half_t* qkv, int ld3, const half_t* o, const half_t* dO, int ldo, half_t* dqkv, int ldd, float* lse, float* delta,
                  int B, int T, int heads, int d, hipStream_t s) {
  if (ld3 % 8 || ldo % 8 || ldd % 4 || d % 8 || ((uintptr_t)qkv & 15) || ((uintptr_t)o & 15) || ((uintptr_t)dO & 15) || ((uintptr_t)dqkv & 7))
    return mvd_fail("bwd_attention: alignment");
#define MVD_ATTN_B(DD) \
  case DD: return launch_attn_bwd_t<DD>(qkv, ld3, o, dO, ldo, dqkv, ldd, lse, delta, B, T, heads, s);
  switch (d) {
    MVD_ATTN_B(8)
    MVD_ATTN_B(16)
    MVD_ATTN_B(32)
    MVD_ATTN_B(40)
    MVD_ATTN_B(64)
    MVD_ATTN_B(80)
    MVD_ATTN_B(160)
    default: return mvd_fail("bwd_attention: unsupported head dim");
  }
#undef MVD_ATTN_B
}
