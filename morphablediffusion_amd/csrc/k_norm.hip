// GroupNorm (two kernels: coalesced slab statistics -> normalise + activation + fp16 cast) and LayerNorm.
// Both are HBM-bound: one fp32 read for the statistics, one fp32 read + one fp16 write for the apply.
// The fp16 output exists only as an MFMA operand of the following implicit GEMM.
#include "common.h"

namespace {

constexpr int GN_MAX_SLABS = 64;
constexpr int GN_CPT = 10;  // channels per thread: supports C <= 2560 with 256 threads

// grid (nslabs, B). Thread t owns channels t, t+256, ...; rows of the slab are walked sequentially so every
// wave-level load is a contiguous run of channels (coalesced).  Per-slab (sum, sumsq) per group are
// written to partial[b][slab][g][2]; the apply kernel combines the slabs in fp64.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ld, int rows_per_sample, int C,
                                                       int G, const float* __restrict__ preadd, int rows_per_slab,
                                                       float* __restrict__ partial) {
  __shared__ float s_sum[GN_CPT * 256];
  __shared__ float s_sq[GN_CPT * 256];
  const int b = blockIdx.y, slab = blockIdx.x, t = threadIdx.x;
  const int r_beg = slab * rows_per_slab;
  const int r_end = min(rows_per_sample, r_beg + rows_per_slab);
  // narrow tensors (C < 256, C | 256): several rows in flight per block so all 256 threads load
  const int row_par = (C < 256 && (256 % C) == 0) ? 256 / C : 1;
  const int lanes_c = C < 256 ? C : 256;
  const bool active = t < lanes_c * row_par;
  const int c0 = t % lanes_c, rsub = t / lanes_c;
  float sum[GN_CPT], sq[GN_CPT], pa[GN_CPT];
#pragma unroll
  for (int i = 0; i < GN_CPT; ++i) {
    sum[i] = 0.f;
    sq[i] = 0.f;
    const int c = c0 + 256 * i;
    pa[i] = (preadd && c < C) ? preadd[(long)b * C + c] : 0.f;
  }
  const float* xb = x + (long)b * rows_per_sample * ld;
  if (active) {
    for (int r = r_beg + rsub; r < r_end; r += row_par) {
      const float* xr = xb + (long)r * ld;
#pragma unroll
      for (int i = 0; i < GN_CPT; ++i) {
        const int c = c0 + 256 * i;
        if (c < C) {
          const float v = xr[c] + pa[i];
          sum[i] += v;
          sq[i] += v * v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < GN_CPT; ++i) {
      const int c = c0 + 256 * i;
      if (c < C) {
        s_sum[rsub * C + c] = sum[i];
        s_sq[rsub * C + c] = sq[i];
      }
    }
  }
  __syncthreads();
  if (t < G) {
    const int cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int rs = 0; rs < row_par; ++rs)
      for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
        a += s_sum[rs * C + c];
        q += s_sq[rs * C + c];
      }
    float* p = partial + (((long)b * gridDim.x + slab) * G + t) * 2;
    p[0] = a;
    p[1] = q;
  }
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + __expf(-v));
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// grid (blocks_per_sample, B); each thread handles 4 consecutive channels of one row per iteration.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ld, int rows_per_sample, int C,
                                                       int G, const float* __restrict__ preadd,
                                                       const float* __restrict__ partial, int nslabs,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int act, half_t* __restrict__ out, int ldo) {
  __shared__ float s_mean[32], s_rstd[32];
  const int b = blockIdx.y, t = threadIdx.x;
  if (t < G) {
    double a = 0.0, q = 0.0;
    for (int s = 0; s < nslabs; ++s) {
      const float* p = partial + (((long)b * nslabs + s) * G + t) * 2;
      a += (double)p[0];
      q += (double)p[1];
    }
    const double n = (double)rows_per_sample * (double)(C / G);
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    s_mean[t] = (float)mean;
    s_rstd[t] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const int cpg = C / G;
  const int c4 = C >> 2;
  const long total = (long)rows_per_sample * c4;
  const float* xb = x + (long)b * rows_per_sample * ld;
  half_t* ob = out + (long)b * rows_per_sample * ldo;
  for (long idx = (long)blockIdx.x * 256 + t; idx < total; idx += (long)gridDim.x * 256) {
    const int r = (int)(idx / c4);
    const int c = (int)(idx - (long)r * c4) * 4;
    const float4 v = *(const float4*)(xb + (long)r * ld + c);
    float vv[4] = {v.x, v.y, v.z, v.w};
    h4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c + j;
      const int g = cc / cpg;
      float u = vv[j];
      if (preadd) u += preadd[(long)b * C + cc];
      u = (u - s_mean[g]) * s_rstd[g] * gamma[cc] + beta[cc];
      o[j] = (half_t)act_apply(u, act);
    }
    *(h4*)(ob + (long)r * ldo + c) = o;
  }
}

// one wave per row
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int rows, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, half_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * C;
  constexpr int MAXI = 24;  // C <= 1536
  float v[MAXI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < C ? xr[c] : 0.f;
    s += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    q += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
  half_t* orow = out + (long)row * C;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = lane + 64 * i;
    if (c < C) orow[c] = (half_t)((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

}  // namespace

int gn_max_slabs() { return GN_MAX_SLABS; }

int launch_gn_stats(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd,
                    float* partial, int* nslabs_out, hipStream_t s) {
  if (C > GN_CPT * 256 || G > 32 || C % G) return mvd_fail("gn_stats: unsupported channel/group count");
  int nslabs = rows_per_sample / 16;
  if (nslabs < 1) nslabs = 1;
  if (nslabs > GN_MAX_SLABS) nslabs = GN_MAX_SLABS;
  const int rps = cdiv(rows_per_sample, nslabs);
  nslabs = cdiv(rows_per_sample, rps);
  *nslabs_out = nslabs;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nslabs, B), dim3(256), 0, s, x, ld, rows_per_sample, C, G, preadd, rps,
                     partial);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_gn_apply(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd,
                    const float* partial, int nslabs, const float* gamma, const float* beta, float eps, int act,
                    half_t* out, int ldo, hipStream_t s) {
  if (C % 4 || ld % 4 || ldo % 4) return mvd_fail("gn_apply: channel counts must be multiples of 4");
  long total = (long)rows_per_sample * (C / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 512) blocks = 512;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, s, x, ld, rows_per_sample, C, G, preadd, partial,
                     nslabs, gamma, beta, eps, act, out, ldo);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_layernorm(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, half_t* out,
                     hipStream_t s) {
  if (C > 64 * 24) return mvd_fail("layernorm: C too large");
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, rows, C, gamma, beta, eps, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
