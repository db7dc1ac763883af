// GroupNorm (two kernels: coalesced slab statistics -> normalise + activation + fp16 cast) and LayerNorm.
// Both are HBM-bound: one fp32 read for the statistics, one fp32 read + one fp16 write for the apply.
// The fp16 output exists only as an MFMA operand of the following implicit GEMM.
//
// Thread mapping (both GN kernels): a thread owns fixed channel quads (float4) and walks rows, so a wave
// always reads a contiguous run of channels (16 B per lane) and no per-element index arithmetic or
// division is needed; narrow tensors (C/4 < 256) put several rows in flight per block.
#include "common.h"

namespace {

constexpr int GN_MAX_SLABS = 64;
constexpr int GN_MAXC = 2560;
constexpr int GN_NQ = 3;  // channel quads per thread when C/4 > 256 (C <= 3072)
constexpr int GN_UNROLL = 4;
constexpr int GN_GROUP_MAXC = 256;  // channels per group the single-pass kernel keeps in LDS (gn_group_eligible)

struct GnMap {
  int Q, lanes_q, row_par, nq;
  bool active;
  int q0, rsub;
};

__device__ __forceinline__ GnMap gn_map(int C, int t) {
  GnMap m;
  m.Q = C >> 2;
  if (m.Q <= 256) {
    m.lanes_q = m.Q;
    m.row_par = 256 / m.Q;
    m.nq = 1;
  } else {
    m.lanes_q = 256;
    m.row_par = 1;
    m.nq = (m.Q + 255) >> 8;
  }
  m.active = t < m.lanes_q * m.row_par;
  m.q0 = t % m.lanes_q;
  m.rsub = t / m.lanes_q;
  return m;
}

// grid (nslabs, B): per-slab (sum, sumsq) per group -> partial[b][slab][g][2]; combined in fp64 by the apply.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ld, int rows_per_sample, int C,
                                                       int G, const float* __restrict__ preadd, int pld, int rows_per_slab,
                                                       float* __restrict__ partial) {
  __shared__ float s_sum[GN_MAXC];
  __shared__ float s_sq[GN_MAXC];
  const int b = blockIdx.y, slab = blockIdx.x, t = threadIdx.x;
  const int r_beg = slab * rows_per_slab;
  const int r_end = min(rows_per_sample, r_beg + rows_per_slab);
  const GnMap m = gn_map(C, t);
  float sum[GN_NQ][4], sq[GN_NQ][4], pa[GN_NQ][4];
#pragma unroll
  for (int i = 0; i < GN_NQ; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sum[i][e] = 0.f;
      sq[i][e] = 0.f;
      const int c = (m.q0 + 256 * i) * 4 + e;
      pa[i][e] = (preadd && i < m.nq && c < C) ? preadd[(long)b * pld + c] : 0.f;
    }
  const float* xb = x + (long)b * rows_per_sample * ld;
  if (m.active) {
    // GN_UNROLL independent 16-byte loads per thread and iteration (rows past the slab are clamped and masked)
    for (int r = r_beg + m.rsub; r < r_end; r += GN_UNROLL * m.row_par) {
#pragma unroll
      for (int i = 0; i < GN_NQ; ++i) {
        const int q = m.q0 + 256 * i;
        if (i < m.nq && q < m.Q) {
          float4 v[GN_UNROLL];
#pragma unroll
          for (int u = 0; u < GN_UNROLL; ++u) {
            const int ru = min(r + u * m.row_par, r_end - 1);
            v[u] = *(const float4*)(xb + (long)ru * ld + q * 4);
          }
#pragma unroll
          for (int u = 0; u < GN_UNROLL; ++u) {
            const float w = (r + u * m.row_par < r_end) ? 1.f : 0.f;
            const float vv[4] = {v[u].x + pa[i][0], v[u].y + pa[i][1], v[u].z + pa[i][2], v[u].w + pa[i][3]};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              sum[i][e] += w * vv[e];
              sq[i][e] += w * vv[e] * vv[e];
            }
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < GN_NQ; ++i) {
      const int q = m.q0 + 256 * i;
      if (i < m.nq && q < m.Q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s_sum[m.rsub * C + q * 4 + e] = sum[i][e];
          s_sq[m.rsub * C + q * 4 + e] = sq[i][e];
        }
      }
    }
  }
  __syncthreads();
  if (t < G) {
    const int cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int rs = 0; rs < m.row_par; ++rs)
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

// grid (blocks_per_sample, B).  y = act(x * scale[c] + shift[c]) with scale/shift precomputed in LDS.
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, int ld, int rows_per_sample, int C,
                                                       int G, const float* __restrict__ preadd, int pld,
                                                       const float* __restrict__ partial, int nslabs,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int act, half_t* __restrict__ out, int ldo, int split) {
  __shared__ float s_mean[32], s_rstd[32];
  __shared__ float s_scale[GN_MAXC], s_shift[GN_MAXC];
  __shared__ double s_pa[256], s_pq[256];
  const int b = blockIdx.y, t = threadIdx.x;
  // slab partials -> (mean, rstd) per group: 256/G threads per group, fp64 combine
  const int parts = 256 / G;
  {
    const int gi = t % G, p = t / G;
    double a = 0.0, q = 0.0;
    if (p < parts)
      for (int s = p; s < nslabs; s += parts) {
        const float2 v = *(const float2*)(partial + (((long)b * nslabs + s) * G + gi) * 2);
        a += (double)v.x;
        q += (double)v.y;
      }
    s_pa[t] = a;
    s_pq[t] = q;
  }
  // this thread's channels: issue the parameter loads before waiting on the reduction
  constexpr int CPT = GN_MAXC / 256;
  float ga[CPT], be[CPT], pa[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = t + 256 * i;
    const bool ok = c < C;
    ga[i] = ok ? gamma[c] : 0.f;
    be[i] = ok ? beta[c] : 0.f;
    pa[i] = (ok && preadd) ? preadd[(long)b * pld + c] : 0.f;
  }
  __syncthreads();
  if (t < G) {
    double a = 0.0, q = 0.0;
    for (int p = 0; p < parts; ++p) {
      a += s_pa[p * G + t];
      q += s_pq[p * G + t];
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
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = t + 256 * i;
    if (c < C) {
      const int gq = c / cpg;
      const float sc = s_rstd[gq] * ga[i];
      s_scale[c] = sc;
      s_shift[c] = be[i] + (pa[i] - s_mean[gq]) * sc;
    }
  }
  __syncthreads();
  const GnMap m = gn_map(C, t);
  if (!m.active) return;
  const int rows_per_block = (rows_per_sample + gridDim.x - 1) / gridDim.x;
  const int r_beg = blockIdx.x * rows_per_block;
  const int r_end = min(rows_per_sample, r_beg + rows_per_block);
  const float* xb = x + (long)b * rows_per_sample * ld;
  half_t* ob = out + (long)b * rows_per_sample * ldo;
#pragma unroll
  for (int i = 0; i < GN_NQ; ++i) {
    const int q = m.q0 + 256 * i;
    if (i >= m.nq || q >= m.Q) continue;
    const float4 sc = *(const float4*)(s_scale + q * 4);
    const float4 sh = *(const float4*)(s_shift + q * 4);
    for (int r = r_beg + m.rsub; r < r_end; r += GN_UNROLL * m.row_par) {
      float4 v[GN_UNROLL];
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
        const int ru = min(r + u * m.row_par, r_end - 1);
        v[u] = *(const float4*)(xb + (long)ru * ld + q * 4);
      }
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
        const int ru = r + u * m.row_par;
        if (ru >= r_end) break;
        const float y0 = act_apply(v[u].x * sc.x + sh.x, act), y1 = act_apply(v[u].y * sc.y + sh.y, act);
        const float y2 = act_apply(v[u].z * sc.z + sh.z, act), y3 = act_apply(v[u].w * sc.w + sh.w, act);
        h4 o;
        o[0] = (half_t)y0; o[1] = (half_t)y1; o[2] = (half_t)y2; o[3] = (half_t)y3;
        half_t* op = ob + (long)ru * ldo + q * 4;
        *(h4*)op = o;
        if (split) {  // [hi | lo | hi]: the operand of an extended-precision consumer
          h4 lo;
          lo[0] = (half_t)(y0 - (float)o[0]); lo[1] = (half_t)(y1 - (float)o[1]);
          lo[2] = (half_t)(y2 - (float)o[2]); lo[3] = (half_t)(y3 - (float)o[3]);
          *(h4*)(op + C) = lo;
          *(h4*)(op + 2 * C) = o;
        }
      }
    }
  }
}

// ---- single-pass form for per-(sample, group) slices that fit in the registers of one workgroup -----------------
// grid = B * G workgroups; workgroup (b, g) owns the rows x (C/G) slice of its group: every thread issues all of its
// (<= MAXE) 8-byte loads up front, the slice stays in registers, mean and the centred second moment are two block
// reductions, and the normalised fp16 slice is written straight back.  x is read from HBM once (the two-pass form
// reads it twice and needs two launches).  Logical workgroup ids are laid out so the groups of one sample share an XCD.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* s_red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();  // s_red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) t += s_red[i];
  return t;
}

#ifdef MVD_TIMELINE
__device__ unsigned long long mvd_gn_tl[8 * 4096];
#define GTL(i)                                                                          \
  do {                                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 4096) mvd_gn_tl[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define GTL(i)
#endif
// (register tiles of up to 10 pairs are held to 64 VGPRs = 2048 resident threads per CU: the 1024-thread variant compiled to
// 70, i.e. one workgroup per CU and four rounds for the 1024 slices of a 32-sample batch)
// NS > 1: x is NS split-K slabs (slab_stride floats apart) of the producing conv; their sum + bias2[c] + the pre-add is the
// tensor to normalise (the conv's reduce pass never runs: GemmArgs::slabs).  NS == 0: the slab count is the run-time `nslab`
// (5 ... 16 slabs: the 4 x 4 level's convolutions split K twelve ways); the slabs are added in slab order either way.
// resid (optional, row stride ldr): added to the sum like the reduce pass's residual epilogue.  mat (optional, row stride
// ldm): the summed tensor is ALSO written there in fp32 -- the materialised output of the producing GEMM for its later
// readers (residual adds, skip connections), which makes this kernel the GEMM's reduce pass and the next block's first
// GroupNorm in one launch.
template <int NT, int MAXE, int NS>
__global__ __launch_bounds__(NT, (MAXE <= 10 ? 8 : 4)) void gn_group_kernel(const float* __restrict__ x, int ld, int rows, int C, int G,
                                                      const float* __restrict__ preadd, int pld,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, int act, half_t* __restrict__ out, int ldo, int split,
                                                      long slab_stride, const float* __restrict__ bias2, int nslab,
                                                      const float* __restrict__ resid, int ldr, float* __restrict__ mat, int ldm) {
  __shared__ float s_red[NT / 64];
  // the group's gain, bias and pre-add in LDS: as global loads inside the store loop (4 loads per 4-byte store, each
  // iteration waiting on its own) they made "normalise + store" 9 of the workgroup's 13.7 us (tools/gn_timeline.py)
  __shared__ __attribute__((aligned(8))) float s_par[3][GN_GROUP_MAXC];
  const int t = threadIdx.x;
  GTL(0);
  int wg = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, slot = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int b = wg / G, g = wg - b * G;
  const int cpg = C / G, h = cpg >> 1, n2 = rows * h;
  const float inv_h = 1.0f / (float)h;
  const float* xb = x + (long)b * rows * ld + g * cpg;
  half_t* ob = out + (long)b * rows * ldo + g * cpg;
  if (t < cpg) {
    s_par[0][t] = gamma[g * cpg + t];
    s_par[1][t] = beta[g * cpg + t];
    s_par[2][t] = (preadd ? preadd[(long)b * pld + g * cpg + t] : 0.f) + (bias2 ? bias2[g * cpg + t] : 0.f);
  }
  const bool has_pre = preadd || bias2;
  float2 v[MAXE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = t + i * NT;
    v[i] = make_float2(0.f, 0.f);
    if (e < n2) {
      const int row = (int)(((float)e + 0.5f) * inv_h), j = e - row * h;
      v[i] = *(const float2*)(xb + (long)row * ld + 2 * j);
      if constexpr (NS == 0) {
        // four slab loads in flight per step (a load -> add -> load chain would pay the L2 latency once per slab); the slabs
        // are still added in slab order
        const float* xs = xb + (long)row * ld + 2 * j;
        for (int s0 = 1; s0 < nslab; s0 += 4) {
          float2 q4[4];
#pragma unroll
          for (int u2 = 0; u2 < 4; ++u2) q4[u2] = *(const float2*)(xs + (long)min(s0 + u2, nslab - 1) * slab_stride);
#pragma unroll
          for (int u2 = 0; u2 < 4; ++u2)
            if (s0 + u2 < nslab) {
              v[i].x += q4[u2].x;
              v[i].y += q4[u2].y;
            }
        }
      } else {
#pragma unroll
        for (int sl = 1; sl < NS; ++sl) {
          const float2 q2 = *(const float2*)(xb + sl * slab_stride + (long)row * ld + 2 * j);
          v[i].x += q2.x;
          v[i].y += q2.y;
        }
      }
      if (resid) {
        const float2 q2 = *(const float2*)(resid + ((long)b * rows + row) * ldr + g * cpg + 2 * j);
        v[i].x += q2.x;
        v[i].y += q2.y;
      }
    }
  }
  if (has_pre) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int e = t + i * NT;
      if (e < n2) {
        const int row = (int)(((float)e + 0.5f) * inv_h), j = e - row * h;
        const float2 p = *(const float2*)(&s_par[2][2 * j]);
        v[i].x += p.x;
        v[i].y += p.y;
      }
    }
  }
  if (mat) {
    float* mb = mat + (long)b * rows * ldm + g * cpg;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
      const int e = t + i * NT;
      if (e < n2) {
        const int row = (int)(((float)e + 0.5f) * inv_h), j = e - row * h;
        *(float2*)(mb + (long)row * ldm + 2 * j) = v[i];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXE; ++i) s += v[i].x + v[i].y;
  const float n = (float)rows * (float)cpg;
  GTL(1);
  const float mean = block_sum<NT>(s, s_red) / n;
  GTL(2);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = t + i * NT;
    if (e < n2) {
      const float dx = v[i].x - mean, dy = v[i].y - mean;
      q += dx * dx + dy * dy;
    }
  }
  const float rstd = rsqrtf(block_sum<NT>(q, s_red) / n + eps);
  GTL(3);
  // (the block reductions above synchronised the workgroup after s_par was written)
  int t2 = t;
  asm volatile("" : "+v"(t2));  // re-derive (row, j) here instead of keeping MAXE pairs of them live across the reductions
#pragma unroll
  for (int i = 0; i < MAXE; ++i) {
    const int e = t2 + i * NT;
    if (e < n2) {
      const int row = (int)(((float)e + 0.5f) * inv_h), j = e - row * h;
      const float2 gm = *(const float2*)(&s_par[0][2 * j]), bt = *(const float2*)(&s_par[1][2 * j]);
      const float y0 = act_apply((v[i].x - mean) * rstd * gm.x + bt.x, act);
      const float y1 = act_apply((v[i].y - mean) * rstd * gm.y + bt.y, act);
      h2 o;
      o[0] = (half_t)y0;
      o[1] = (half_t)y1;
      half_t* op = ob + (long)row * ldo + 2 * j;
      if (split == 2) {  // fp32 output (the exact output head): `out` is a float buffer, ldo / 2 floats per row
        *(float2*)((float*)out + ((long)b * rows + row) * (ldo >> 1) + g * cpg + 2 * j) = make_float2(y0, y1);
        continue;
      }
      *(h2*)op = o;
      if (split) {  // [hi | lo | hi]: the operand of an extended-precision consumer
        h2 lo;
        lo[0] = (half_t)(y0 - (float)o[0]);
        lo[1] = (half_t)(y1 - (float)o[1]);
        *(h2*)(op + C) = lo;
        *(h2*)(op + 2 * C) = o;
      }
    }
  }
  GTL(4);
}

#ifdef MVD_TIMELINE
extern "C" int mvd_debug_gn_timeline(unsigned long long* host_out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mvd_gn_tl), (size_t)n * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(mvd_gn_tl)) != hipSuccess) return -1;
  return hipMemset(p, 0, sizeof(mvd_gn_tl)) == hipSuccess ? 0 : -1;
}
#endif

// tile / slab partials [B][nslabs][G][2] -> per-sample scale[c] = rstd*gamma[c], shift[c] = beta[c] - mean*rstd*gamma[c]
// in global memory: a GroupNorm folded into the epilogue of the GEMM that re-computes its input (engine_unet: do_cond)
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nslabs, int rows_per_sample,
                                                          int C, int G, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ scale,
                                                          float* __restrict__ shift, int ld) {
  __shared__ double s_pa[256], s_pq[256];
  __shared__ float s_mean[32], s_rstd[32];
  const int b = blockIdx.x, t = threadIdx.x;
  const int parts = 256 / G, gi = t % G, p = t / G;
  double a = 0.0, q = 0.0;
  if (p < parts)
    for (int s = p; s < nslabs; s += parts) {
      const float2 v = *(const float2*)(partial + (((long)b * nslabs + s) * G + gi) * 2);
      a += (double)v.x;
      q += (double)v.y;
    }
  s_pa[t] = a;
  s_pq[t] = q;
  __syncthreads();
  if (t < G) {
    a = q = 0.0;
    for (int k = 0; k < parts; ++k) {
      a += s_pa[k * G + t];
      q += s_pq[k * G + t];
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
  for (int c = t; c < C; c += 256) {
    const float sc = s_rstd[c / cpg] * gamma[c];
    scale[(long)b * ld + c] = sc;
    shift[(long)b * ld + c] = beta[c] - s_mean[c / cpg] * sc;
  }
}

// one wave per row
template <typename OT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx, int rows, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, OT* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
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
  OT* orow = out + (long)row * C;
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int c = lane + 64 * i;
    if (c < C) orow[c] = (OT)((v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
}

// LayerNorm with 16-byte accesses: a row of C = 8 * L * NO floats is owned by L lanes (L = 8, 16, 32 or 64), each lane
// holding NO octets (32-byte loads, 16-byte fp16 stores; the scalar form above moves 4 bytes per lane and instruction and
// reaches ~2.5 TB/s); a wave normalises 64 / L rows at a time.  Mean, then the centred second moment, both reduced with
// xor shuffles inside the L-lane group.
// SLABS: x is `nslab` split-K slabs (slab_stride floats apart) of the producing GEMM; their sum + bias[c] + the per-sample bias
// rowbias[row / T][c] + the residual is the row to normalise, and that finished row is also written to `mat` in fp32 (the later
// residual adds read it) -- the GEMM's reduce pass and the LayerNorm in one launch (as gn_group_kernel does for GroupNorm).
struct LnSlabs {
  int nslab;
  long slab_stride;
  const float* bias;
  const float* rowbias;  // [B][rb_ld] or null
  int rb_ld, T;
  const float* resid;    // [rows][ldr] or null
  int ldr;
  float* mat;            // [rows][C]
  half_t* raw;           // optional: the row to normalise (before the norm) also as fp16, rows ld_raw halfs apart -- the second
  int ld_raw;            // operand block of a GEMM that takes [f(LN(x)) | x] (engine_unet.hip: the folded FF2 + proj_out)
};
template <int L, int NO, bool SLABS = false>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float* __restrict__ x, int rows, int C,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, half_t* __restrict__ out, const LnSlabs sl = LnSlabs()) {
  constexpr int RPW = 64 / L;  // rows per wave
  // gain and bias through LDS (loaded once per workgroup while the rows are in flight): as global loads inside the store
  // loop every iteration waited on its own
  __shared__ __attribute__((aligned(16))) float s_gb[2][8 * L * NO];
  for (int i = threadIdx.x; i < C; i += 256) {
    s_gb[0][i] = gamma[i];
    s_gb[1][i] = beta[i];
  }
  const int lane = threadIdx.x & 63, sub = lane & (L - 1);
  const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / L;
  const bool ok = row < rows;
  const float* xr = x + (long)(ok ? row : 0) * C;
  float v[NO][8];
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int c = (sub + L * i) * 8;
    const float4 a = *(const float4*)(xr + c), b = *(const float4*)(xr + c + 4);
    v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
  }
  if constexpr (SLABS) {
    const long r0 = ok ? row : 0;
    for (int sb = 1; sb < sl.nslab; ++sb) {  // slab order: the sum is the reduce pass's
#pragma unroll
      for (int i = 0; i < NO; ++i) {
        const float* p = x + (long)sb * sl.slab_stride + r0 * C + (sub + L * i) * 8;
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[i][0] += a.x; v[i][1] += a.y; v[i][2] += a.z; v[i][3] += a.w; v[i][4] += b.x; v[i][5] += b.y; v[i][6] += b.z; v[i][7] += b.w;
      }
    }
    const int smp = (int)(r0 / sl.T);
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      const int c = (sub + L * i) * 8;
      auto add8 = [&](const float* p) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[i][0] += a.x; v[i][1] += a.y; v[i][2] += a.z; v[i][3] += a.w; v[i][4] += b.x; v[i][5] += b.y; v[i][6] += b.z; v[i][7] += b.w;
      };
      // the order of the reduce pass's epilogue (igemm_epilogue.h epilogue_vec4): ((sum + bias) + per-sample bias) + residual --
      // the finished row has the bits the reduce pass would have written
      if (sl.bias) add8(sl.bias + c);
      if (sl.rowbias) add8(sl.rowbias + (long)smp * sl.rb_ld + c);
      if (sl.resid) add8(sl.resid + r0 * sl.ldr + c);
      if (ok) {
        float* m = sl.mat + r0 * C + c;
        *(float4*)m = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        *(float4*)(m + 4) = make_float4(v[i][4], v[i][5], v[i][6], v[i][7]);
      }
    }
  }
  if (sl.raw && ok) {
#pragma unroll
    for (int i = 0; i < NO; ++i) {
      h8 r;
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = (half_t)v[i][k];
      *(h8*)(sl.raw + (long)row * sl.ld_raw + (sub + L * i) * 8) = r;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[i][k];
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NO; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = v[i][k] - mean;
      q += d * d;
    }
#pragma unroll
  for (int o = L / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)C + eps);
  __syncthreads();
  if (!ok) return;
  half_t* orow = out + (long)row * C;
#pragma unroll
  for (int i = 0; i < NO; ++i) {
    const int c = (sub + L * i) * 8;
    const float4 g0 = *(const float4*)(&s_gb[0][c]), g1 = *(const float4*)(&s_gb[0][c + 4]);
    const float4 b0 = *(const float4*)(&s_gb[1][c]), b1 = *(const float4*)(&s_gb[1][c + 4]);
    const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    h8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (half_t)((v[i][k] - mean) * rstd * ga[k] + be[k]);
    *(h8*)(orow + c) = o;
  }
}

}  // namespace

int gn_max_slabs() { return GN_MAX_SLABS; }

int launch_gn_stats(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd, int pld,
                    float* partial, int* nslabs_out, hipStream_t s) {
  if (C > GN_MAXC || G > 32 || C % G || C % 4 || ld % 4) return mvd_fail("gn_stats: unsupported channel/group count");
  int nslabs = rows_per_sample / 16;
  if (nslabs < 1) nslabs = 1;
  if (nslabs > GN_MAX_SLABS) nslabs = GN_MAX_SLABS;
  // keep the whole chip busy when the batch is small, without shrinking slabs below a few rows per thread row
  while (nslabs * B < 256 && nslabs * 2 <= GN_MAX_SLABS && rows_per_sample / (nslabs * 2) >= 4) nslabs *= 2;
  const int rps = cdiv(rows_per_sample, nslabs);
  nslabs = cdiv(rows_per_sample, rps);
  *nslabs_out = nslabs;
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nslabs, B), dim3(256), 0, s, x, ld, rows_per_sample, C, G, preadd, pld ? pld : C,
                     rps, partial);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_gn_apply(const float* x, int ld, int B, int rows_per_sample, int C, int G, const float* preadd, int pld,
                    const float* partial, int nslabs, const float* gamma, const float* beta, float eps, int act,
                    half_t* out, int ldo, hipStream_t s, int split) {
  if (C > GN_MAXC || C % 4 || ld % 4 || ldo % 4) return mvd_fail("gn_apply: channel counts must be multiples of 4");
  if (split == 2) return mvd_fail("gn_apply: the fp32 output is written by the one-launch form only");
  if (split && ldo < 3 * C) return mvd_fail("gn_apply: a split output needs ldo >= 3C");
  int blocks = rows_per_sample / 8;
  if (blocks < 1) blocks = 1;
  const int cap = B >= 16 ? 64 : (B >= 4 ? 128 : 512);
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks, B), dim3(256), 0, s, x, ld, rows_per_sample, C, G, preadd, pld ? pld : C,
                     partial, nslabs, gamma, beta, eps, act, out, ldo, split);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// rows * C/G <= 32768 elements per (sample, group), even channels per group, 8-byte aligned slices
bool gn_group_eligible(int ld, int rows, int C, int G, int pld, int ldo) {
  if (C % G) return false;
  const int cpg = C / G;
  return (cpg % 2) == 0 && cpg <= GN_GROUP_MAXC && (ld % 2) == 0 && (ldo % 2) == 0 && (pld % 2) == 0 &&
         (long)rows * cpg <= 32768 && rows < (1 << 22);
}

int launch_gn_group(const float* x, int ld, int B, int rows, int C, int G, const float* preadd, int pld, const float* gamma,
                    const float* beta, float eps, int act, half_t* out, int ldo, hipStream_t s, int split, int nslab,
                    size_t slab_stride, const float* bias2, const float* resid, int ldr, float* mat, int ldm) {
  const int n2 = rows * (C / G) / 2;
  if (split == 1 && ldo < 3 * C) return mvd_fail("gn_group: a split output needs ldo >= 3C");
  if (split == 2 && (ldo < 2 * C || (ldo & 3) || ((uintptr_t)out & 7))) return mvd_fail("gn_group: an fp32 output needs ldo >= 2C halfs, 8-byte aligned rows");
  if (nslab < 1 || nslab > 64) return mvd_fail("gn_group: 1 to 64 slabs");
  if ((resid && (ldr & 1)) || (mat && (ldm & 1))) return mvd_fail("gn_group: residual / materialised rows must be 8-byte aligned");
  const dim3 grid(B * G);
#define MVD_GN1(NT, ME, NS_)                                                                                                     \
  hipLaunchKernelGGL((gn_group_kernel<NT, ME, NS_>), grid, dim3(NT), 0, s, x, ld, rows, C, G, preadd, pld, gamma, beta, eps, act, \
                     out, ldo, split, (long)slab_stride, bias2, nslab, resid, ldr, mat, ldm)
#define MVD_GN(NT, ME)                      \
  do {                                      \
    if (nslab == 1) MVD_GN1(NT, ME, 1);     \
    else if (nslab == 2) MVD_GN1(NT, ME, 2); \
    else if (nslab == 3) MVD_GN1(NT, ME, 3); \
    else if (nslab == 4) MVD_GN1(NT, ME, 4); \
    else MVD_GN1(NT, ME, 0);                \
  } while (0)
  // the smallest register tile that holds the group (unused slots still cost predicated loop iterations), 512-thread
  // workgroups up to 8192 pairs: swept on the UNet's shapes (tools/gn_bench.py), e.g. C=320 @32x32: 35 -> 27 us
  if (n2 <= 256 * 4) MVD_GN(256, 4);
  else if (n2 <= 512 * 4) MVD_GN(512, 4);
  else if (n2 <= 512 * 10) MVD_GN(512, 10);
  else if (n2 <= 512 * 16) MVD_GN(512, 16);
  else if (n2 <= 1024 * 10) MVD_GN(1024, 10);
  else MVD_GN(1024, 16);
#undef MVD_GN
#undef MVD_GN1
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_gn_finalize(const float* partial, int B, int nslabs, int rows_per_sample, int C, int G, const float* gamma,
                       const float* beta, float eps, float* scale, float* shift, int ld, hipStream_t s) {
  if (G > 32 || C % G) return mvd_fail("gn_finalize: unsupported channel/group count");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(256), 0, s, partial, nslabs, rows_per_sample, C, G, gamma, beta, eps,
                     scale, shift, ld);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_layernorm(const float* x, int rows, int C, const float* gamma, const float* beta, float eps, half_t* out,
                     hipStream_t s, half_t* raw, int ld_raw) {
  if (C > 64 * 24) return mvd_fail("layernorm: C too large");
  if (raw && (!layernorm_slabs_takes(C) || (ld_raw & 7) || ((uintptr_t)raw & 15) || (((uintptr_t)x | (uintptr_t)out) & 15)))
    return mvd_fail("layernorm: the fp16 copy of the input needs a width of 8 * L * 5 and 16-byte aligned rows");
  LnSlabs sl = LnSlabs();
  sl.raw = raw;
  sl.ld_raw = ld_raw;
  static const bool no_vec = getenv("MVD_LN_SCALAR") != nullptr;
  const bool aligned = !(((uintptr_t)x | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta) & 15);
  if (!no_vec && aligned && C % 40 == 0) {  // C = 8 * L * 5: the UNet's 320 / 640 / 1280 / 2560-wide rows
    const int L = C / 40;
#define MVD_LNV(L_) \
  hipLaunchKernelGGL((layernorm_vec_kernel<L_, 5>), dim3(cdiv(rows, 4 * (64 / L_))), dim3(256), 0, s, x, rows, C, gamma, beta, eps, out, sl)
    if (L == 8 || L == 16 || L == 32 || L == 64) {
      if (L == 8) MVD_LNV(8);
      else if (L == 16) MVD_LNV(16);
      else if (L == 32) MVD_LNV(32);
      else MVD_LNV(64);
      HIP_CHECK_RET(hipGetLastError());
      return 0;
    }
#undef MVD_LNV
  }
  if (raw) return mvd_fail("layernorm: the fp16 copy of the input is written by the vector form only (MVD_LN_SCALAR / alignment)");
  hipLaunchKernelGGL(layernorm_kernel<half_t>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, (long)C, rows, C, gamma, beta, eps, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// The split-K form: x = nslab slabs of [rows][C]; see LnSlabs.  C = 8 * L * 5 (the UNet's widths), 16-byte aligned operands.
bool layernorm_slabs_takes(int C) { return C % 40 == 0 && (C / 40 == 8 || C / 40 == 16 || C / 40 == 32 || C / 40 == 64); }
int launch_layernorm_slabs(const float* slabs, int nslab, size_t slab_stride, int rows, int C, const float* bias, const float* rowbias,
                           int rb_ld, int T, const float* resid, int ldr, float* mat, const float* gamma, const float* beta, float eps,
                           half_t* out, hipStream_t s, half_t* raw, int ld_raw) {
  if (raw && ((ld_raw & 7) || ((uintptr_t)raw & 15))) return mvd_fail("layernorm_slabs: the fp16 copy needs 16-byte aligned rows");
  if (!layernorm_slabs_takes(C) || nslab < 1 || T < 1 || !mat) return mvd_fail("layernorm_slabs: unsupported width / arguments");
  if ((((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)mat | (uintptr_t)bias | (uintptr_t)rowbias |
        (uintptr_t)resid) & 15) || (slab_stride & 3) || (rb_ld & 3) || (ldr & 3))
    return mvd_fail("layernorm_slabs: operands must be 16-byte aligned");
  LnSlabs sl;
  sl.nslab = nslab; sl.slab_stride = (long)slab_stride; sl.bias = bias; sl.rowbias = rowbias; sl.rb_ld = rb_ld; sl.T = T;
  sl.resid = resid; sl.ldr = ldr; sl.mat = mat; sl.raw = raw; sl.ld_raw = ld_raw;
  const int L = C / 40;
#define MVD_LNS(L_) \
  hipLaunchKernelGGL((layernorm_vec_kernel<L_, 5, true>), dim3(cdiv(rows, 4 * (64 / L_))), dim3(256), 0, s, slabs, rows, C, gamma, beta, eps, out, sl)
  if (L == 8) MVD_LNS(8);
  else if (L == 16) MVD_LNS(16);
  else if (L == 32) MVD_LNS(32);
  else MVD_LNS(64);
#undef MVD_LNS
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// fp32 output (a residual stream) from rows `ldx` floats apart (e.g. the class token of every sample)
int launch_layernorm_f32(const float* x, long ldx, int rows, int C, const float* gamma, const float* beta, float eps,
                         float* out, hipStream_t s) {
  if (C > 64 * 24) return mvd_fail("layernorm: C too large");
  hipLaunchKernelGGL(layernorm_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, rows, C, gamma, beta, eps, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
