// Training-step backward kernels (first slice): everything needed to back-propagate the MSE loss of the
// reference's training_step (morphable_diffusion.py:520-549) from the UNet output through the output head
// (openaimodel.py:717-721) into the LAST DepthTransformer (attention.py:49-84, output_conditions.8) and to produce the
// gradient of each of its parameters.  fp32 throughout, channels-last rows, deterministic (fixed summation orders, no
// atomics).  These are correctness-first kernels: a tiled fp32 GEMM with a two-stage split over the reduction axis, gather-form
// im2col / col2im, GroupNorm forward / backward with SiLU / ReLU, the depth attention forward / backward.
#include "common.h"

namespace {

constexpr int TS = 64, TK = 16;  // 64 x 64 output tile, 16-deep k slab, 256 threads x (4 x 4) outputs

// C[M,N] (+)= op(A) op(B);  A is [M,K] (ta = 0, row-major, lda) or [K,M] (ta = 1);  B is [K,N] (tb = 0) or [N,K] (tb = 1).
// grid.z splits K into equal slabs; with gridDim.z > 1 the partial products go to C + z * M * ldc (reduced by sum_slabs).
__global__ __launch_bounds__(256) void sgemm_kernel(const float* __restrict__ A, int lda, int ta, const float* __restrict__ B, int ldb,
                                                    int tb, float* __restrict__ C, int ldc, int M, int N, int K, int kper) {
  __shared__ float sA[TK][TS + 1], sB[TK][TS + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * TS, n0 = blockIdx.x * TS;
  const int kb = blockIdx.z * kper, ke = min(K, kb + kper);
  float acc[4][4] = {};
  for (int k0 = kb; k0 < ke; k0 += TK) {
    for (int i = threadIdx.x; i < TS * TK; i += 256) {
      int kk, mm;
      if (ta) { mm = i % TS; kk = i / TS; } else { kk = i % TK; mm = i / TK; }
      const int m = m0 + mm, k = k0 + kk;
      sA[kk][mm] = (m < M && k < ke) ? (ta ? A[(long)k * lda + m] : A[(long)m * lda + k]) : 0.f;
      int kn, nn;
      if (tb) { kn = i % TK; nn = i / TK; } else { nn = i % TS; kn = i / TS; }
      const int n = n0 + nn, k2 = k0 + kn;
      sB[kn][nn] = (n < N && k2 < ke) ? (tb ? B[(long)n * ldb + k2] : B[(long)k2 * ldb + n]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = sB[kk][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
  float* Cz = C + (long)blockIdx.z * M * ldc;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < M && n < N) Cz[(long)m * ldc + n] = acc[i][j];
    }
}

// out[i] = bias? + sum_s part[s][i]  (fixed order)
__global__ void sum_slabs_kernel(const float* __restrict__ part, int S, long n, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(long)s * n + i];
    out[i] = a;
  }
}

// col[r][tap * C + c] = X[b, y + dy, x + dx, c] (zero outside), tap = (dy+1)*3 + (dx+1)
__global__ void im2col3_kernel(const float* __restrict__ X, int B, int H, int W, int C, float* __restrict__ col) {
  const long total = (long)B * H * W * 9 * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int tap = (int)(t % 9);
    const long r = t / 9;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    col[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? X[(((long)b * H + yy) * W + xx) * C + c] : 0.f;
  }
}

// dX[b,y,x,c] = sum_tap dcol[(b, y - dy, x - dx)][tap * C + c]  (the transpose of im2col3 in gather form)
__global__ void col2im3_kernel(const float* __restrict__ dcol, int B, int H, int W, int C, float* __restrict__ dX) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int x = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    float a = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int ys = y - (tap / 3 - 1), xs = x - (tap % 3 - 1);
      if (ys >= 0 && ys < H && xs >= 0 && xs < W) a += dcol[(((long)b * H + ys) * W + xs) * 9 * C + tap * C + c];
    }
    dX[i] = a;
  }
}

// PyTorch conv weight [N][C][3][3] <-> GEMM matrix [N][9][C]
__global__ void perm_w3_kernel(const float* __restrict__ src, int N, int C, int to_mat, float* __restrict__ dst) {
  const long total = (long)N * C * 9;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9), c = (int)((i / 9) % C), n = (int)(i / (9L * C));  // index in the PyTorch layout
    const long m = ((long)n * 9 + tap) * C + c;
    if (to_mat) dst[m] = src[i];
    else dst[i] = src[m];
  }
}

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

// GroupNorm forward on x [B][rows][C] (channels-last), one block per (b, g): y = act(gamma xhat + beta); stats [B][G][2]
__global__ __launch_bounds__(256) void gn_fwd_kernel(const float* __restrict__ x, int rows, int C, int G, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, int act, float* __restrict__ y,
                                                     float* __restrict__ stats) {
  __shared__ float s_red[256];
  const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, n = rows * cpg;
  const float* xb = x + (long)b * rows * C + g * cpg;
  float a = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) a += xb[(long)(e / cpg) * C + e % cpg];
  const float mean = block_sum256(a, s_red) / (float)n;
  float q = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const float d = xb[(long)(e / cpg) * C + e % cpg] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(block_sum256(q, s_red) / (float)n + eps);
  if (threadIdx.x == 0) {
    stats[(b * G + g) * 2] = mean;
    stats[(b * G + g) * 2 + 1] = rstd;
  }
  float* yb = y + (long)b * rows * C + g * cpg;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int c = e % cpg;
    const long o = (long)(e / cpg) * C + c;
    yb[o] = act_f((xb[o] - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c], act);
  }
}

// GroupNorm (+ activation) backward, one block per (b, g):  du = dy act'(u);  dxhat = du gamma;
// dx = rstd (dxhat - mean_g(dxhat) - xhat mean_g(dxhat xhat)).  Also writes du xhat and du per element (dg_el, db_el may alias
// scratch) for the per-channel parameter reductions.
__global__ __launch_bounds__(256) void gn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, int rows, int C, int G,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ stats, int act, float* __restrict__ dx,
                                                     float* __restrict__ dg_el, float* __restrict__ db_el) {
  __shared__ float s_red[256];
  const int b = blockIdx.x / G, g = blockIdx.x % G, cpg = C / G, n = rows * cpg;
  const long base = (long)b * rows * C + g * cpg;
  const float mean = stats[(b * G + g) * 2], rstd = stats[(b * G + g) * 2 + 1];
  float s1 = 0.f, s2 = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int c = e % cpg;
    const long o = base + (long)(e / cpg) * C + c;
    const float xh = (x[o] - mean) * rstd, u = gamma[g * cpg + c] * xh + beta[g * cpg + c];
    const float du = dy[o] * act_d(u, act), dxh = du * gamma[g * cpg + c];
    s1 += dxh;
    s2 += dxh * xh;
    dg_el[o] = du * xh;
    db_el[o] = du;
  }
  const float m1 = block_sum256(s1, s_red) / (float)n, m2 = block_sum256(s2, s_red) / (float)n;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int c = e % cpg;
    const long o = base + (long)(e / cpg) * C + c;
    const float xh = (x[o] - mean) * rstd;
    const float dxh = db_el[o] * gamma[g * cpg + c];
    dx[o] = rstd * (dxh - m1 - xh * m2);
  }
}

// out[c] = sum over R rows of v[r][c]: one block per channel, fixed order (two-level: 256 partial sums, tree)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ v, long R, int C, float* __restrict__ out) {
  __shared__ float s_red[256];
  const int c = blockIdx.x;
  float a = 0.f;
  for (long r = threadIdx.x; r < R; r += 256) a += v[r * C + c];
  const float s = block_sum256(a, s_red);
  if (threadIdx.x == 0) out[c] = s;
}

// DepthAttention.forward (attention.py:26-47) per pixel; q [R][hn*hd], k / v [B*D*HW][hn*hd] (row = (b*D + d)*HW + p).
// One block of 128 threads per pixel (thread = head * hd + c, hn * hd <= 128, D <= 64): attn [R][hn][D], z [R][hn*hd].
__global__ __launch_bounds__(128) void depth_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        int HW, int D, int hn, int hd, float scale, float* __restrict__ attn,
                                                        float* __restrict__ z) {
  __shared__ float s_q[128], s_sim[4 * 64];
  const int r = blockIdx.x, b = r / HW, p = r % HW, t = threadIdx.x, I = hn * hd;
  if (t < I) s_q[t] = q[(long)r * I + t];
  __syncthreads();
  for (int e = t; e < hn * D; e += 128) {
    const int h = e / D, d = e % D;
    const float* kr = k + (((long)b * D + d) * HW + p) * I + h * hd;
    float a = 0.f;
    for (int c = 0; c < hd; ++c) a += s_q[h * hd + c] * kr[c];
    s_sim[h * 64 + d] = a * scale;
  }
  __syncthreads();
  if (t < hn) {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, s_sim[t * 64 + d]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = __expf(s_sim[t * 64 + d] - mx);
      s_sim[t * 64 + d] = e;
      sum += e;
    }
    for (int d = 0; d < D; ++d) {
      s_sim[t * 64 + d] /= sum;
      attn[((long)r * hn + t) * D + d] = s_sim[t * 64 + d];
    }
  }
  __syncthreads();
  if (t < I) {
    const int h = t / hd;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += s_sim[h * 64 + d] * v[(((long)b * D + d) * HW + p) * I + t];
    z[(long)r * I + t] = a;
  }
}

// backward of the above: dq [R][I], dk / dv [B*D*HW][I]
__global__ __launch_bounds__(128) void depth_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        const float* __restrict__ attn, const float* __restrict__ dz, int HW, int D,
                                                        int hn, int hd, float scale, float* __restrict__ dq, float* __restrict__ dk,
                                                        float* __restrict__ dv) {
  __shared__ float s_q[128], s_dz[128], s_a[4 * 64], s_ds[4 * 64];
  const int r = blockIdx.x, b = r / HW, p = r % HW, t = threadIdx.x, I = hn * hd;
  if (t < I) {
    s_q[t] = q[(long)r * I + t];
    s_dz[t] = dz[(long)r * I + t];
  }
  for (int e = t; e < hn * D; e += 128) s_a[(e / D) * 64 + e % D] = attn[((long)r * hn + e / D) * D + e % D];
  __syncthreads();
  for (int e = t; e < hn * D; e += 128) {  // dattn[h][d] = sum_c dz[h,c] v[d,h,c]
    const int h = e / D, d = e % D;
    const float* vr = v + (((long)b * D + d) * HW + p) * I + h * hd;
    float a = 0.f;
    for (int c = 0; c < hd; ++c) a += s_dz[h * hd + c] * vr[c];
    s_ds[h * 64 + d] = a;
  }
  __syncthreads();
  if (t < hn) {  // softmax backward: dsim = attn (dattn - sum attn dattn)
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot += s_a[t * 64 + d] * s_ds[t * 64 + d];
    for (int d = 0; d < D; ++d) s_ds[t * 64 + d] = s_a[t * 64 + d] * (s_ds[t * 64 + d] - dot);
  }
  __syncthreads();
  if (t < I) {
    const int h = t / hd;
    float a = 0.f;
    for (int d = 0; d < D; ++d) {
      const long o = (((long)b * D + d) * HW + p) * I + t;
      a += s_ds[h * 64 + d] * k[o];
      dk[o] = scale * s_ds[h * 64 + d] * s_q[t];
      dv[o] = s_a[h * 64 + d] * s_dz[t];
    }
    dq[(long)r * I + t] = scale * a;
  }
}

__global__ void scale_sub_kernel(const float* __restrict__ a, const float* __restrict__ b, float k, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = k * (a[i] - b[i]);
}
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}
__global__ void copy_rows_kernel(const float* __restrict__ src, int lds_, long rows, int C, float* __restrict__ dst) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[(i / C) * lds_ + i % C];
}

__global__ void add_bias_rows_kernel(float* __restrict__ x, long rows, int C, const float* __restrict__ bias) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) x[i] += bias[i % C];
}

inline int gridn(size_t n) {
  size_t g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

// C[M,N] = op(A) op(B); `scratch` (>= splits * M * N floats) is needed when the reduction axis is split (K > 4096)
int train_sgemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C, int M, int N, int K, float* scratch,
                size_t scratch_floats, hipStream_t s) {
  if (M <= 0 || N <= 0 || K <= 0) return mvd_fail("train_sgemm: empty problem");
  int splits = 1;
  const int tiles = cdiv(M, TS) * cdiv(N, TS);
  if (K > 4096 && tiles < 512) {
    splits = cdiv(1024, tiles);
    if (splits > cdiv(K, 512)) splits = cdiv(K, 512);
    if (splits > 256) splits = 256;
  }
  int kper = cdiv(cdiv(K, splits), TK) * TK;
  splits = cdiv(K, kper);
  if (splits > 1 && (size_t)splits * M * N > scratch_floats) return mvd_fail("train_sgemm: scratch too small for the split reduction");
  float* dst = splits > 1 ? scratch : C;
  hipLaunchKernelGGL(sgemm_kernel, dim3(cdiv(N, TS), cdiv(M, TS), splits), dim3(256), 0, s, A, lda, ta, B, ldb, tb, dst, N, M, N, K, kper);
  if (splits > 1) hipLaunchKernelGGL(sum_slabs_kernel, dim3(gridn((size_t)M * N)), dim3(256), 0, s, scratch, splits, (long)M * N, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_im2col3(const float* X, int B, int H, int W, int C, float* col, hipStream_t s) {
  hipLaunchKernelGGL(im2col3_kernel, dim3(gridn((size_t)B * H * W * 9 * C)), dim3(256), 0, s, X, B, H, W, C, col);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_col2im3(const float* dcol, int B, int H, int W, int C, float* dX, hipStream_t s) {
  hipLaunchKernelGGL(col2im3_kernel, dim3(gridn((size_t)B * H * W * C)), dim3(256), 0, s, dcol, B, H, W, C, dX);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_perm_w3(const float* src, int N, int C, int to_mat, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(perm_w3_kernel, dim3(gridn((size_t)N * C * 9)), dim3(256), 0, s, src, N, C, to_mat, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_gn_fwd(const float* x, int B, int rows, int C, int G, const float* gamma, const float* beta, float eps, int act, float* y,
                 float* stats, hipStream_t s) {
  hipLaunchKernelGGL(gn_fwd_kernel, dim3(B * G), dim3(256), 0, s, x, rows, C, G, gamma, beta, eps, act, y, stats);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// dx; dgamma / dbeta [C] (tmp1, tmp2: [B*rows*C] scratch each)
int train_gn_bwd(const float* x, const float* dy, int B, int rows, int C, int G, const float* gamma, const float* beta,
                 const float* stats, int act, float* dx, float* dgamma, float* dbeta, float* tmp1, float* tmp2, hipStream_t s) {
  hipLaunchKernelGGL(gn_bwd_kernel, dim3(B * G), dim3(256), 0, s, x, dy, rows, C, G, gamma, beta, stats, act, dx, tmp1, tmp2);
  if (dgamma) hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, s, tmp1, (long)B * rows, C, dgamma);
  if (dbeta) hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, s, tmp2, (long)B * rows, C, dbeta);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_colsum(const float* v, long R, int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(colsum_kernel, dim3(C), dim3(256), 0, s, v, R, C, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_depth_fwd(const float* q, const float* k, const float* v, int R, int HW, int D, int hn, int hd, float scale, float* attn,
                    float* z, hipStream_t s) {
  if (hn > 4 || hn * hd > 128 || D > 64) return mvd_fail("train_depth: needs heads <= 4, heads*dim_head <= 128, D <= 64");
  hipLaunchKernelGGL(depth_fwd_kernel, dim3(R), dim3(128), 0, s, q, k, v, HW, D, hn, hd, scale, attn, z);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_depth_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dz, int R, int HW, int D, int hn,
                    int hd, float scale, float* dq, float* dk, float* dv, hipStream_t s) {
  hipLaunchKernelGGL(depth_bwd_kernel, dim3(R), dim3(128), 0, s, q, k, v, attn, dz, HW, D, hn, hd, scale, dq, dk, dv);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_scale_sub(const float* a, const float* b, float k, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(scale_sub_kernel, dim3(gridn(n)), dim3(256), 0, s, a, b, k, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_add_inplace(float* a, const float* b, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(add_inplace_kernel, dim3(gridn(n)), dim3(256), 0, s, a, b, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_add_bias_rows(float* x, long rows, int C, const float* bias, hipStream_t s) {
  hipLaunchKernelGGL(add_bias_rows_kernel, dim3(gridn((size_t)rows * C)), dim3(256), 0, s, x, rows, C, bias);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_copy_rows(const float* src, int ld, long rows, int C, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(copy_rows_kernel, dim3(gridn((size_t)rows * C)), dim3(256), 0, s, src, ld, rows, C, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
