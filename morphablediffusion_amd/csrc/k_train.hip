// Training-step helper kernels: the fp32 pieces of the DepthTransformer backward (attention.py:49-84) that
// are bandwidth-bound and stay off the matrix cores -- gather-form im2col / col2im for its two 3x3 convs (their GEMMs run on
// the MFMA kernels through engine_train.hip: tgemm), the depth attention forward / backward, row
// utilities.  Deterministic (fixed summation orders, no atomics).
#include "common.h"

namespace {

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




// DepthAttention.forward (attention.py:26-47) per pixel; q [R][I], k / v [B*D*HW][I] (row = (b*D + d)*HW + p), I = hn * hd.
// One workgroup of 256 threads per pixel: the hn * D scores are wave-parallel dot products of length hd, the softmax over the
// D depth samples is done by one thread per head, z / dq / dk / dv are thread-per-channel.  attn [R][hn][D], z [R][I].
constexpr int DEPTH_MAX_I = 1024, DEPTH_MAX_D = 64, DEPTH_MAX_H = 4;
__global__ __launch_bounds__(256) void depth_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        int HW, int D, int hn, int hd, float scale, float* __restrict__ attn,
                                                        float* __restrict__ z) {
  __shared__ float s_q[DEPTH_MAX_I], s_sim[DEPTH_MAX_H * DEPTH_MAX_D];
  const int r = blockIdx.x, b = r / HW, p = r % HW, t = threadIdx.x, I = hn * hd, lane = t & 63, wave = t >> 6;
  for (int i = t; i < I; i += 256) s_q[i] = q[(long)r * I + i];
  __syncthreads();
  for (int e = wave; e < hn * D; e += 4) {
    const int h = e / D, d = e % D;
    const float* kr = k + (((long)b * D + d) * HW + p) * I + h * hd;
    float a = 0.f;
    for (int c = lane; c < hd; c += 64) a += s_q[h * hd + c] * kr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) s_sim[h * DEPTH_MAX_D + d] = a * scale;
  }
  __syncthreads();
  if (t < hn) {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, s_sim[t * DEPTH_MAX_D + d]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = __expf(s_sim[t * DEPTH_MAX_D + d] - mx);
      s_sim[t * DEPTH_MAX_D + d] = e;
      sum += e;
    }
    for (int d = 0; d < D; ++d) {
      s_sim[t * DEPTH_MAX_D + d] /= sum;
      attn[((long)r * hn + t) * D + d] = s_sim[t * DEPTH_MAX_D + d];
    }
  }
  __syncthreads();
  for (int i = t; i < I; i += 256) {
    const int h = i / hd;
    float a = 0.f;
    for (int d = 0; d < D; ++d) a += s_sim[h * DEPTH_MAX_D + d] * v[(((long)b * D + d) * HW + p) * I + i];
    z[(long)r * I + i] = a;
  }
}

// backward of the above: dq [R][I], dk / dv [B*D*HW][I]
__global__ __launch_bounds__(256) void depth_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                        const float* __restrict__ attn, const float* __restrict__ dz, int HW, int D,
                                                        int hn, int hd, float scale, float* __restrict__ dq, float* __restrict__ dk,
                                                        float* __restrict__ dv) {
  __shared__ float s_q[DEPTH_MAX_I], s_dz[DEPTH_MAX_I], s_a[DEPTH_MAX_H * DEPTH_MAX_D], s_ds[DEPTH_MAX_H * DEPTH_MAX_D];
  const int r = blockIdx.x, b = r / HW, p = r % HW, t = threadIdx.x, I = hn * hd, lane = t & 63, wave = t >> 6;
  for (int i = t; i < I; i += 256) {
    s_q[i] = q[(long)r * I + i];
    s_dz[i] = dz[(long)r * I + i];
  }
  for (int e = t; e < hn * D; e += 256) s_a[(e / D) * DEPTH_MAX_D + e % D] = attn[((long)r * hn + e / D) * D + e % D];
  __syncthreads();
  for (int e = wave; e < hn * D; e += 4) {  // dattn[h][d] = sum_c dz[h,c] v[d,h,c]
    const int h = e / D, d = e % D;
    const float* vr = v + (((long)b * D + d) * HW + p) * I + h * hd;
    float a = 0.f;
    for (int c = lane; c < hd; c += 64) a += s_dz[h * hd + c] * vr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) s_ds[h * DEPTH_MAX_D + d] = a;
  }
  __syncthreads();
  if (t < hn) {  // softmax backward: dsim = attn (dattn - sum attn dattn)
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot += s_a[t * DEPTH_MAX_D + d] * s_ds[t * DEPTH_MAX_D + d];
    for (int d = 0; d < D; ++d) s_ds[t * DEPTH_MAX_D + d] = s_a[t * DEPTH_MAX_D + d] * (s_ds[t * DEPTH_MAX_D + d] - dot);
  }
  __syncthreads();
  for (int i = t; i < I; i += 256) {
    const int h = i / hd;
    float a = 0.f;
    for (int d = 0; d < D; ++d) {
      const long o = (((long)b * D + d) * HW + p) * I + i;
      a += s_ds[h * DEPTH_MAX_D + d] * k[o];
      dk[o] = scale * s_ds[h * DEPTH_MAX_D + d] * s_q[i];
      dv[o] = s_a[h * DEPTH_MAX_D + d] * s_dz[i];
    }
    dq[(long)r * I + i] = scale * a;
  }
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
int train_depth_fwd(const float* q, const float* k, const float* v, int R, int HW, int D, int hn, int hd, float scale, float* attn,
                    float* z, hipStream_t s) {
  if (hn > DEPTH_MAX_H || hn * hd > DEPTH_MAX_I || D > DEPTH_MAX_D) return mvd_fail("train_depth: needs heads <= 4, heads*dim_head <= 1024, D <= 64");
  hipLaunchKernelGGL(depth_fwd_kernel, dim3(R), dim3(256), 0, s, q, k, v, HW, D, hn, hd, scale, attn, z);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int train_depth_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dz, int R, int HW, int D, int hn,
                    int hd, float scale, float* dq, float* dk, float* dv, hipStream_t s) {
  if (hn > DEPTH_MAX_H || hn * hd > DEPTH_MAX_I || D > DEPTH_MAX_D) return mvd_fail("train_depth: needs heads <= 4, heads*dim_head <= 1024, D <= 64");
  hipLaunchKernelGGL(depth_bwd_kernel, dim3(R), dim3(256), 0, s, q, k, v, attn, dz, HW, D, hn, hd, scale, dq, dk, dv);
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
