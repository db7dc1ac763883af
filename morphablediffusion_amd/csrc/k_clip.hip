// CLIP image embedding front end (FrozenCLIPImageEmbedder.preprocess, ldm/modules/encoders/modules.py:363-371, and the
// patch / token assembly of the vision transformer): the transformer body itself reuses the UNet's GEMM, LayerNorm and
// flash-attention kernels (engine_clip.hip).
#include "common.h"

namespace {

// PyTorch's bicubic kernel (A = -0.75), the interpolation kornia.geometry.resize(..., 'bicubic') ends up in
__device__ __forceinline__ float cubic1(float x) { return ((-0.75f + 2.f) * x - (-0.75f + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x) { return ((-0.75f * x + 3.75f) * x - 6.f) * x + 3.f; }

// resize (bicubic, align_corners=True, no antialias) -> (v+1)/2 -> normalise -> im2col of the patch convolution:
// out[(b*G + py)*G + px][c*P*P + ky*P + kx] in fp16, columns >= 3*P*P zero (K padded to a multiple of 8)
__global__ void clip_patches_kernel(const float* __restrict__ img, int B, int H, int W, int S, int P, int Kp,
                                    half_t* __restrict__ out) {
  const int G = S / P, K = 3 * P * P;
  const long total = (long)B * G * G * Kp;
  const float sy = S > 1 ? (float)(H - 1) / (float)(S - 1) : 0.f, sx = S > 1 ? (float)(W - 1) / (float)(S - 1) : 0.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int col = (int)(idx % Kp);
    const long row = idx / Kp;
    if (col >= K) {
      out[idx] = (half_t)0;
      continue;
    }
    const int px = (int)(row % G), py = (int)((row / G) % G), b = (int)(row / ((long)G * G));
    const int c = col / (P * P), ky = (col / P) % P, kx = col % P;
    const int y = py * P + ky, x = px * P + kx;
    const float ry = sy * y, rx = sx * x;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    const float ty = ry - iy, tx = rx - ix;
    const float wy[4] = {cubic2(ty + 1.f), cubic1(ty), cubic1(1.f - ty), cubic2(2.f - ty)};
    const float wx[4] = {cubic2(tx + 1.f), cubic1(tx), cubic1(1.f - tx), cubic2(2.f - tx)};
    const float* pl = img + ((long)b * 3 + c) * H * W;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yy = min(max(iy - 1 + j, 0), H - 1);
      float r = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) r += wx[i] * pl[(long)yy * W + min(max(ix - 1 + i, 0), W - 1)];
      acc += wy[j] * r;
    }
    const float mean = c == 0 ? 0.48145466f : (c == 1 ? 0.4578275f : 0.40821073f);
    const float stdv = c == 0 ? 0.26862954f : (c == 1 ? 0.26130258f : 0.27577711f);
    out[idx] = (half_t)(((acc + 1.f) * 0.5f - mean) / stdv);
  }
}

// x[b][t] = (t == 0 ? class_embedding : patch_embed[b][t-1]) + positional_embedding[t] for t < T; rows T..Tp-1 of every
// sample are zero padding (the token axis is padded to a multiple of 8 for the attention kernel's vector loads)
__global__ void clip_tokens_kernel(const float* __restrict__ pe, const float* __restrict__ cls, const float* __restrict__ pos,
                                   int B, int T, int Tp, int C, float* __restrict__ x) {
  const long total = (long)B * Tp * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % C);
    const long row = idx / C;
    const int t = (int)(row % Tp), b = (int)(row / Tp);
    float v = 0.f;
    if (t < T) v = (t == 0 ? cls[ch] : pe[((long)b * (T - 1) + t - 1) * C + ch]) + pos[(long)t * C + ch];
    x[idx] = v;
  }
}

__global__ void scale_copy_kernel(const float* __restrict__ src, size_t n, float k, float* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i] * k;
}

inline unsigned grid_1d(size_t n) {
  size_t g = (n + 255) / 256;
  return (unsigned)(g > 65535 ? 65535 : (g ? g : 1));
}

}  // namespace

int launch_clip_patches(const float* img, int B, int H, int W, int S, int P, int Kp, half_t* out, hipStream_t s) {
  if (S % P || Kp < 3 * P * P || (Kp & 7)) return mvd_fail("clip_patches: image % patch == 0 and a padded K expected");
  hipLaunchKernelGGL(clip_patches_kernel, dim3(grid_1d((size_t)B * (S / P) * (S / P) * Kp)), dim3(256), 0, s, img, B, H, W, S, P,
                     Kp, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_clip_tokens(const float* pe, const float* cls, const float* pos, int B, int T, int Tp, int C, float* x,
                       hipStream_t s) {
  hipLaunchKernelGGL(clip_tokens_kernel, dim3(grid_1d((size_t)B * Tp * C)), dim3(256), 0, s, pe, cls, pos, B, T, Tp, C, x);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_scale_copy(const float* src, size_t n, float k, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_1d(n)), dim3(256), 0, s, src, n, k, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
