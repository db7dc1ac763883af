// Depth-wise attention along each pixel's frustum ray (reference: DepthAttention.forward,
// ldm/models/diffusion/attention.py:26-47) WITHOUT materialising K and V.
//
// The reference projects the context volume to K,V = W_k c, W_v c  ([b, 2Cc, D, h, w] each: the largest
// activations of the whole UNet) and then reduces over channels / depth.  By associativity
//     sim[h,d]  = sum_c q[h,c] K[h,c,d]        = sum_j (W_k,h^T q_h)[j] * ctx[j,d]   =: qk_h . ctx[:,d]
//     out[h,:]  = sum_d a[h,d] V[h,:,d]        = W_v,h (sum_d a[h,d] ctx[:,d])       =: W_v,h z_h
// so the per-pixel work only needs the normalised context column ctx[D][Cc] (read ONCE, fp16) and the
// folded query qk[heads][Cc]; W_k^T W_q and W_o W_v are folded into two ordinary GEMMs at weight upload.
// One wave per pixel; the column is staged in LDS and consumed twice (row-wise for the scores, column-wise
// for z).  HBM-bound: D*Cc*2 bytes per pixel.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void depth_attn_kernel(const float* __restrict__ qk, const half_t* __restrict__ ctxn,
                                                         half_t* __restrict__ z, int npix, int HW, int D, int Cc, int split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int H = 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xld = Cc + 8;
  const int x_bytes = ((D * xld * 2 + 15) / 16) * 16;
  const int per_wave = x_bytes + H * Cc * 4 + 64 * H * 4;
  char* base = smem + wave * per_wave;
  half_t* sX = (half_t*)base;
  float* sQ = (float*)(base + x_bytes);
  float* sA = sQ + H * Cc;  // [d][h] scores then probabilities, D <= 64
  const int pix = blockIdx.x * 4 + wave;
  const bool live = pix < npix;
  const int b = live ? pix / HW : 0, p = live ? pix - b * HW : 0;

  if (live) {
    const int cpr = Cc / 8;
    for (int idx = lane; idx < D * cpr; idx += 64) {
      const int d = idx / cpr, ch = idx - d * cpr;
      *(h8*)(sX + d * xld + ch * 8) = *(const h8*)(ctxn + (((long)b * D + d) * HW + p) * Cc + ch * 8);
    }
    const float4* q4 = (const float4*)(qk + (long)pix * H * Cc);
    for (int idx = lane; idx < H * Cc / 4; idx += 64) ((float4*)sQ)[idx] = q4[idx];
  }
  __syncthreads();
  if (live) {
    for (int e = lane; e < H * D; e += 64) {
      const int h = e / D, d = e - h * D;
      const half_t* xr = sX + d * xld;
      const float* qr = sQ + h * Cc;
      float acc = 0.f;
      for (int c = 0; c < Cc; c += 8) {
        const h8 xv = *(const h8*)(xr + c);
        const float4 qa = *(const float4*)(qr + c), qb = *(const float4*)(qr + c + 4);
        acc += (float)xv[0] * qa.x + (float)xv[1] * qa.y + (float)xv[2] * qa.z + (float)xv[3] * qa.w +
               (float)xv[4] * qb.x + (float)xv[5] * qb.y + (float)xv[6] * qb.z + (float)xv[7] * qb.w;
      }
      sA[d * H + h] = acc;
    }
  }
  __syncthreads();
  if (live && lane < H) {
    float mx = -INFINITY;
    for (int d = 0; d < D; ++d) mx = fmaxf(mx, sA[d * H + lane]);
    float sum = 0.f;
    for (int d = 0; d < D; ++d) {
      const float e = __expf(sA[d * H + lane] - mx);
      sA[d * H + lane] = e;
      sum += e;
    }
    const float inv = 1.0f / sum;
    for (int d = 0; d < D; ++d) sA[d * H + lane] *= inv;
  }
  __syncthreads();
  if (live) {
    for (int c = lane; c < Cc; c += 64) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      for (int d = 0; d < D; ++d) {
        const float x = (float)sX[d * xld + c];
        const float4 a = *(const float4*)(sA + d * H);
        a0 += a.x * x;
        a1 += a.y * x;
        a2 += a.z * x;
        a3 += a.w * x;
      }
      const int W4 = H * Cc;
      half_t* zr = z + (long)pix * (split ? 3 * W4 : W4) + c;
      const float av[4] = {a0, a1, a2, a3};
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const half_t hi = (half_t)av[h];
        zr[h * Cc] = hi;
        if (split) {  // [hi | lo | hi] rows for the extended-precision output projection
          zr[W4 + h * Cc] = (half_t)(av[h] - (float)hi);
          zr[2 * W4 + h * Cc] = hi;
        }
      }
    }
  }
}

}  // namespace

int launch_depth_attn(const float* qk, const half_t* ctxn, half_t* z, int n_cond, int HW, int D, int Cc, int heads,
                      hipStream_t s, int split) {
  if (heads != 4) return mvd_fail("depth_attn: the reference always uses 4 heads (attention.py:97-115)");
  if (Cc % 8 || D > 64) return mvd_fail("depth_attn: Cc must be a multiple of 8 and D <= 64");
  const int npix = n_cond * HW;
  if (npix == 0) return 0;
  const int x_bytes = ((D * (Cc + 8) * 2 + 15) / 16) * 16;
  const int per_wave = x_bytes + 4 * Cc * 4 + 64 * 4 * 4;
  const int lds = per_wave * 4;
  if (lds > 160 * 1024) return mvd_fail("depth_attn: LDS budget exceeded");
  static bool attr_done[MVD_MAX_DEVICES] = {false};  // the attribute is per device
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)depth_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(depth_attn_kernel, dim3(cdiv(npix, 4)), dim3(256), lds, s, qk, ctxn, z, npix, HW, D, Cc, split);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
