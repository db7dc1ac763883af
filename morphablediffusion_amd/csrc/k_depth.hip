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
                                                         half_t* __restrict__ z, int npix, int HW, int D, int Cc, int split,
                                                         int nfill, const half_t* __restrict__ fill_row, int ldx) {
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
  // rows npix .. npix + nfill - 1 (the CFG-unconditional samples: all-zero context -> uniform softmax -> z = relu(beta)
  // for every head) are written from fill_row by the surplus waves of the same launch
  if (!live && pix < npix + nfill) {
    const int rowlen = (split ? 3 : 1) * H * Cc;
    const h8* src = (const h8*)fill_row;
    h8* dst = (h8*)(z + (long)pix * rowlen);
    for (int i = lane; i < rowlen / 8; i += 64) dst[i] = src[i];
  }
  const int b = live ? pix / HW : 0, p = live ? pix - b * HW : 0;

  if (live) {
    const int cpr = Cc / 8;
    for (int idx = lane; idx < D * cpr; idx += 64) {
      const int d = idx / cpr, ch = idx - d * cpr;
      *(h8*)(sX + d * xld + ch * 8) = *(const h8*)(ctxn + (((long)b * D + d) * HW + p) * ldx + ch * 8);
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
  if (live) {
    // softmax over d for the four heads, all 64 lanes: lane = (part, h), part strides d by 16; the maximum and the sum are
    // xor-shuffle reductions over the 16 parts (a 4-lane serial scan over D cost ~7 us per pixel at D = 48)
    const int h = lane & 3, part = lane >> 2;
    float mx = -INFINITY;
    for (int d = part; d < D; d += 16) mx = fmaxf(mx, sA[d * H + h]);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int d = part; d < D; d += 16) {
      const float e = __expf(sA[d * H + h] - mx);
      sA[d * H + h] = e;
      sum += e;
    }
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int d = part; d < D; d += 16) sA[d * H + h] *= inv;
  }
  __syncthreads();
  if (live) {
    // z[h][c] = sum_d a[h][d] x[d][c]: an item is (d-part, head, channel octet) -- 16-byte LDS reads of the staged column,
    // 16-byte fp16 stores; with fewer than 64 (head, octet) pairs the depth range is split over P lane groups and the
    // partial sums meet through xor shuffles
    const int cpr = Cc / 8, pairs = H * cpr;
    const int P = (pairs < 64 && (pairs & (pairs - 1)) == 0) ? 64 / pairs : 1;  // Cc = 64, 128, 256, 512: 32 ... 256 pairs
    const int W4 = H * Cc;
    for (int it = lane; it < pairs * P; it += 64) {
      const int part = it / pairs, rem = it - part * pairs, h = rem / cpr, oct = rem - h * cpr;
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int d = part; d < D; d += P) {
        const h8 xv = *(const h8*)(sX + d * xld + oct * 8);
        const float a = sA[d * H + h];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += a * (float)xv[k];
      }
      for (int o = pairs; o < 64; o <<= 1) {  // P > 1: lanes it, it + pairs, ... hold the same (head, octet)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], o);
      }
      if (part == 0) {
        h8 hi;
#pragma unroll
        for (int k = 0; k < 8; ++k) hi[k] = (half_t)acc[k];
        half_t* zr = z + (long)pix * (split ? 3 * W4 : W4) + h * Cc + oct * 8;
        *(h8*)zr = hi;
        if (split) {  // [hi | lo | hi] rows for the extended-precision output projection
          h8 lo;
#pragma unroll
          for (int k = 0; k < 8; ++k) lo[k] = (half_t)(acc[k] - (float)hi[k]);
          *(h8*)(zr + W4) = lo;
          *(h8*)(zr + 2 * W4) = hi;
        }
      }
    }
  }
}

}  // namespace

int launch_depth_attn(const float* qk, const half_t* ctxn, half_t* z, int n_cond, int HW, int D, int Cc, int heads,
                      hipStream_t s, int split, int nfill, const half_t* fill_row, int ldx) {
  if (heads != 4) return mvd_fail("depth_attn: the reference always uses 4 heads (attention.py:97-115)");
  if (Cc % 8 || D > 64) return mvd_fail("depth_attn: Cc must be a multiple of 8 and D <= 64");
  if (((uintptr_t)z & 15)) return mvd_fail("depth_attn: output must be 16-byte aligned");
  if (ldx == 0) ldx = Cc;
  if (ldx < Cc || (ldx & 7) || ((uintptr_t)ctxn & 15)) return mvd_fail("depth_attn: context rows must be 16-byte aligned and at least Cc wide");
  const int npix = n_cond * HW;
  if (nfill < 0 || (nfill > 0 && (!fill_row || ((uintptr_t)fill_row & 15)))) return mvd_fail("depth_attn: bad fill row");
  if (npix + nfill == 0) return 0;
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
  hipLaunchKernelGGL(depth_attn_kernel, dim3(cdiv(npix + nfill, 4)), dim3(256), lds, s, qk, ctxn, z, npix, HW, D, Cc, split,
                     nfill, fill_row, ldx);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
