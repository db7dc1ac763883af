// NoisyTargetViewEncoder (network.py:181-207) as ONE kernel: the 2-D encoder of the conditioner is eight 3x3 convs
// (4 -> 16 -> ... -> 16 channels) and seven GroupNorm(8)+SiLU over a 32 x 32 image per view -- 75 MFLOP per conv for 16
// views, which as separate launches (1 layout pass + 8 implicit GEMMs + 7 norms, each 10-22 us of launch, set-up and drain)
// cost ~0.3 ms of every step, on its critical path.  Here one workgroup (16 waves) owns one view:
//   * the activations that a conv reads live in LDS as a zero-bordered 34 x 34 x 16 fp16 tile (the same fp16 operand
//     rounding as the implicit-GEMM path: GroupNorm / SiLU in fp32, one rounding to fp16, fp32 accumulation);
//   * a conv is 64 output tiles of 16 pixels x 16 channels, 5 x v_mfma_f32_16x16x32_f16 each (k = tap * 16 + channel,
//     9 taps padded to 10; A fragments are 16-byte LDS reads of the shifted pixel, B fragments -- the packed [tap][n][cin]
//     weights of the implicit GEMM -- sit in registers for the whole conv);
//   * the residual stream h and the block-internal tensor stay in registers in the MFMA C layout (lane = channel
//     n = lane & 15, four pixels per tile), so GroupNorm's per-group sums are in-lane sums + three xor shuffles + one
//     exchange between the waves through LDS.
// Output: feats [views * 1024][16] fp32, the operand of the vertex gather.
#include "common.h"

namespace {

[[maybe_unused]] constexpr int ES = 32, EPX = ES * ES, ETS = ES + 2, EPS = 16;  // image side, pixels, halo tile side, halfs per tile pixel
[[maybe_unused]] constexpr int ENT = 1024, ETILES = EPX / 16 / (ENT / 64);       // 4 output tiles of 16 pixels per wave

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct EncArgs {
  const half_t* w[8];   // packed conv weights [9][16][cin] fp16: init, (c1, c2) x 3, final
  const float* bias[8];
  int cin[8];           // 8 for the init conv (4 latent channels padded), 16 otherwise
  const float* gamma[7];  // GroupNorm(8): (n1, n2) x 3, final
  const float* beta[7];
};

__global__ __launch_bounds__(ENT) void target_encoder_kernel(const float* __restrict__ x, const float* __restrict__ pre,
                                                             const EncArgs W, float* __restrict__ feats) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) half_t tile[ETS * ETS * EPS];
  __shared__ float red[14][ENT / 64][16];
  const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;

  for (int i = tid; i < ETS * ETS * EPS / 8; i += ENT) ((h8*)tile)[i] = (h8)(half_t)0;
  __syncthreads();
  // the noisy latent (4 channels, NCHW) -> channels 0..3 of the tile interior
  for (int i = tid; i < 4 * EPX; i += ENT) {
    const int c = i >> 10, p = i & (EPX - 1);
    tile[(((p >> 5) + 1) * ETS + (p & 31) + 1) * EPS + c] = (half_t)x[((long)v * 4 + c) * EPX + p];
  }
  __syncthreads();

  // out[t][r]: channel n of pixel (wave * 16 + t) * 16 + q * 4 + r
  auto conv = [&](int ci, float (&out)[ETILES][4], const float (*resid)[4]) {
    const half_t* w = W.w[ci];
    const int cin = W.cin[ci];
    h8 bf[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int tap = 2 * ks + (q >> 1), c0 = (q & 1) * 8;
      bf[ks] = (h8)(half_t)0;
      if (tap < 9 && c0 < cin) bf[ks] = *(const h8*)(w + ((long)tap * 16 + n) * cin + c0);
    }
    const float b = W.bias[ci][n];
#pragma unroll
    for (int t = 0; t < ETILES; ++t) {
      const int p0 = (wave * ETILES + t) * 16, y = p0 >> 5, x0 = p0 & 31;
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        int tap = 2 * ks + (q >> 1);
        if (tap > 8) tap = 8;  // the padding tap: zero weights, any finite operand
        const int dy = tap / 3, dx = tap - dy * 3;  // tile coordinates are image coordinates + 1
        const h8 af = *(const h8*)(tile + ((y + dy) * ETS + x0 + n + dx) * EPS + (q & 1) * 8);
        acc = MVD_MFMA_16x16x32(af, bf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) out[t][r] = acc[r] + b + (resid ? resid[t][r] : 0.f);
    }
  };

  // tile <- fp16(silu(GroupNorm8(u + add[n]))) of the wave-distributed tensor u (the conv that follows reads it)
  auto norm_to_tile = [&](int ni, const float (&u)[ETILES][4], float add) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < ETILES; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += u[t][r] + add;
    s += __shfl_xor(s, 1);   // the group's other channel
    s += __shfl_xor(s, 16);  // the other pixel quarters of the tiles
    s += __shfl_xor(s, 32);
    if (q == 0) red[2 * ni][wave][n] = s;
    __syncthreads();  // also: every wave is done reading the tile (the previous conv)
    float mean = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < ENT / 64; ++w2) mean += red[2 * ni][w2][n];
    mean *= 1.0f / (2.0f * EPX);
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < ETILES; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = u[t][r] + add - mean;
        sq += d * d;
      }
    sq += __shfl_xor(sq, 1);
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    if (q == 0) red[2 * ni + 1][wave][n] = sq;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < ENT / 64; ++w2) var += red[2 * ni + 1][w2][n];
    const float rstd = rsqrtf(var * (1.0f / (2.0f * EPX)) + 1e-5f);
    const float ga = W.gamma[ni][n], be = W.beta[ni][n];
#pragma unroll
    for (int t = 0; t < ETILES; ++t) {
      const int p0 = (wave * ETILES + t) * 16 + q * 4, y = p0 >> 5, x0 = p0 & 31;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = (u[t][r] + add - mean) * rstd * ga + be;
        tile[((y + 1) * ETS + x0 + r + 1) * EPS + n] = (half_t)(z / (1.0f + __expf(-z)));
      }
    }
    __syncthreads();
  };

  float h[ETILES][4], r1[ETILES][4];
  conv(0, h, nullptr);
  for (int i = 0; i < 3; ++i) {
    norm_to_tile(2 * i, h, pre[(long)v * 48 + 16 * i + n]);
    conv(1 + 2 * i, r1, nullptr);
    norm_to_tile(2 * i + 1, r1, 0.f);
    conv(2 + 2 * i, h, h);
  }
  norm_to_tile(6, h, 0.f);
  conv(7, r1, nullptr);
#pragma unroll
  for (int t = 0; t < ETILES; ++t) {
    const int p0 = (wave * ETILES + t) * 16 + q * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) feats[((long)v * EPX + p0 + r) * 16 + n] = r1[t][r];
  }
#endif
}

}  // namespace

// weights: the implicit GEMM's packed tensors ([9][16][cin] fp16, bias fp32[16]) in layer order; norms: gamma / beta fp32[16]
int launch_target_encoder(const float* x, const float* pre, int n_views, const half_t* const* w, const float* const* bias,
                          const int* cin, const float* const* gamma, const float* const* beta, float* feats, hipStream_t s) {
  EncArgs a;
  for (int i = 0; i < 8; ++i) {
    if (!w[i] || !bias[i] || (cin[i] != 8 && cin[i] != 16) || ((uintptr_t)w[i] & 15))
      return mvd_fail("target_encoder: packed 16-channel conv weights with bias expected");
    a.w[i] = w[i];
    a.bias[i] = bias[i];
    a.cin[i] = cin[i];
  }
  for (int i = 0; i < 7; ++i) {
    if (!gamma[i] || !beta[i]) return mvd_fail("target_encoder: norm parameters missing");
    a.gamma[i] = gamma[i];
    a.beta[i] = beta[i];
  }
  if (n_views <= 0) return 0;
  hipLaunchKernelGGL(target_encoder_kernel, dim3(n_views), dim3(ENT), 0, s, x, pre, a, feats);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
