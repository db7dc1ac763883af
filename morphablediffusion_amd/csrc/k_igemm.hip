// Implicit-GEMM on the CDNA4 matrix cores: every 3x3 / 3x3x3 / transposed / 1x1 convolution and every
// Linear of the denoiser goes through this one kernel.
//
//   out[row(m)][n] = epilogue( sum_{tap} sum_{c} A[in(m,tap)][c] * W[tap][n][c] )
//
// * 128x128x64 tile, 256 threads = 4 waves in a 2x2 grid, each wave a 64x64 sub-tile = 2x2 fragments of
//   v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate).
// * A (channels-last activations, fp32 or fp16 in HBM) is gathered per tap with zero fill at the borders,
//   converted to fp16 while it is staged; W is fp16 [tap][N][Cin].  Both land in LDS as [row][64 halfs]
//   with the 16-byte chunk index XORed by (row>>1)&7, which makes the 16-lane groups of ds_read_b128
//   conflict-free for the fragment reads (row = lane&31, chunk = 2*kk + lane>>5).
// * register-staged double buffering: the global loads of k-step s+1 are in flight while the MFMAs of
//   k-step s run; one barrier per k-step.
// * fused epilogue: alpha, bias[n], per-sample bias[b][n] (timestep embedding), residual, GEGLU pairing,
//   fp16 or fp32 store with an arbitrary row mapping (concat-by-construction, transposed-conv parity
//   scatter).  split-K writes fp32 partials and a second kernel applies the same epilogue.
// * blockIdx is remapped so that consecutive logical tiles (which share A rows) run on the same XCD.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NT = 256;

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ long out_row(const IGemm& g, int m) {
  if (g.out_linear) return m;
  int x = m % g.X;
  int t = m / g.X;
  int y = t % g.Y;
  t /= g.Y;
  int z = t % g.Z;
  int b = t / g.Z;
  return ((long)(b * g.OZ + z * g.ozm + g.ozo) * g.OY + (y * g.oym + g.oyo)) * g.OX + (x * g.oxm + g.oxo);
}

// v: accumulator for column n (and `gate` for column n+32 when geglu)
__device__ __forceinline__ void epilogue_store(const IGemm& g, int m, long orow, int n, float v, float gate) {
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
  else ((half_t*)g.out)[orow * g.ldc + ncol] = (half_t)v;
}

template <bool A_F32>
__global__ __launch_bounds__(NT, 2) void igemm_kernel(const IGemm g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = g.B * g.Z * g.Y * g.X;
  const int N = g.N;
  const int tiles_n = (N + BN - 1) / BN;
  const int tiles_m = (M + BM - 1) / BM;
  // XCD-aware bijective remap (hardware places block b on XCD b % 8)
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const int cpt = (g.Cin + BK - 1) / BK;
  const int ksteps = g.ntaps * cpt;
  int kbeg = 0, kend = ksteps;
  if (g.splitk > 1) {
    const int per = (ksteps + g.splitk - 1) / g.splitk;
    kbeg = blockIdx.y * per;
    kend = min(ksteps, kbeg + per);
  }

  const int chunk = tid & 7, r0 = tid >> 3;
  int ab[4], az[4], ay[4], ax[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < M) {
      int x = m % g.X;
      int t = m / g.X;
      int y = t % g.Y;
      t /= g.Y;
      int z = t % g.Z;
      ab[i] = t / g.Z;
      az[i] = z * g.sz;
      ay[i] = y * g.sy;
      ax[i] = x * g.sx;
    } else {
      ab[i] = -1;
      az[i] = ay[i] = ax[i] = 0;
    }
  }

  float4 ra32[A_F32 ? 4 : 1][2];
  h8 ra16[A_F32 ? 1 : 4];
  h8 rb[4];

  // Per-tap gather state: element offsets of the 4 A rows / 4 W rows this thread stages (-1 = zero fill).
  // Recomputed only when the tap changes; inside a tap the k-steps just advance the channel offset.
  long a_off[4], b_off[4];
  int ld_tap = kbeg / cpt, ld_cc = kbeg - ld_tap * cpt;
  auto set_tap = [&](int tap) {
    const int dz = g.dz[tap], dy = g.dy[tap], dx = g.dx[tap];
    const int wslab = g.wt[tap];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iz = az[i] + dz, iy = ay[i] + dy, ix = ax[i] + dx;
      const bool ok = ab[i] >= 0 && iz >= 0 && iz < g.IZ && iy >= 0 && iy < g.IY && ix >= 0 && ix < g.IX;
      const long pix = ((long)(ab[i] * g.PZ + (iz >> g.ups)) * g.PY + (iy >> g.ups)) * g.PX + (ix >> g.ups);
      a_off[i] = ok ? pix * g.lda + chunk * 8 : -1;
      const int n = n0 + r0 + 32 * i;
      b_off[i] = n < N ? ((long)wslab * N + n) * g.Cin + chunk * 8 : -1;
    }
  };
  if (kbeg < kend) set_tap(ld_tap);

  auto load_tiles = [&]() {
    const int cb = ld_cc * BK;
    const bool cok = cb + chunk * 8 < g.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = cok && a_off[i] >= 0;
      if constexpr (A_F32) {
        if (ok) {
          const float4* p = (const float4*)((const float*)g.a + a_off[i] + cb);
          ra32[i][0] = p[0];
          ra32[i][1] = p[1];
        } else {
          ra32[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
          ra32[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        if (ok) ra16[i] = *(const h8*)((const half_t*)g.a + a_off[i] + cb);
        else ra16[i] = (h8)(half_t)0;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (cok && b_off[i] >= 0) rb[i] = *(const h8*)(g.w + b_off[i] + cb);
      else rb[i] = (h8)(half_t)0;
    }
    // advance to the next k-step
    if (++ld_cc == cpt) {
      ld_cc = 0;
      ++ld_tap;
      if (ld_tap < g.ntaps) set_tap(ld_tap);
    }
  };
  auto store_tiles = [&](int buf) {
    char* sA = smem + buf * 32768;
    char* sB = sA + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      h8 v;
      if constexpr (A_F32) {
        v[0] = (half_t)ra32[i][0].x; v[1] = (half_t)ra32[i][0].y; v[2] = (half_t)ra32[i][0].z; v[3] = (half_t)ra32[i][0].w;
        v[4] = (half_t)ra32[i][1].x; v[5] = (half_t)ra32[i][1].y; v[6] = (half_t)ra32[i][1].z; v[7] = (half_t)ra32[i][1].w;
      } else {
        v = ra16[i];
      }
      *(h8*)(sA + swz(row, chunk)) = v;
      *(h8*)(sB + swz(row, chunk)) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kbeg < kend) {
    load_tiles();
    store_tiles(0);
  }
  __syncthreads();
  int cur = 0;
  for (int ks = kbeg; ks < kend; ++ks) {
    const bool more = ks + 1 < kend;
    if (more) load_tiles();
    const char* sA = smem + cur * 32768;
    const char* sB = sA + 16384;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int ch = kk * 2 + (lane >> 5);
      h8 af[2], bf[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        af[f] = *(const h8*)(sA + swz(wm * 64 + f * 32 + (lane & 31), ch));
        bf[f] = *(const h8*)(sB + swz(wn * 64 + f * 32 + (lane & 31), ch));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ----
  const int ncol0 = n0 + wn * 64 + (lane & 31);
  if (g.splitk > 1) {
    float* part = g.partial + (long)blockIdx.y * M * N;
#pragma unroll
    for (int fm = 0; fm < 2; ++fm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
          const int n = ncol0 + fn * 32;
          if (n < N) part[(long)m * N + n] = acc[fm][fn][r];
        }
      }
    return;
  }
#pragma unroll
  for (int fm = 0; fm < 2; ++fm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      const long orow = out_row(g, m);
      if (g.geglu) {
        if (ncol0 + 32 < N) epilogue_store(g, m, orow, ncol0, acc[fm][0][r], acc[fm][1][r]);
      } else {
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
          const int n = ncol0 + fn * 32;
          if (n < N) epilogue_store(g, m, orow, n, acc[fm][fn][r], 0.f);
        }
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IGemm g) {
  const int M = g.B * g.Z * g.Y * g.X;
  const int N = g.N;
  const long total = (long)M * N;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
    if (g.geglu && (n & 32)) continue;
    float v = 0.f, gate = 0.f;
    for (int s = 0; s < g.splitk; ++s) {
      v += g.partial[(long)s * total + idx];
      if (g.geglu) gate += g.partial[(long)s * total + idx + 32];
    }
    epilogue_store(g, m, out_row(g, m), n, v, gate);
  }
}

}  // namespace

int igemm_pick_splitk(int M, int N, int ksteps) {
  const int tiles = cdiv(M, BM) * cdiv(N, BN);
  if (tiles >= 192 || ksteps < 8) return 1;
  int sk = cdiv(512, tiles);
  if (sk > ksteps / 4) sk = ksteps / 4;
  if (sk > 16) sk = 16;
  return sk < 1 ? 1 : sk;
}

size_t igemm_partial_bytes(const IGemm& g) {
  if (g.splitk <= 1) return 0;
  return (size_t)g.splitk * g.B * g.Z * g.Y * g.X * g.N * sizeof(float);
}

int launch_igemm(const IGemm& g, hipStream_t s) {
  const int M = g.B * g.Z * g.Y * g.X;
  if (M <= 0 || g.N <= 0) return 0;
  if (g.Cin % 8) return mvd_fail("igemm: Cin must be a multiple of 8");
  if (g.ntaps < 1 || g.ntaps > MVD_MAX_TAPS) return mvd_fail("igemm: bad tap count");
  if (g.geglu && (g.N % 64)) return mvd_fail("igemm: GEGLU needs N % 64 == 0");
  if (g.splitk > 1 && !g.partial) return mvd_fail("igemm: split-K without a partial buffer");
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    attr_set = true;
  }
  dim3 grid(cdiv(M, BM) * cdiv(g.N, BN), g.splitk > 1 ? g.splitk : 1);
  if (g.a_f32) hipLaunchKernelGGL(igemm_kernel<true>, grid, dim3(NT), 65536, s, g);
  else hipLaunchKernelGGL(igemm_kernel<false>, grid, dim3(NT), 65536, s, g);
  HIP_CHECK_RET(hipGetLastError());
  if (g.splitk > 1) {
    long total = (long)M * g.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
    HIP_CHECK_RET(hipGetLastError());
  }
  return 0;
}
