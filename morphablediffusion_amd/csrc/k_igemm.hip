// Implicit-GEMM on the CDNA4 matrix cores: every 3x3 / 3x3x3 / transposed / 1x1 convolution and every
// Linear of the denoiser goes through this one kernel.
//
//   out[row(m)][n] = epilogue( sum_{tap} sum_{c} A[in(m,tap)][c] * W[tap][n][c] )
//
// * 128 x BN x 64 tile, 256 threads = 4 waves, v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate).
//   Three column widths so that the layer widths of this model tile without waste:
//     BN=128: waves 2x2, 64x64 each (2x2 fragments)      -- N = 640/1280/1920/2560, GEGLU
//     BN=160: waves 4x1, 32x160 each (1x5 fragments)     -- N = 320/960 (level-32 convs: exactly 2 tiles)
//     BN= 64: waves 4x1, 32x64 each (1x2 fragments)      -- N <= 64 (frustum / context / encoder layers)
// * A (channels-last activations, fp32 or fp16 in HBM) is gathered per tap through buffer loads whose
//   out-of-range offsets (spatial padding, M/N/K tails) return zeros in hardware: no divergent control
//   flow around the loads, all loads of a k-step are issued back to back and stay in flight behind the
//   MFMAs of the previous k-step (register-staged double buffering, one barrier per k-step).
// * LDS layout [row][64 halfs] with the 16-byte chunk index XORed by (row>>1)&7: the 16-lane groups of
//   ds_read_b128 are conflict-free for the fragment reads (row = lane&31, chunk = 2*kk + lane>>5).
// * fused epilogue: alpha, bias[n], per-sample bias[b][n] (timestep embedding), residual, SiLU, GEGLU
//   pairing, fp16 or fp32 store with an arbitrary row mapping (concat-by-construction, transposed-conv
//   parity scatter).  split-K writes fp32 partials and a second kernel applies the same epilogue.
// * blockIdx is remapped so that consecutive logical tiles (which share A rows) run on the same XCD.
#include "common.h"
#include "igemm_epilogue.h"

namespace {

constexpr int BM = 128, BK = 64, NT = 256;

#ifdef MVD_TIMELINE
// investigation build only (make EXTRA=-DMVD_TIMELINE): per-workgroup phase timestamps of the last launch
__device__ unsigned long long mvd_ig_tl[8 * 8192];
#define TLI(i)                                                                                               \
  do {                                                                                                       \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) mvd_ig_tl[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define TLI(i)
#endif
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// PLAIN: one centre tap, unit strides, linear input and output rows (Linear layers, 1x1 convs): no integer division in the
// set-up or the epilogue (see k_gemm.hip: the first pass over a kernel's code runs at instruction-fetch speed).
template <bool A_F32, int BN, int WAVES_M, int WAVES_N, bool PLAIN = false>
__global__ __launch_bounds__(NT, 2) void igemm_kernel(const IGemm g) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;  // wave sub-tile
  constexpr int FM = WM / 32, FN = WN / 32;             // 32x32 fragments per wave
  constexpr int NB = BN / 32;                           // W rows staged per thread
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  TLI(0);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
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
    if constexpr (PLAIN) {
      ab[i] = m < M ? 0 : -1;
      az[i] = ay[i] = 0;
      ax[i] = m;
    } else if (m < M) {
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

  constexpr unsigned OOB = 0xFFFFFFFFu;
  constexpr int ESZ = A_F32 ? 4 : 2;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), (short)0, 0xFFFFFFFEu, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.w), (short)0, 0xFFFFFFFEu, 0x00020000);
  i32x4 ra[A_F32 ? 8 : 4];
  i32x4 rb[NB];
  unsigned a_off[4], b_off[NB];  // byte offsets of this thread's A rows / W rows for the current tap
  int ld_tap = kbeg / cpt, ld_cc = kbeg - ld_tap * cpt, set_for = -1;
  auto set_tap = [&](int tap) {
    // one packed dword per tap, fetched with a scalar load (the tap index is wave-uniform)
    const int ti = g.tap[__builtin_amdgcn_readfirstlane(tap)];
    const int dz = (ti & 3) - 1, dy = ((ti >> 2) & 3) - 1, dx = ((ti >> 4) & 3) - 1;
    const int wslab = ti >> 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (PLAIN) {
        a_off[i] = (((unsigned)ax[i] * (unsigned)g.lda + chunk * 8) * ESZ) | (0u - (unsigned)(ab[i] < 0));
        continue;
      }
      const int iz = az[i] + dz, iy = ay[i] + dy, ix = ax[i] + dx;
      const bool ok = ab[i] >= 0 && iz >= 0 && iz < g.IZ && iy >= 0 && iy < g.IY && ix >= 0 && ix < g.IX;
      const unsigned pix = (unsigned)(((ab[i] * g.PZ + (iz >> g.ups)) * g.PY + (iy >> g.ups)) * g.PX + (ix >> g.ups));
      // invalid rows are forced to 0xFFFFFFFF with a bit mask (no select -> no divergent branch)
      a_off[i] = ((pix * (unsigned)g.lda + chunk * 8) * ESZ) | (0u - (unsigned)(!ok));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + r0 + 32 * i;
      b_off[i] = ((((unsigned)wslab * N + n) * (unsigned)g.Cin + chunk * 8) * 2) | (0u - (unsigned)(n >= N));
    }
  };

  auto load_tiles = [&]() {
    // (re)derive the gather offsets BEFORE issuing this k-step's loads: nothing is in flight here, so the
    // scalar-load wait of set_tap cannot drain a tile load
    if (ld_tap != set_for) {
      set_tap(ld_tap);
      set_for = ld_tap;
    }
    const int cb = ld_cc * BK;
    const unsigned kmask = 0u - (unsigned)(cb + chunk * 8 >= g.Cin);  // K tail of narrow layers
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned inval = kmask | (0u - (unsigned)(a_off[i] == OOB));
      const unsigned oa = (a_off[i] + cb * ESZ) | inval;
      if constexpr (A_F32) {
        ra[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, oa, 0, 0);
        ra[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (a_off[i] + cb * ESZ + 16) | inval, 0, 0);
      } else {
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, oa, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const unsigned ob = (b_off[i] + cb * 2) | kmask | (0u - (unsigned)(b_off[i] == OOB));
      rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, ob, 0, 0);
    }
    if (++ld_cc == cpt) {
      ld_cc = 0;
      ++ld_tap;
    }
  };
  auto store_tiles = [&](int buf) {
    char* sA = smem + buf * STAGE;
    char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      if constexpr (A_F32) {
        const f32x4 lo = __builtin_bit_cast(f32x4, ra[2 * i]), hi = __builtin_bit_cast(f32x4, ra[2 * i + 1]);
        h8 v;
        v[0] = (half_t)lo[0]; v[1] = (half_t)lo[1]; v[2] = (half_t)lo[2]; v[3] = (half_t)lo[3];
        v[4] = (half_t)hi[0]; v[5] = (half_t)hi[1]; v[6] = (half_t)hi[2]; v[7] = (half_t)hi[3];
        *(h8*)(sA + swz(row, chunk)) = v;
      } else {
        *(i32x4*)(sA + swz(row, chunk)) = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) *(i32x4*)(sB + swz(r0 + 32 * i, chunk)) = rb[i];
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TLI(6);
  if (kbeg < kend) load_tiles();  // the first tile's loads go out before the bias loads below are waited for
  // bias (and, for plain GEMMs whose 32-row bands lie inside one sample, the per-sample bias) as the accumulators' initial
  // value: one load per fragment column here instead of dependent 16-byte loads between the epilogue's transposes
  bool folded = false;
  if (g.splitk <= 1 && g.alpha == 1.0f && !g.geglu && igemm_fast_epi(g) && (g.bias || (PLAIN && g.rowbias))) {
    const int rps = g.Z * g.Y * g.X;
    const bool rb_ok = PLAIN && g.rowbias && (rps & 31) == 0;
    if (!g.rowbias || rb_ok) {
      folded = true;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int bsu = rb_ok ? (int)(((float)min(m0 + wm * WM + i * 32, M - 1) + 0.5f) / (float)rps) : 0;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = min(n0 + wn * WN + j * 32 + (lane & 31), N - 1);
          const float b = (g.bias ? g.bias[n] : 0.f) + (rb_ok ? g.rowbias[(long)bsu * g.rb_ld + n] : 0.f);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
        }
      }
    }
  }

  if (kbeg < kend) store_tiles(0);
  __syncthreads();
  TLI(1);
  int cur = 0;
  for (int ks = kbeg; ks < kend; ++ks) {
    const bool more = ks + 1 < kend;
    if (more) load_tiles();
    const char* sA = smem + cur * STAGE;
    const char* sB = sA + A_BYTES;
    // fragment reads are software-pipelined one kk ahead of the MFMAs (two fragment register sets), so the
    // LDS latency of kk+1 hides behind the FM*FN MFMAs of kk instead of being exposed before every kk
    h8 af[2][FM], bf[2][FN];
    auto read_frags = [&](int kk, h8 (&a)[FM], h8 (&b)[FN]) {
      const int ch = kk * 2 + (lane >> 5);
#pragma unroll
      for (int f = 0; f < FM; ++f) a[f] = *(const h8*)(sA + swz(wm * WM + f * 32 + (lane & 31), ch));
#pragma unroll
      for (int f = 0; f < FN; ++f) b[f] = *(const h8*)(sB + swz(wn * WN + f * 32 + (lane & 31), ch));
    };
    read_frags(0, af[0], bf[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) read_frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads of kk+1 ahead of the MFMAs of kk
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = MVD_MFMA_32x32x16(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
    cur ^= 1;
    if (ks == kbeg) TLI(2);
    if (ks == kbeg + 1) TLI(3);
  }
  TLI(4);

  // ---- epilogue ----
  [[maybe_unused]] const float inv_rps = 1.0f / (float)(g.Z * g.Y * g.X);
  if (igemm_fast_epi(g)) {
    // all tile reads are done (the loop ends on a barrier): reuse the LDS as per-wave transpose scratch
    float* scratch = (float*)(smem + wave * EPI_WAVE_BYTES);
    float* part = g.splitk > 1 ? g.partial + (long)blockIdx.y * M * N : nullptr;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      int rows4[4], bs4[4];
      long orow4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * WM + fm * 32 + (lane >> 3) + 8 * i;
        rows4[i] = m < M ? m : -1;
        if constexpr (PLAIN) {
          orow4[i] = m < M ? m : 0;
          bs4[i] = g.rowbias ? (int)(((float)m + 0.5f) * inv_rps) : 0;  // exact for m < 2^22 (launch_igemm checks)
        } else {
          orow4[i] = m < M ? out_row(g, m) : 0;
          bs4[i] = -1;
        }
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
        epilogue_frag_store(g, acc[fm][fn], scratch, lane, rows4, orow4, n0 + wn * WN + fn * 32, part, PLAIN ? bs4 : nullptr,
                            folded);
    }
    TLI(5);
    return;
  }
  if constexpr (FN == 2 && WN == 64) {
    if (g.geglu && g.splitk <= 1 && (N & 63) == 0 && (g.ldc & 3) == 0) {  // vectorised GEGLU (FF1 of every block)
      float* sx = (float*)(smem + wave * 2 * EPI_WAVE_BYTES);
      float* sg = sx + EPI_WAVE_BYTES / 4;
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        int rows4[4];
        long orow4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = m0 + wm * WM + fm * 32 + (lane >> 3) + 8 * i;
          rows4[i] = m < M ? m : -1;
          orow4[i] = m < M ? out_row(g, m) : 0;
        }
        epilogue_geglu_frag_store(g, acc[fm][0], acc[fm][1], sx, sg, lane, rows4, orow4, n0 + wn * WN);
      }
      return;
    }
  }
  const int ncol0 = n0 + wn * WN + (lane & 31);
  if (g.splitk > 1) {
    float* part = g.partial + (long)blockIdx.y * M * N;
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WM + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
          const int n = ncol0 + fn * 32;
          if (n < N) part[(long)m * N + n] = acc[fm][fn][r];
        }
      }
    return;
  }
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * WM + fm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= M) continue;
      const long orow = out_row(g, m);
      if constexpr (FN == 2 && WN == 64) {
        if (g.geglu) {
          if (ncol0 + 32 < N) igemm_epilogue_store(g, m, orow, ncol0, acc[fm][0][r], acc[fm][1][r]);
          continue;
        }
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = ncol0 + fn * 32;
        if (n < N) igemm_epilogue_store(g, m, orow, n, acc[fm][fn][r], 0.f);
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const IGemm g) {
  const int M = g.B * g.Z * g.Y * g.X;
  const int N = g.N;
  const long total = (long)M * N;
  if (igemm_fast_epi(g)) {  // 16-byte path: 4 consecutive columns per thread
    // (row, column quad) advance incrementally (one 32-bit division per thread, none per element); the slabs of a quad are
    // loaded four at a time before they are added, in slab order (the sum is the same as a serial loop's)
    const int N4 = N >> 2, total4 = (int)(total >> 2), stride = gridDim.x * blockDim.x;
    int i4 = blockIdx.x * blockDim.x + threadIdx.x;
    int m = i4 / N4, q = i4 - m * N4;
    const int dm = stride / N4, dq = stride - dm * N4;
    const float inv_rps = 1.0f / (float)(g.Z * g.Y * g.X);
    const bool fast_bs = M < (1 << 22);
    for (; i4 < total4; i4 += stride) {
      const long idx = (long)i4 << 2;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int s0 = 0; s0 < g.splitk; s0 += 4) {
        float4 qv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int sj = min(s0 + j, g.splitk - 1);
          qv[j] = *(const float4*)(g.partial + (long)sj * total + idx);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (s0 + j < g.splitk) {
            v.x += qv[j].x; v.y += qv[j].y; v.z += qv[j].z; v.w += qv[j].w;
          }
      }
      epilogue_vec4(g, m, out_row(g, m), q << 2, v, fast_bs ? (int)(((float)m + 0.5f) * inv_rps) : -1);
      m += dm;
      q += dq;
      if (q >= N4) {
        q -= N4;
        ++m;
      }
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
    if (g.geglu && (n & 32)) continue;
    float v = 0.f, gate = 0.f;
    for (int s = 0; s < g.splitk; ++s) {
      v += g.partial[(long)s * total + idx];
      if (g.geglu) gate += g.partial[(long)s * total + idx + 32];
    }
    igemm_epilogue_store(g, m, out_row(g, m), n, v, gate);
  }
}

template <bool A_F32, int BN, int WAVES_M, int WAVES_N, bool PLAIN = false>
int launch_variant(const IGemm& g, int M, hipStream_t s) {
  constexpr int LDS = 2 * (BM * 128 + BN * 128);
  static bool attr_done[MVD_MAX_DEVICES] = {false};  // the attribute is per device
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)igemm_kernel<A_F32, BN, WAVES_M, WAVES_N, PLAIN>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  dim3 grid(cdiv(M, BM) * cdiv(g.N, BN), g.splitk > 1 ? g.splitk : 1);
  hipLaunchKernelGGL((igemm_kernel<A_F32, BN, WAVES_M, WAVES_N, PLAIN>), grid, dim3(NT), LDS, s, g);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

#ifdef MVD_TIMELINE
extern "C" int mvd_debug_igemm_timeline(unsigned long long* host_out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mvd_ig_tl), (size_t)n * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(mvd_ig_tl)) != hipSuccess) return -1;
  return hipMemset(p, 0, sizeof(mvd_ig_tl)) == hipSuccess ? 0 : -1;
}
#endif

int launch_splitk_reduce(const IGemm& g, hipStream_t s) {
  const long total = (long)g.B * g.Z * g.Y * g.X * g.N;
  int blocks = (int)((total / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, g);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// column-tile width that wastes the fewest MFMA columns for this N (GEGLU needs the 64-column wave tile)
int igemm_pick_bn(int N, int geglu) {
  if (geglu) return 128;
  if (N <= 64) return 64;
  const int w128 = cdiv(N, 128) * 128, w160 = cdiv(N, 160) * 160;
  return w160 < w128 ? 160 : 128;
}

int igemm_pick_splitk(int M, int N, int ksteps, int bn) {
  const int tiles = cdiv(M, BM) * cdiv(N, bn);
  if (tiles >= 192 || ksteps < 8) return 1;
  int sk = cdiv(256, tiles);  // one workgroup per CU (swept 128..1024 on the 2- and 16-views-per-rank steps)
  if (sk > ksteps / 4) sk = ksteps / 4;
  if (sk > 16) sk = 16;
  return sk < 1 ? 1 : sk;
}

int launch_igemm(const IGemm& g, hipStream_t s) {
  const int M = g.B * g.Z * g.Y * g.X;
  if (M <= 0 || g.N <= 0) return 0;
  if (g.Cin % 8) return mvd_fail("igemm: Cin must be a multiple of 8");
  if (g.ntaps < 1 || g.ntaps > MVD_MAX_TAPS) return mvd_fail("igemm: bad tap count");
  if (g.geglu && (g.N % 64)) return mvd_fail("igemm: GEGLU needs N % 64 == 0");
  if (g.splitk > 1 && !g.partial) return mvd_fail("igemm: split-K without a partial buffer");
  {  // 32-bit buffer offsets
    const long a_bytes = (long)g.B * g.PZ * g.PY * g.PX * g.lda * (g.a_f32 ? 4 : 2);
    int slabs = 1;
    for (int t = 0; t < g.ntaps; ++t) slabs = (g.tap[t] >> 8) + 1 > slabs ? (g.tap[t] >> 8) + 1 : slabs;
    const long w_bytes = (long)slabs * g.N * g.Cin * 2;
    if (a_bytes >= 0xFFFFFF00L || w_bytes >= 0xFFFFFF00L) return mvd_fail("igemm: operand exceeds 4 GiB buffer addressing");
  }
  const int bn = g.bn ? g.bn : igemm_pick_bn(g.N, g.geglu);
  int r;
  const bool plain = g.ntaps == 1 && g.tap[0] == igemm_tap(0, 0, 0, 0) && g.out_linear && g.ups == 0 && g.sz == 1 && g.sy == 1 &&
                     g.sx == 1 && g.PZ == g.Z && g.PY == g.Y && g.PX == g.X && g.IZ == g.Z && g.IY == g.Y && g.IX == g.X &&
                     M < (1 << 22) && !g.geglu;
  if (plain) {
    if (bn == 160) r = g.a_f32 ? launch_variant<true, 160, 4, 1, true>(g, M, s) : launch_variant<false, 160, 4, 1, true>(g, M, s);
    else if (bn == 64) r = g.a_f32 ? launch_variant<true, 64, 4, 1, true>(g, M, s) : launch_variant<false, 64, 4, 1, true>(g, M, s);
    else r = g.a_f32 ? launch_variant<true, 128, 2, 2, true>(g, M, s) : launch_variant<false, 128, 2, 2, true>(g, M, s);
    return r;
  }
  if (bn == 160) r = g.a_f32 ? launch_variant<true, 160, 4, 1>(g, M, s) : launch_variant<false, 160, 4, 1>(g, M, s);
  else if (bn == 64) r = g.a_f32 ? launch_variant<true, 64, 4, 1>(g, M, s) : launch_variant<false, 64, 4, 1>(g, M, s);
  else r = g.a_f32 ? launch_variant<true, 128, 2, 2>(g, M, s) : launch_variant<false, 128, 2, 2>(g, M, s);
  return r;  // split-K: the caller (igemm_go) runs launch_splitk_reduce
}
