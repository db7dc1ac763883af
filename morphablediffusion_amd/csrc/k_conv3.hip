// 3x3 stride-1 convolution with an LDS-resident input halo tile (the dominant kernel of the denoiser).
//
// Why: the generic implicit GEMM re-fetches every activation row once per tap and once per column tile; on
// MI355X that makes it cache-bandwidth bound (~8.5 TB/s delivered to the CUs, 13.7 B per kFLOP -> ~620 TF).
// Here a workgroup owns 256 output pixels (NI blocks of IH x IW pixels) and, per 64-channel chunk, stages the
// (IH+2) x (IW+2) input halo of each block in LDS ONCE; the nine taps are nine shifted views of that tile, so
// activation traffic drops ~6x and, with 256 rows per weight tile, weight traffic halves (4.7 B per kFLOP).
//
//   for cc in channel chunks:            halo(cc)  : LDS, double buffered, loaded during the taps of cc-1
//     for tap in 0..8:                   W(tap,cc) : LDS, 3-stage ring, loaded two steps ahead
//        4 x (1 A fragment from the shifted halo, BN/32 W fragments, BN/32 MFMAs 32x32x16 f16)
//
// 512 threads = 8 waves, wave w owns tile pixels [32w, 32w+32) x all BN columns.  Zero padding at the image
// border comes from buffer loads with out-of-range offsets (hardware returns 0), so the halo fill is
// branch-free.  LDS rows are 128 B (64 halfs) with the 16-byte chunk XOR-swizzled by (row>>1)&7.
// The epilogue (bias / per-sample bias / residual / SiLU, fp16 or fp32 store, split-K partials) is the
// implicit GEMM's.
#include "common.h"
#include "igemm_epilogue.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BMP = 256, NT3 = 512;

#ifdef MVD_TIMELINE
// investigation build only (make EXTRA=-DMVD_TIMELINE): per-workgroup phase timestamps of the last launch
__device__ unsigned long long mvd_c3_tl[8 * 4096];
#define TL3(i)                                                                                              \
  do {                                                                                                      \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) mvd_c3_tl[blockIdx.x * 8 + (i)] = wall_clock64(); \
  } while (0)
#else
#define TL3(i)
#endif

[[maybe_unused]] __device__ __forceinline__ int swz3(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---------------------------------------------------------------------------------------------------------
// The halo and the weight tiles go HBM/L2 -> LDS directly (buffer_load ... lds, 1 KiB = 8
// swizzled rows per wave-instruction), no VGPR staging and no ds_write.  Weights run through a WST-stage ring
// so TWO steps of weight loads are in flight behind every MFMA block; completion is counted by hand
// (s_waitcnt vmcnt(N) + raw s_barrier), because only the issuing wave's vmcnt orders an LDS-DMA.
// Wave w issues 8-row groups w, w+8, ...; waves may differ by one instruction per tile, so the waits use the
// minimum per-wave count (a wave with one more instruction merely waits for its oldest prefetch too).
// ---------------------------------------------------------------------------------------------------------
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int BN, int IW, int IH>
__global__ __launch_bounds__(NT3, 1) void conv3_dma_kernel(const IGemm g) {
#if defined(__HIP_DEVICE_COMPILE__)  // LDS-DMA builtins exist only in the gfx950 device pass; the host pass needs just the stub
  constexpr int NI = BMP / (IW * IH);
  constexpr int HW_ = IW + 2, HH = IH + 2, HPB = HW_ * HH, HP = NI * HPB;
  constexpr int HG = (HP + 7) / 8;             // halo groups of 8 rows (one DMA instruction each)
  constexpr int WGR = BN / 8;                  // weight groups per stage
  constexpr int NH = (HG + 7) / 8, NH_MIN = HG / 8;
  constexpr int NW = (WGR + 7) / 8, NW_MIN = WGR / 8;
  constexpr int WST = 3;
  constexpr int FN = BN / 32;
  constexpr int HALO_BYTES = HG * 1024, W_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sHalo = smem;                       // [2][HG*8 rows][128 B]
  char* sW = smem + 2 * HALO_BYTES;         // [WST][BN][128 B]
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  TL3(0);
  const int H = g.Y, W = g.X, N = g.N, Cin = g.Cin;
  const int bx_per = W / IW, by_per = H / IH, bpi = bx_per * by_per;
  const int nblocks = g.B * bpi;
  const int tiles_m = (nblocks + NI - 1) / NI, tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int n0 = tn * BN;
  const int ncc = Cin / 64;
  int cc_beg = 0, cc_end = ncc;
  if (g.splitk > 1) {
    const int per = (ncc + g.splitk - 1) / g.splitk;
    cc_beg = blockIdx.y * per;
    cc_end = min(ncc, cc_beg + per);
  }
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.a), (short)0, 0xFFFFFFFEu, 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(g.w), (short)0, 0xFFFFFFFEu, 0x00020000);
  constexpr unsigned OOB = 0xFFFFFFFFu;  // out of range for the resource: the hardware returns zeros, no memory access

  // per-lane source offsets: LDS position (row, pos = lane&7) receives source chunk pos ^ ((row>>1)&7)
  // block -> (sample, block row, block column) with a float reciprocal instead of integer divisions (exact far beyond the
  // block counts that occur; a kernel's first pass over its code runs at instruction-fetch speed, and each runtime
  // integer division is ~35 instructions of it)
  const float inv_bpi = 1.0f / (float)bpi, inv_bxp = 1.0f / (float)bx_per;
  auto block_pos = [&](int gb, int& b, int& by, int& bx) {
    b = (int)(((float)gb + 0.5f) * inv_bpi);
    const int rem = gb - b * bpi;
    by = (int)(((float)rem + 0.5f) * inv_bxp);
    bx = rem - by * bx_per;
  };
  unsigned h_off[NH], w_off[NW];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    const int hp = (wave + 8 * i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((hp >> 1) & 7);
    const int j = hp / HPB, hr = hp - j * HPB;
    const int hy = hr / HW_, hx = hr - hy * HW_;
    int b, by, bx;
    block_pos(tm * NI + j, b, by, bx);
    const int y = by * IH + hy - 1, x = bx * IW + hx - 1;
    const bool ok = hp < HP && b < g.B && y >= 0 && y < H && x >= 0 && x < W;
    const unsigned pix = (unsigned)((b * H + y) * W + x);
    h_off[i] = ok ? (pix * (unsigned)g.lda + chunk * 8) * 2 : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const int r = (wave + 8 * i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((r >> 1) & 7);
    const int n = n0 + r;
    const bool ok = r < BN && n < N;
    w_off[i] = ok ? ((unsigned)n * (unsigned)Cin + chunk * 8) * 2 : 0xFFFFFFFFu;
  }
  const unsigned tap_stride = (unsigned)N * (unsigned)Cin * 2;

  auto dma_halo = [&](int cc, int buf) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int grp = wave + 8 * i;
      if (grp < HG) {
        const unsigned off = h_off[i] == 0xFFFFFFFFu ? OOB : h_off[i] + cc * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr)(sHalo + buf * HALO_BYTES + grp * 1024), 16, off, 0, 0, 0);
      }
    }
  };
  auto dma_w = [&](int tap, int cc, int stage) {
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int grp = wave + 8 * i;
      if (grp < WGR) {
        const unsigned off = w_off[i] == 0xFFFFFFFFu ? OOB : w_off[i] + tap * tap_stride + cc * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr)(sW + stage * W_BYTES + grp * 1024), 16, off, 0, 0, 0);
      }
    }
  };

  const int p = (tid >> 6) * 32 + (lane & 31);
  const int pj = p / (IW * IH), pr = p - pj * (IW * IH);
  const int centre = pj * HPB + (pr / IW + 1) * HW_ + (pr % IW) + 1;

  f32x16 acc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // bias + per-sample bias as the accumulators' initial value (alpha = 1, no split-K; every block of NI image blocks that
  // belongs to one sample qualifies -- a lane owns one column per fragment, so two loads per fragment, issued before the
  // first tile load and summed only after the first counted wait; see k_gemm.hip)
  bool folded = false;
  float binit[FN], rinit[FN];
  {
    int b_first, b_last, t0, t1;
    block_pos(tm * NI, b_first, t0, t1);
    block_pos(min(tm * NI + NI - 1, nblocks - 1), b_last, t0, t1);
    folded = g.splitk <= 1 && g.alpha == 1.0f && igemm_fast_epi(g) && (g.bias || g.rowbias) && b_first == b_last;
    if (folded) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = min(n0 + fn * 32 + (lane & 31), N - 1);
        binit[fn] = g.bias ? g.bias[n] : 0.f;
        rinit[fn] = g.rowbias ? g.rowbias[(long)b_first * g.rb_ld + n] : 0.f;
      }
    }
  }
  const int nsteps = (cc_end - cc_beg) * 9;
  TL3(6);
  if (nsteps > 0) {
    dma_halo(cc_beg, 0);
    dma_w(0, cc_beg, 0);
    if (nsteps > 1) {
      dma_w(1, cc_beg, 1);
      wait_vm<NW_MIN>();
    } else {
      wait_vm<0>();
    }
  }
  __builtin_amdgcn_s_barrier();
  TL3(1);
  if (folded) {
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[fn][r] = binit[fn] + rinit[fn];
  }
  int cc = cc_beg, tap = 0, hbuf = 0, stage = 0;
  h8 af[2], bf[2][FN];
  for (int s = 0; s < nsteps; ++s) {
    // prefetch (issued inside the kk loop): weights two steps ahead into the ring slot that was read at step s-1;
    // halo one chunk ahead
    const bool pf_w = s + 2 < nsteps;
    const bool pf_h = tap == 0 && cc + 1 < cc_end;
    const char* hb = sHalo + hbuf * HALO_BYTES;
    const char* wb = sW + stage * W_BYTES;
    const int hrow = centre + (tap / 3 - 1) * HW_ + (tap % 3 - 1);
    auto read_frags = [&](int kk, h8& a, h8 (&b)[FN]) {
      const int ch = kk * 2 + (lane >> 5);
      a = *(const h8*)(hb + swz3(hrow, ch));
#pragma unroll
      for (int f = 0; f < FN; ++f) b[f] = *(const h8*)(wb + swz3(f * 32 + (lane & 31), ch));
    };
    read_frags(0, af[0], bf[0]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk < 3) read_frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[j] = MVD_MFMA_32x32x16(af[kk & 1], bf[kk & 1][j], acc[j], 0, 0, 0);
      // issue order within the kk block: one LDS read of kk+1 between consecutive MFMAs of kk, so the eight waves
      // (in lockstep after the step barrier) do not hit the LDS with 48 reads at once
      if (kk < 3) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk == 1) {
        // this step's prefetches are issued mid-step, after the fragment reads of the first two kk blocks are under way
        // (issuing them at the top of the step delays those reads behind the DMA address set-up: 68.2 -> 64.4 us for
        // the level-32 conv; later placements are slower again)
        if (pf_h) dma_halo(cc + 1, hbuf ^ 1);
        if (pf_w) {
          int t2 = tap + 2, c2 = cc;
          if (t2 >= 9) { t2 -= 9; c2 += 1; }
          int st2 = stage + 2;
          if (st2 >= WST) st2 -= WST;
          dma_w(t2, c2, st2);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // everything issued BEFORE this step (W(s+1), earlier halos) must have landed; only this step's own
    // prefetches may stay in flight across the barrier
    if (pf_w) {
      if (pf_h) wait_vm<NW_MIN + NH_MIN>();
      else wait_vm<NW_MIN>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (s == 0) TL3(2);
    if (s == 1) TL3(3);
    if (++tap == 9) {
      tap = 0;
      ++cc;
      hbuf ^= 1;
    }
    if (++stage == WST) stage = 0;
  }

  TL3(4);
  const int M = g.B * H * W;
  if (igemm_fast_epi(g)) {
    float* scratch = (float*)(smem + (tid >> 6) * EPI_WAVE_BYTES);
    float* part = g.splitk > 1 ? g.partial + (long)blockIdx.y * M * N : nullptr;
    int rows4[4], bs4[4];
    long orow4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pp = (tid >> 6) * 32 + (lane >> 3) + 8 * i;
      const int j = pp / (IW * IH), rr = pp - j * (IW * IH);
      int b, by, bx;
      block_pos(tm * NI + j, b, by, bx);
      const int y = by * IH + rr / IW, x = bx * IW + rr % IW;
      rows4[i] = b < g.B ? (b * H + y) * W + x : -1;
      orow4[i] = rows4[i];
      bs4[i] = b < g.B ? b : 0;
    }
    if (part) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) epilogue_frag_store(g, acc[fn], scratch, lane, rows4, orow4, n0 + fn * 32, part);
    } else {
      float4 pre[FN][4];  // all bias / residual loads of the tile in flight before the first transpose
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) epilogue_prefetch(g, lane, rows4, orow4, n0 + fn * 32, pre[fn], bs4, folded);
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) epilogue_frag_store_pre(g, acc[fn], scratch, lane, rows4, orow4, n0 + fn * 32, pre[fn]);
    }
    TL3(5);
    return;
  }
  const int ncol0 = n0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pp = (tid >> 6) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int j = pp / (IW * IH), rr = pp - j * (IW * IH);
    int b, by, bx;
    block_pos(tm * NI + j, b, by, bx);
    if (b >= g.B) continue;
    const int y = by * IH + rr / IW, x = bx * IW + rr % IW;
    const int m = (b * H + y) * W + x;
    if (g.splitk > 1) {
      float* part = g.partial + (long)blockIdx.y * M * N;
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = ncol0 + fn * 32;
        if (n < N) part[(long)m * N + n] = acc[fn][r];
      }
    } else {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int n = ncol0 + fn * 32;
        if (n < N) igemm_epilogue_store(g, m, m, n, acc[fn][r], 0.f);
      }
    }
  }
#endif
}

template <int BN, int IW, int IH>
int launch_c3_dma(const IGemm& g, hipStream_t s) {
  constexpr int NI = BMP / (IW * IH);
  constexpr int HP = NI * (IW + 2) * (IH + 2);
  constexpr int LDS = 2 * ((HP + 7) / 8) * 1024 + 3 * BN * 128;
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static bool attr_done[MVD_MAX_DEVICES] = {false};  // the attribute is per device
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv3_dma_kernel<BN, IW, IH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int nblocks = g.B * (g.Y / IH) * (g.X / IW);
  dim3 grid(cdiv(nblocks, NI) * cdiv(g.N, BN), g.splitk > 1 ? g.splitk : 1);
  hipLaunchKernelGGL((conv3_dma_kernel<BN, IW, IH>), grid, dim3(NT3), LDS, s, g);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

#ifdef MVD_TIMELINE
extern "C" int mvd_debug_conv3_timeline(unsigned long long* host_out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mvd_c3_tl), (size_t)n * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(mvd_c3_tl)) != hipSuccess) return -1;
  return hipMemset(p, 0, sizeof(mvd_c3_tl)) == hipSuccess ? 0 : -1;
}
#endif

// eligibility: fp16 channels-last input, 3x3 stride 1 pad 1, no fused upsample, Cin % 64 == 0,
// square-ish power-of-two images of side >= 8
bool conv3_halo_eligible(const IGemm& g) {
  if (g.a_f32 || g.ntaps != 9 || g.sy != 1 || g.sx != 1 || g.ups || g.Z != 1 || g.geglu || !g.out_linear) return false;
  if (g.Cin % 64) return false;
  const int H = g.Y, W = g.X;
  if (H != g.IY || W != g.IX) return false;
  const bool ok16 = (H % 16 == 0 && W % 16 == 0), ok8 = (H == 8 && W == 8);
  return ok16 || ok8;
}

int conv3_halo_tiles(const IGemm& g, int bn) {
  const int IWH = (g.X % 16 == 0) ? 16 : 8;
  const int NI = BMP / (IWH * IWH);
  return cdiv(g.B * (g.Y / IWH) * (g.X / IWH), NI) * cdiv(g.N, bn);
}

int launch_conv3_halo(const IGemm& g, hipStream_t s) {
  const bool big = g.X % 16 == 0;
  const int bn = g.bn == 160 ? 160 : 128;
  if (g.splitk > 1 && !g.partial) return mvd_fail("conv3_halo: split-K without a partial buffer");
  if (big) return bn == 160 ? launch_c3_dma<160, 16, 16>(g, s) : launch_c3_dma<128, 16, 16>(g, s);
  return bn == 160 ? launch_c3_dma<160, 8, 8>(g, s) : launch_c3_dma<128, 8, 8>(g, s);
}
