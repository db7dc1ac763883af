// 3x3 stride-1 convolution with an LDS-resident input halo tile (the dominant kernel of the denoiser).
//
// Why: the generic implicit GEMM re-fetches every activation row once per tap and once per column tile; on
// MI355X that makes it cache-bandwidth bound (~8.5 TB/s delivered to the CUs, 13.7 B per kFLOP -> ~620 TF).
// Here a workgroup owns 256 output pixels (NI blocks of IH x IW pixels) and, per 64-channel chunk, stages the
// (IH+2) x (IW+2) input halo of each block in LDS ONCE; the nine taps are nine shifted views of that tile, so
// activation traffic drops ~6x and, with 256 rows per weight tile, weight traffic halves (4.7 B per kFLOP).
//
//   for cc in channel chunks:            halo(cc)  : LDS, double buffered, loaded during the taps of cc-1
//     for tap in 0..8:                   W(tap,cc) : LDS, double buffered, loaded during the previous step
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

__device__ __forceinline__ int swz3(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int BN, int IW, int IH>
__global__ __launch_bounds__(NT3, 2) void conv3_halo_kernel(const IGemm g) {
  constexpr int NI = BMP / (IW * IH);          // image blocks per tile
  constexpr int HW_ = IW + 2, HH = IH + 2;     // halo extent of one block
  constexpr int HPB = HW_ * HH;                // halo pixels per block
  constexpr int HP = NI * HPB;                 // halo pixels per tile
  constexpr int HSLOTS = (HP * 8 + NT3 - 1) / NT3;   // 16-byte halo chunks per thread
  constexpr int BSLOTS = (BN * 8 + NT3 - 1) / NT3;   // 16-byte weight chunks per thread
  constexpr int FN = BN / 32;
  constexpr int HALO_BYTES = HP * 128, W_BYTES = BN * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sHalo = smem;                     // [2][HP][128 B]
  char* sW = smem + 2 * HALO_BYTES;       // [2][BN][128 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = g.Y, W = g.X, N = g.N, Cin = g.Cin;
  const int bx_per = W / IW, by_per = H / IH, bpi = bx_per * by_per;  // blocks per image
  const int nblocks = g.B * bpi;
  const int tiles_m = (nblocks + NI - 1) / NI, tiles_n = (N + BN - 1) / BN;
  int bid = blockIdx.x;
  {  // XCD-aware bijective remap: consecutive tiles (same pixels, next column tile) share an L2
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

  // ---- per-thread halo slots: global byte offset of (pixel, 16-byte chunk) at channel 0, or OOB ----
  unsigned h_off[HSLOTS];
  int h_lds[HSLOTS];
#pragma unroll
  for (int i = 0; i < HSLOTS; ++i) {
    const int q = tid + i * NT3;
    const int hp = q >> 3, chunk = q & 7;
    const int j = hp / HPB, hr = hp - j * HPB;
    const int hy = hr / HW_, hx = hr - hy * HW_;
    const int gb = tm * NI + j;
    const int b = gb / bpi, rem = gb - b * bpi;
    const int y = (rem / bx_per) * IH + hy - 1, x = (rem % bx_per) * IW + hx - 1;
    const bool ok = hp < HP && b < g.B && y >= 0 && y < H && x >= 0 && x < W;
    const unsigned pix = (unsigned)((b * H + y) * W + x);
    h_off[i] = ((pix * (unsigned)g.lda + chunk * 8) * 2) | (0u - (unsigned)(!ok));
    h_lds[i] = hp < HP ? swz3(hp, chunk) : -1;
  }
  // ---- per-thread weight slots ----
  unsigned w_off[BSLOTS];
  int w_lds[BSLOTS];
#pragma unroll
  for (int i = 0; i < BSLOTS; ++i) {
    const int q = tid + i * NT3;
    const int r = q >> 3, chunk = q & 7;
    const int n = n0 + r;
    const bool ok = r < BN && n < N;
    w_off[i] = (((unsigned)n * (unsigned)Cin + chunk * 8) * 2) | (0u - (unsigned)(!ok));
    w_lds[i] = r < BN ? swz3(r, chunk) : -1;
  }
  const unsigned tap_stride = (unsigned)N * (unsigned)Cin * 2;  // bytes between weight taps

  i32x4 rh[HSLOTS], rw[BSLOTS];
  auto load_halo = [&](int cc) {
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i) {
      const unsigned inval = 0u - (unsigned)(h_off[i] == 0xFFFFFFFFu);
      rh[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, (h_off[i] + cc * 128) | inval, 0, 0);
    }
  };
  auto store_halo = [&](int buf) {
#pragma unroll
    for (int i = 0; i < HSLOTS; ++i)
      if (h_lds[i] >= 0) *(i32x4*)(sHalo + buf * HALO_BYTES + h_lds[i]) = rh[i];
  };
  auto load_w = [&](int tap, int cc) {
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
      const unsigned inval = 0u - (unsigned)(w_off[i] == 0xFFFFFFFFu);
      rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrcW, (w_off[i] + tap * tap_stride + cc * 128) | inval, 0, 0);
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i)
      if (w_lds[i] >= 0) *(i32x4*)(sW + buf * W_BYTES + w_lds[i]) = rw[i];
  };

  // this lane's A-fragment row: tile pixel p -> centre index inside the halo
  const int p = wave * 32 + (lane & 31);
  const int pj = p / (IW * IH), pr = p - pj * (IW * IH);
  const int centre = pj * HPB + (pr / IW + 1) * HW_ + (pr % IW) + 1;

  f32x16 acc[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nsteps = (cc_end - cc_beg) * 9;
  if (nsteps > 0) {
    load_halo(cc_beg);
    load_w(0, cc_beg);
    store_halo(0);
    store_w(0);
  }
  __syncthreads();
  int cc = cc_beg, tap = 0, hbuf = 0;
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    int ntap = tap + 1, ncc_ = cc;
    if (ntap == 9) { ntap = 0; ncc_ = cc + 1; }
    if (more) load_w(ntap, ncc_);
    const bool pf_halo = tap == 0 && cc + 1 < cc_end;   // prefetch the next chunk's halo behind 9 taps of MFMAs
    if (pf_halo) load_halo(cc + 1);

    const char* hb = sHalo + hbuf * HALO_BYTES;
    const char* wb = sW + (s & 1) * W_BYTES;
    const int hrow = centre + (tap / 3 - 1) * HW_ + (tap % 3 - 1);
    h8 af[2], bf[2][FN];
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
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk & 1], bf[kk & 1][j], acc[j], 0, 0, 0);
    }
    if (more) store_w((s + 1) & 1);
    if (tap == 8 && cc + 1 < cc_end) store_halo(hbuf ^ 1);
    __syncthreads();
    if (ntap == 0) hbuf ^= 1;
    tap = ntap;
    cc = ncc_;
  }

  // ---- epilogue: C layout row = (r&3) + 8(r>>2) + 4(lane>>5) inside the wave's 32 pixels ----
  const int M = g.B * H * W;
  const int ncol0 = n0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int pp = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int j = pp / (IW * IH), rr = pp - j * (IW * IH);
    const int gb = tm * NI + j;
    const int b = gb / bpi, rem = gb - b * bpi;
    if (b >= g.B) continue;
    const int y = (rem / bx_per) * IH + rr / IW, x = (rem % bx_per) * IW + rr % IW;
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
}

template <int BN, int IW, int IH>
int launch_c3(const IGemm& g, hipStream_t s) {
  constexpr int NI = BMP / (IW * IH);
  constexpr int HP = NI * (IW + 2) * (IH + 2);
  constexpr int LDS = 2 * HP * 128 + 2 * BN * 128;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv3_halo_kernel<BN, IW, IH>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int nblocks = g.B * (g.Y / IH) * (g.X / IW);
  dim3 grid(cdiv(nblocks, NI) * cdiv(g.N, BN), g.splitk > 1 ? g.splitk : 1);
  hipLaunchKernelGGL((conv3_halo_kernel<BN, IW, IH>), grid, dim3(NT3), LDS, s, g);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

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
  if (big) return bn == 160 ? launch_c3<160, 16, 16>(g, s) : launch_c3<128, 16, 16>(g, s);
  return bn == 160 ? launch_c3<160, 8, 8>(g, s) : launch_c3<128, 8, 8>(g, s);
}
