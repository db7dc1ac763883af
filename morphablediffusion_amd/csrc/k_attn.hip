// Spatial self-attention (per view, 8 heads, T = h*w tokens <= 1024) as a flash-style kernel on the matrix
// cores.  Both products are computed TRANSPOSED so that everything that belongs to one query lives in one
// lane:
//     S^T = K Q^T        (A = K rows from LDS, B = Q fragment held in registers)
//     O^T = V^T P^T      (A = V^T rows from LDS, B = P^T straight from the S^T accumulators)
// In the 32x32 accumulator layout lane l owns query column q = l&31 and 16 key rows
// {(r&3) + 8(r>>2) + 4(l>>5)}; the softmax max/sum over keys is therefore 16 in-lane ops plus ONE
// cross-half exchange, and the running rescale of O^T is lane-local.  The fp16 P^T fragment that feeds
// the second MFMA is exactly the accumulator register order, provided V^T is read with the same key
// permutation (two 8-byte LDS reads per fragment instead of one 16-byte read) -- no shuffles, no LDS
// round trip for P.  q, k and v come out of ONE fused projection GEMM ([token][q | k | v]); the V^T image in LDS is
// built by the staging pass (16-byte row-major loads, 2-byte transposing LDS stores) -- that costs the kernel ~10 % at
// d = 40 and saves a swapped V^T = W_v X^T GEMM launch per attention (step: +1.6 %, 2-views-per-rank step: +3 %).
#include "common.h"

namespace {

template <int D>
__global__ __launch_bounds__(256, (D <= 40 ? 4 : D <= 80 ? 3 : 2)) void attn_kernel(const half_t* __restrict__ qk, int ldqk,
                                                   const half_t* __restrict__ v, int ldv,
                                                   half_t* __restrict__ out, int ldo, int T, int heads, float scale,
                                                   int Tstride, int xcd_remap) {
  constexpr int KS = (D + 15) / 16;        // k-steps of the QK^T product
  constexpr int DK = KS * 16;
  constexpr int DVF = (D + 31) / 32;       // 32-row fragments of O^T
  constexpr int DVP = DVF * 32;
  constexpr int KLD = DK + 8;              // halfs per K row in LDS
  // V stays ROW-MAJOR in LDS ([key][d], one 16-byte store per staged chunk) and the V^T fragments of the second MFMA come out of
  // gfx950's transposing LDS read: in a 16-lane group lane i supplies the 8-byte address of row i / 4, columns 4 (i % 4) .. of a
  // [4 keys][16 columns] block and receives column i, keys 0 .. 3 (tools/tr_b16_probe.hip) -- two reads give a lane the 8 keys of
  // its d-column in exactly the order the P^T fragment uses.  Rounds 1-5 built a transposed image with eight 2-byte stores per
  // chunk instead: bank-conflicted stores (5-way at a 72-half pitch, 3-way at 68) on top of 16 store instructions per thread and
  // key tile.  Row pitch VLD: half of it = 16 or 48 (mod 64) dwords puts the 4 key rows of a 32-lane group on 4 disjoint
  // 16-bank ranges (tools/lds_bank_model.py).
  constexpr int VLD = DVP <= 32 ? 32 : (DVP <= 96 ? 96 : 160);
  // d = 8, 16, 40, 80: the last 32-row fragment of O^T has padding rows.  Column D of the V image is then all ONES, so row D of
  // O^T accumulates sum_k P[q][k] -- the softmax denominator comes out of the second MFMA (over exactly the fp16 P the numerator
  // uses, rescaled with O^T for free) instead of 32 adds + a cross-half exchange per key tile of this VALU-bound kernel.
  constexpr bool ROWSUM = DVP > D;
  constexpr int RS_F = D / 32, RS_ROW = D - 32 * RS_F;                                  // fragment and row of the sum
  constexpr int RS_HH = (RS_ROW >> 2) & 1, RS_R = (RS_ROW & 3) + 4 * (RS_ROW >> 3);    // half-wave and register holding it
  __shared__ __attribute__((aligned(16))) half_t sK[64 * KLD];
  __shared__ __attribute__((aligned(16))) half_t sV[64 * VLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = lane >> 5, lq = lane & 31;
  // One-dimensional grid [sample][head][query tile], query tile fastest.  The hardware deals consecutive workgroups to the 8 XCDs
  // round-robin: with T = 1024 (8 query tiles) every tile of a (sample, head) pair landed on a different XCD and each of the 8
  // L2s fetched that head's K and V for itself (counter traffic 3.2 x the algorithmic bytes).  The bijective remap below gives
  // an XCD a contiguous range of the linear index, so the query tiles of one head share one L2 (same arithmetic per workgroup:
  // bit-identical results).
  int bid = blockIdx.x;
  if (xcd_remap) {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  }
  const int nqt = (T + 127) >> 7;
  const int bh = bid / nqt, qt = bid - bh * nqt;
  const int b = bh / heads, head = bh - b * heads;
  const int C = heads * D;
  const int q_row = qt * 128 + wave * 32 + lq;
  const long tok0 = (long)b * Tstride;  // samples are Tstride rows apart (Tstride > T: padded token axis)

  // Q fragment (B operand): lane (q, hh) holds d = ks*16 + hh*8 .. +8
  h8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int d0 = ks * 16 + hh * 8;
    if (q_row < T && d0 < D) qf[ks] = *(const h8*)(qk + (tok0 + q_row) * ldqk + head * D + d0);
    else qf[ks] = (h8)(half_t)0;
  }

  f32x16 o[DVF];
#pragma unroll
  for (int f = 0; f < DVF; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[f][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // K / V tiles are register-staged one tile ahead: the global loads of tile i+1 are in flight behind the MFMAs and the
  // softmax of tile i.  V arrives row-major ([token][head*d + dv], the third block of the fused q|k|v projection) and is
  // transposed by the LDS write pass (eight 2-byte stores per 16-byte load) into the V^T image the second MFMA reads.
  // A thread's slots -- (key, 8-channel chunk) pairs, D/8 chunks per key for K and for V -- its source pointers and its LDS
  // addresses are fixed for the whole kernel: they are derived ONCE here and the loop only advances the pointers by 64 rows
  // (round 4: with the index arithmetic and the bounds tests inside the loop they were ~180 of the ~450 VALU instructions
  // per key tile of this VALU-bound kernel).  The padding chunk of K (d = D .. DK-1) is zeroed once and never rewritten.
  constexpr int CH = D / 8, NSLOT = (64 * CH + 255) / 256;
  h8 kreg[NSLOT], vreg[NSLOT];
  const half_t *gk[NSLOT], *gv[NSLOT];
  int skey[NSLOT], sch[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int idx = tid + i * 256;
    const bool ok = idx < 64 * CH;
    const int key = ok ? idx / CH : 0, ch = ok ? idx - key * CH : 0;
    skey[i] = ok ? key : -1;
    sch[i] = ch;
    gk[i] = qk + (tok0 + key) * ldqk + C + head * D + ch * 8;
    gv[i] = v + (tok0 + key) * ldv + head * D + ch * 8;
  }
  auto load_tiles = [&](int k0) {
    if (k0 + 64 <= T) {  // wave-uniform.  Full tile: every live slot is loaded (dead slots are never stored), nothing to clear
#pragma unroll
      for (int i = 0; i < NSLOT; ++i)
        if (skey[i] >= 0) {
          kreg[i] = *(const h8*)gk[i];
          vreg[i] = *(const h8*)gv[i];
        }
    } else {
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        kreg[i] = (h8)(half_t)0;
        vreg[i] = (h8)(half_t)0;  // keys >= T must be zeros: their P is 0, but 0 x garbage could be NaN
        if (skey[i] >= 0 && k0 + skey[i] < T) {
          kreg[i] = *(const h8*)gk[i];
          vreg[i] = *(const h8*)gv[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      gk[i] += 64 * (long)ldqk;
      gv[i] += 64 * (long)ldv;
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
      if (skey[i] >= 0) {
        *(h8*)(sK + skey[i] * KLD + sch[i] * 8) = kreg[i];
        *(h8*)(sV + skey[i] * VLD + sch[i] * 8) = vreg[i];
      }
    }
  };
  // columns D..DVP-1 of the V image are padding of the last 32-row fragment of O^T, columns D..DK-1 of the K image padding of the
  // last k-step: written once, never rewritten
  if constexpr (DVP > D) {
    for (int idx = tid; idx < (DVP - D) * 64; idx += 256) {
      const int key = idx / (DVP - D), col = D + idx - key * (DVP - D);
      sV[key * VLD + col] = col == D ? (half_t)1 : (half_t)0;  // (ROWSUM == DVP > D)
    }
  }
  if constexpr (DK > D)
    for (int idx = tid; idx < 64 * (DK - D); idx += 256) sK[(idx / (DK - D)) * KLD + D + idx % (DK - D)] = (half_t)0;
  // this lane's address inside a [4 keys][16 columns] block of the transposing read: lane i of a 16-lane group -> key row i / 4,
  // columns 4 (i % 4) ..; the group's block: keys 4 hh .., columns 16 ((lane >> 4) & 1) ..
  const half_t* vbase = sV + (4 * hh + ((lane & 15) >> 2)) * VLD + ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
  auto tr_read = [](const half_t* p) -> h4 {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p));
    return __builtin_bit_cast(h4, r);
  };
  load_tiles(0);
  // The Q fragments came from global loads issued before the loop.  Without a use in front of the loop their first use is the
  // first MFMA INSIDE it, and the compiler's wait-count pass (which merges the loop's entry and back edge) then puts an
  // `s_waitcnt vmcnt(0)` in front of that MFMA in EVERY iteration -- right behind the loads of the next K / V tile, whose latency
  // the one-tile-ahead staging exists to hide.  The empty asm consumes the fragments here: the wait lands in front of the loop.
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
  for (int k0 = 0; k0 < T; k0 += 64) {
    __syncthreads();  // every wave is done reading the previous tile
    store_tiles();
    __syncthreads();
    if (k0 + 64 < T) load_tiles(k0 + 64);

    f32x16 s[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[f][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const h8 kf = *(const h8*)(sK + (f * 32 + lq) * KLD + ks * 16 + hh * 8);
        s[f] = MVD_MFMA_32x32x16(kf, qf[ks], s[f], 0, 0, 0);
      }
    }
    // online softmax over the key axis (rows of S^T).  The running maximum is tracked on the scaled scores; the
    // scale (> 0) is applied inside the exponent's fma, so a score costs max + fma + exp2 + add
    float mx = -INFINITY;
    if (k0 + 64 > T) {  // last, partial tile: keys >= T are masked out
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = k0 + f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= T) s[f][r] = -INFINITY;
        }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx * scale);
    float psum = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], scale, -m_new));  // scores carry log2(e)
        s[f][r] = p;
        if constexpr (!ROWSUM) psum += p;
      }
    if constexpr (!ROWSUM) psum += __shfl_xor(psum, 32);
    if (__builtin_amdgcn_ballot_w64(m_new != m_run)) {  // wave-uniform: once the maxima settle nothing is rescaled
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int f = 0; f < DVF; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[f][r] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
    // O^T += V^T P^T : 4 k-steps of 16 keys
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      h8 pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb[j] = (half_t)s[kk >> 1][8 * (kk & 1) + j];
#pragma unroll
      for (int f = 0; f < DVF; ++f) {
        const half_t* vblk = vbase + (kk * 16) * VLD + f * 32;
        const h4 v0 = tr_read(vblk);            // keys 16 kk + 4 hh + 0..3 of d-column f * 32 + lq
        const h4 v1 = tr_read(vblk + 8 * VLD);  // keys 16 kk + 8 + 4 hh + 0..3
        h8 va;
        va[0] = v0[0]; va[1] = v0[1]; va[2] = v0[2]; va[3] = v0[3];
        va[4] = v1[0]; va[5] = v1[1]; va[6] = v1[2]; va[7] = v1[3];
        o[f] = MVD_MFMA_32x32x16(va, pb, o[f], 0, 0, 0);
      }
    }
  }

  if constexpr (ROWSUM) {  // the denominator sits in register RS_R of fragment RS_F, in the RS_HH half of the wave
    const float mine = o[RS_F][RS_R], other = __shfl_xor(mine, 32);
    l_run = hh == RS_HH ? mine : other;
  }
  if (q_row < T) {
    const float inv = 1.0f / l_run;
    half_t* orow = out + (tok0 + q_row) * ldo + head * D;
#pragma unroll
    for (int f = 0; f < DVF; ++f)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int dv = f * 32 + 8 * rg + 4 * hh;
        if (dv < D) {
          h4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (half_t)(o[f][rg * 4 + j] * inv);
          *(h4*)(orow + dv) = v;
        }
      }
  }
}

// rows of fp32 scores -> fp16 probabilities (single-head attention of the first-stage decoder: the scores come from a
// GEMM because d = 512 does not fit the flash kernel's register tile); one workgroup per row, cols <= 4096
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, int cols, half_t* __restrict__ p) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* sr = s + row * cols;
  half_t* pr = p + row * cols;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = c < cols ? sr[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = __expf(v[i] - mx);
    sum += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < cols) pr[c] = (half_t)(v[i] * inv);
  }
}

// fp32 -> fp32 variant (in place allowed): the extended-precision attention of the first-stage model
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* __restrict__ s, int cols, float* __restrict__ p) {
  __shared__ float red[4];
  const long row = blockIdx.x;
  const float* sr = s + row * cols;
  float* pr = p + row * cols;
  float v[16];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = threadIdx.x + 256 * i;
    v[i] = c < cols ? sr[c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = expf(v[i] - mx);
    sum += v[i];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < cols) pr[c] = v[i] * inv;
  }
}

}  // namespace

int launch_softmax_rows_f32(const float* s, long rows, int cols, float* p, hipStream_t st) {
  if (cols > 4096 || rows > 0x7FFFFFFF) return mvd_fail("softmax_rows: cols <= 4096 expected");
  hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_softmax_rows(const float* s, long rows, int cols, half_t* p, hipStream_t st) {
  if (cols > 4096 || rows > 0x7FFFFFFF) return mvd_fail("softmax_rows: cols <= 4096 expected");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, s, cols, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_attention(const half_t* qk, int ldqk, const half_t* v, int ldv, half_t* out, int ldo, int B, int T, int heads,
                     int d, hipStream_t s, int Tstride) {
  if (Tstride <= 0) Tstride = T;
  if (Tstride < T || ldqk % 8 || ldv % 8 || ldo % 4 || d % 8 || ((uintptr_t)qk & 15) || ((uintptr_t)v & 15))
    return mvd_fail("attention: alignment (row strides and d multiples of 8, 16-byte aligned operands)");
  if ((long)cdiv(T, 128) * heads * B > 0x7FFFFFFFL) return mvd_fail("attention: grid too large");
  dim3 grid((unsigned)(cdiv(T, 128) * heads * B));
  static const int xcd_remap = getenv("MVD_ATTN_NO_XCD") == nullptr;  // A/B switch: the pre-remap workgroup order
  const float scale = 1.4426950408889634f / sqrtf((float)d);  // softmax scale * log2(e): the kernel uses exp2
#define MVD_ATTN(DD) \
  case DD: hipLaunchKernelGGL(attn_kernel<DD>, grid, dim3(256), 0, s, qk, ldqk, v, ldv, out, ldo, T, heads, scale, Tstride, xcd_remap); break;
  switch (d) {
    MVD_ATTN(8)
    MVD_ATTN(16)
    MVD_ATTN(32)
    MVD_ATTN(40)
    MVD_ATTN(64)
    MVD_ATTN(80)
    MVD_ATTN(160)
    default: return mvd_fail("attention: unsupported head dim");
  }
#undef MVD_ATTN
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
