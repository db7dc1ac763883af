// Mesh-conditioner gathers (all HBM/L2-bound scatter-gather, no GEMM shape):
//   vertex_gather   fused  project lattice -> bilinear 2-D sample -> trilinear gather at mesh vertices
//                   (morphable_diffusion.py:211-229): the 32^3 x 16N unprojection volume is never built
//   fuse_views      mean over views + 16x16 k=1 conv (network.py:41-72)
//   sparse_conv     submanifold / strided sparse 3x3x3 conv through a host-built neighbour table,
//                   eval BatchNorm + ReLU folded (network.py:74-161)
//   latent_gather   trilinear sample of the sparse CNN output at the 32^3 lattice (:232-257)
//   frustum_gather  trilinear sample of the fused volume along each target view's frustum (:302-315)
#include "common.h"

namespace {

// torch.linspace(a, b, n)[i] in fp32 (symmetric evaluation, as ATen does)
__device__ __forceinline__ float linspace_at(float a, float b, int n, int i) {
  const float step = (b - a) / (float)(n - 1);
  return i < n / 2 ? a + step * (float)i : b - step * (float)(n - 1 - i);
}

__global__ __launch_bounds__(256) void vertex_gather_kernel(const float* __restrict__ feats, const ViewCam* __restrict__ cams,
                                                            const int* __restrict__ view_idx, int n_views, const float* __restrict__ verts, int Nv, int V,
                                                            float vol_len, int S, int persp, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_views * Nv) return;
  const int view = idx / Nv, vi = idx - view * Nv;
  const ViewCam cam = cams[view_idx[view]];
  float pos[3], fr[3];
  int lo[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = verts[vi * 3 + a] / vol_len;
    pos[a] = (g + 1.0f) * 0.5f * (float)(V - 1);
    const float f = floorf(pos[a]);
    lo[a] = (int)f;
    fr[a] = pos[a] - f;
  }
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
  const float* fv = feats + (long)view * S * S * 16;
  for (int corner = 0; corner < 8; ++corner) {
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
    const int ix = lo[0] + bx, iy = lo[1] + by, iz = lo[2] + bz;
    if (ix < 0 || ix > V - 1 || iy < 0 || iy > V - 1 || iz < 0 || iz > V - 1) continue;
    const float w3 = (bx ? fr[0] : 1.f - fr[0]) * (by ? fr[1] : 1.f - fr[1]) * (bz ? fr[2] : 1.f - fr[2]);
    const float X = linspace_at(-vol_len, vol_len, V, ix), Y = linspace_at(-vol_len, vol_len, V, iy),
                Z = linspace_at(-vol_len, vol_len, V, iz);
    const float u = cam.P[0] * X + cam.P[1] * Y + cam.P[2] * Z + cam.P[3];
    const float v = cam.P[4] * X + cam.P[5] * Y + cam.P[6] * Z + cam.P[7];
    float px, py;
    if (persp) {
      float w = cam.P[8] * X + cam.P[9] * Y + cam.P[10] * Z + cam.P[11];
      w = w < 1e-4f ? 1e-4f : w;
      const float hs = (float)(S - 1) * 0.5f;
      px = ((u / w) / hs - 1.0f + 1.0f) * 0.5f * (float)(S - 1);
      py = ((v / w) / hs - 1.0f + 1.0f) * 0.5f * (float)(S - 1);
    } else {
      px = (u + 1.0f) * 0.5f * (float)(S - 1);
      py = (v + 1.0f) * 0.5f * (float)(S - 1);
    }
    const float fx0 = floorf(px), fy0 = floorf(py);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const float tx = px - fx0, ty = py - fy0;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int xx = x0 + (tap & 1), yy = y0 + (tap >> 1);
      if (xx < 0 || xx > S - 1 || yy < 0 || yy > S - 1) continue;
      const float w2 = ((tap & 1) ? tx : 1.f - tx) * ((tap >> 1) ? ty : 1.f - ty) * w3;
      const float4* f4 = (const float4*)(fv + ((long)yy * S + xx) * 16);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 f = f4[q];
        acc[q * 4 + 0] += w2 * f.x;
        acc[q * 4 + 1] += w2 * f.y;
        acc[q * 4 + 2] += w2 * f.z;
        acc[q * 4 + 3] += w2 * f.w;
      }
    }
  }
  float4* o4 = (float4*)(out + (long)idx * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) o4[q] = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
}

// out[v][co] (+)= (1/total_views) * sum_view sum_ci w[co][ci] * vf[view][v][ci]  (+ bias)
__global__ void fuse_views_kernel(const float* __restrict__ vf, int n_views, int Nv, int total_views,
                                  const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ out,
                                  int accumulate) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Nv * 16) return;
  const int v = idx >> 4, co = idx & 15;
  float acc = 0.f;
  for (int view = 0; view < n_views; ++view) {
    const float* f = vf + ((long)view * Nv + v) * 16;
#pragma unroll
    for (int ci = 0; ci < 16; ++ci) acc += w[co * 16 + ci] * f[ci];
  }
  acc /= (float)total_views;
  if (b) acc += b[co];
  out[idx] = accumulate ? out[idx] + acc : acc;
}

// in [n_in][Cin], nbr [n_out][27] (-1 = inactive), w [27][Cin][Cout]; out = relu(conv*scale + shift)
// One site per block: thread = (tap group p, output channel co); the 27 taps are dealt round-robin to the
// 256/Cout groups and combined through LDS, so a thread's dependent chain is <= 7 taps x Cin/8 vector steps
// instead of 27 x Cin scalar ones (the layer is latency-, not throughput-bound: <= 64 channels, ~10^3-10^4 sites).
__global__ __launch_bounds__(256) void sparse_conv_kernel(const float* __restrict__ in, const int* __restrict__ nbr,
                                                          int n_out, int Cin, int Cout, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* __restrict__ out) {
  __shared__ int s_nb[27];
  __shared__ float s_red[256];
  const int site = blockIdx.x, t = threadIdx.x;
  if (t < 27) s_nb[t] = nbr[(long)site * 27 + t];
  __syncthreads();
  const int co = t % Cout, p = t / Cout, P = 256 / Cout;
  float acc = 0.f;
  if (p < P) {
    for (int k = p; k < 27; k += P) {
      const int nb = s_nb[k];
      if (nb < 0) continue;
      const float* f = in + (long)nb * Cin;
      const float* wk = w + (long)k * Cin * Cout + co;
      if ((Cin & 7) == 0) {
        for (int c = 0; c < Cin; c += 8) {
          const float4 f0 = *(const float4*)(f + c), f1 = *(const float4*)(f + c + 4);
          float wv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) wv[e] = wk[(long)(c + e) * Cout];
          acc += f0.x * wv[0] + f0.y * wv[1] + f0.z * wv[2] + f0.w * wv[3];
          acc += f1.x * wv[4] + f1.y * wv[5] + f1.z * wv[6] + f1.w * wv[7];
        }
      } else {
        for (int c = 0; c < Cin; ++c) acc += f[c] * wk[(long)c * Cout];
      }
    }
  }
  s_red[t] = acc;
  __syncthreads();
  if (t < Cout) {
    float v = 0.f;
    for (int q = 0; q < P; ++q) v += s_red[q * Cout + t];
    out[(long)site * Cout + t] = scale ? fmaxf(v * scale[t] + shift[t], 0.f) : v;  // scale == null: raw conv output
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same layer as a tiled gather-GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: an exact fp32 fma chain, so the
// layer stays fp32 -- the conditioner's BatchNorm / ReLU masks depend on it).  A workgroup owns 16 output sites x all Cout;
// its four waves split the 27 taps (tap t -> wave t & 3) and combine through LDS in wave order (fixed summation order).
// Per tap a lane gathers ONE 16-byte piece of its neighbour row -- lane (m = lane & 15, q = lane >> 4) holds channels
// 4q .. 4q+3 of site m, which is the A operand of four consecutive MFMAs (k index of MFMA j = channel 16 cb + 4q + j) -- and the
// weights come pre-ordered as B fragments (sparse_w_frag_kernel: one coalesced 16-byte load per lane and fragment).  A tap none
// of the 16 sites has is skipped (wave-uniform); the operands of the next tap are in flight while the current one multiplies.
// Also the layer's data-gradient (dgrad): a sparse conv of d_out with the transposed weights through the same table read with
// the tap flipped (submanifold: nbr[s][k] = s' <=> nbr[s'][26-k] = s) or through the inverse table (strided), see k_cond_bwd.hip.
// NBW: 16-channel output blocks per workgroup (grid.y = Cout / 16 / NBW): the 64-channel layers have few sites (one or two
// thousand voxels after two strided layers), so their output blocks go to separate workgroups.
template <int CIN, int COUT, int NBW>
__global__ __launch_bounds__(256) void sparse_mfma_kernel(const float* __restrict__ in, const int* __restrict__ nbr, int n_out,
                                                          const float* __restrict__ wp, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int mask_nonrep, float* __restrict__ out) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CB = CIN / 16, NBT = COUT / 16, NB = NBW, CW = 16 * NBW;  // CW: output channels of this workgroup
  __shared__ int s_nb[16 * 27];
  __shared__ float s_red[4][16 * CW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s0 = blockIdx.x * 16, b0 = blockIdx.y * NBW;
  for (int i = tid; i < 16 * 27; i += 256) s_nb[i] = s0 + i / 27 < n_out ? nbr[(long)s0 * 27 + i] : -1;
  __syncthreads();
  const int m = lane & 15, q = lane >> 4;
  f32x4 acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // taps of this wave that at least one of the 16 sites has (bit i: tap wave + 4 i)
  unsigned act = 0;
  int nbs[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int t = wave + 4 * i;
    nbs[i] = t < 27 ? s_nb[m * 27 + t] : -1;
    if (__ballot(nbs[i] >= 0) != 0ull) act |= 1u << i;
  }
  f32x4 a0[CB], a1[CB], w0[CB * NB], w1[CB * NB];
  auto load = [&](int i, f32x4 (&a)[CB], f32x4 (&w)[CB * NB]) {
    int nb = -1;  // nbs[i] with a wave-uniform i: selected without dynamic register indexing
#pragma unroll
    for (int j = 0; j < 7; ++j) nb = i == j ? nbs[j] : nb;
    const int t = wave + 4 * i;
    const float* row = in + (long)(nb < 0 ? 0 : nb) * CIN + 4 * q;
    const float* wt = wp + (((long)t * CB * NBT + b0) * 64 + lane) * 4;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) {
      a[cb] = nb >= 0 ? *(const f32x4*)(row + cb * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < NB; ++b) w[cb * NB + b] = *(const f32x4*)(wt + (long)(cb * NBT + b) * 256);
    }
  };
  auto compute = [&](const f32x4 (&a)[CB], const f32x4 (&w)[CB * NB]) {
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb][j], w[cb * NB + b][j], acc[b], 0, 0, 0);
  };
  auto next = [&]() -> int {  // lowest remaining active tap slot, or -1
    if (!act) return -1;
    const int i = __builtin_ctz(act);
    act &= act - 1;
    return i;
  };
  int i0 = next(), i1 = -1;
  if (i0 >= 0) load(i0, a0, w0);
  while (i0 >= 0) {
    i1 = next();
    if (i1 >= 0) load(i1, a1, w1);
    compute(a0, w0);
    if (i1 < 0) break;
    i0 = next();
    if (i0 >= 0) load(i0, a0, w0);
    compute(a1, w1);
  }
  // D fragment: lane holds sites 4q .. 4q+3 (register r) x output channel 16 b + m
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) s_red[wave][(4 * q + r) * CW + b * 16 + m] = acc[b][r];
  __syncthreads();
  for (int e = tid; e < 16 * CW; e += 256) {
    const int sl = e / CW, co = b0 * 16 + e - sl * CW, site = s0 + sl;
    if (site >= n_out) break;
    float v = ((s_red[0][e] + s_red[1][e]) + s_red[2][e]) + s_red[3][e];
    if (scale) v = fmaxf(v * scale[co] + shift[co], 0.f);
    if (mask_nonrep && s_nb[sl * 27 + 13] != site) v = 0.f;  // a duplicate vertex's row: nothing reads it
    out[(long)site * COUT + co] = v;
  }
#endif
}

// w_src -> B fragments of sparse_mfma_kernel: dst[t][cb][b][lane][j] = W(t, in = 16 cb + 4 (lane >> 4) + j, out = 16 b + (lane & 15)).
// transposed == 0: W(t, in, out) = w[t][in][out] of the forward pack [27][Cin][Cout] (Cl_in = Cin, Cl_out = Cout).
// transposed == 1: the data-gradient's layer (its inputs are the forward outputs): W(t, in, out) = w[flip ? 26 - t : t][out][in].
__global__ void sparse_w_frag_kernel(const float* __restrict__ w, int Cl_in, int Cl_out, int transposed, int flip, float* __restrict__ dst) {
  const long total = (long)27 * Cl_in * Cl_out;
  const int CB = Cl_in / 16, NB = Cl_out / 16;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long f = i >> 8;
    const int b = (int)(f % NB);
    f /= NB;
    const int cb = (int)(f % CB), t = (int)(f / CB);
    const int ci = cb * 16 + 4 * (lane >> 4) + j, co = b * 16 + (lane & 15);
    dst[i] = transposed ? w[((long)(flip ? 26 - t : t) * Cl_out + co) * Cl_in + ci] : w[((long)t * Cl_in + ci) * Cl_out + co];
  }
}

__global__ void sparse_w_pack_kernel(const float* __restrict__ src, int Cin, int Cout, int layout, float* __restrict__ dst) {
  const long total = (long)27 * Cin * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), k = (int)(i / ((long)Cout * Cin));
    const long sidx = layout == 0 ? ((long)co * Cin + ci) * 27 + k : layout == 1 ? ((long)co * 27 + k) * Cin + ci : i;
    dst[i] = src[sidx];
  }
}
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - mean[c] * sc;
}

// BatchNorm1d on batch statistics (biased variance) + ReLU over the n active rows of a sparse layer: one workgroup per channel --
// mean, then centred second moment over the n rows (fp32, fixed order), then the apply (y may be x).  stats_out (optional):
// [mean | rstd] for the backward pass.  (A coalesced two-launch form with a shifted one-pass variance was built and is 2x
// faster, 23 -> 12 us, but its statistics differ from these in the last bits, which re-draws near-tie ReLU masks downstream: the
// gradients of the 4x4 DepthTransformer moved from 3.7e-2 to 1.0e-1 of the reference's.  The backward pass, which draws no
// masks, uses the coalesced form: k_cond_bwd.hip.)
__global__ __launch_bounds__(256) void bn_rows_relu_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float* __restrict__ stats_out, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float momentum) {
  __shared__ float s_red[256];
  const int c = blockIdx.x, t = threadIdx.x;
  float a = 0.f;
#pragma unroll 8
  for (int r = t; r < n; r += 256) a += x[(long)r * C + c];
  s_red[t] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  const float mean = s_red[0] / (float)n;
  __syncthreads();
  float q = 0.f;
#pragma unroll 8
  for (int r = t; r < n; r += 256) {
    const float d = x[(long)r * C + c] - mean;
    q += d * d;
  }
  s_red[t] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  const float rstd = rsqrtf(s_red[0] / (float)n + eps), g = gamma[c], b = beta[c];
  if (stats_out && t == 0) {
    stats_out[c] = mean;
    stats_out[C + c] = rstd;
  }
  if (rmean && t == 0) {  // nn.BatchNorm1d in train mode: running statistics, the variance as the unbiased estimate
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (s_red[0] / (float)(n > 1 ? n - 1 : 1));
  }
#pragma unroll 8
  for (int r = t; r < n; r += 256) y[(long)r * C + c] = fmaxf((x[(long)r * C + c] - mean) * rstd * g + b, 0.f);
}

// The same kernel for layers of at most 256 * BN_MAXR rows (every level of a FLAME-sized mesh): a thread's rows are loaded ONCE,
// all loads in flight together, and stay in registers for the mean, the centred second moment and the apply -- the looped form
// above reads the column three times in dependent batches of eight strided loads (20 us per launch, 144 launches per training
// step at 8 samples).  Same partial sums in the same order, same tree: bit-identical statistics and outputs (the masks drawn from
// them do not move).
// (two sizes: 24 rows per thread, and 48 -- a sparse point set DILATES under the strided conv, level 1 of a 4.6 k-voxel mesh has
// 9.4 k sites)
template <int BN_MAXR>
__global__ __launch_bounds__(256) void bn_rows_relu_reg_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                              float* __restrict__ stats_out, float* __restrict__ rmean,
                                                              float* __restrict__ rvar, float momentum) {
  __shared__ float s_red[256];
  const int c = blockIdx.x, t = threadIdx.x;
  float v[BN_MAXR];
#pragma unroll
  for (int i = 0; i < BN_MAXR; ++i) {
    const int r = t + 256 * i;
    v[i] = r < n ? x[(long)r * C + c] : 0.f;
  }
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < BN_MAXR; ++i)
    if (t + 256 * i < n) a += v[i];
  s_red[t] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  const float mean = s_red[0] / (float)n;
  __syncthreads();
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < BN_MAXR; ++i)
    if (t + 256 * i < n) {
      const float d = v[i] - mean;
      q += d * d;
    }
  s_red[t] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  const float rstd = rsqrtf(s_red[0] / (float)n + eps), g = gamma[c], b = beta[c];
  if (stats_out && t == 0) {
    stats_out[c] = mean;
    stats_out[C + c] = rstd;
  }
  if (rmean && t == 0) {
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (s_red[0] / (float)(n > 1 ? n - 1 : 1));
  }
#pragma unroll
  for (int i = 0; i < BN_MAXR; ++i) {
    const int r = t + 256 * i;
    if (r < n) y[(long)r * C + c] = fmaxf((v[i] - mean) * rstd * g + b, 0.f);
  }
}

// mean((a - b)^2) of n elements into out[0]: one workgroup, fixed summation order
__global__ __launch_bounds__(1024) void mse_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                   float* __restrict__ out) {
  __shared__ double s_red[1024];
  const int t = threadIdx.x;
  double acc = 0.0;
  for (size_t i = t; i < n; i += 1024) {
    const double d = (double)a[i] - (double)b[i];
    acc += d * d;
  }
  s_red[t] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) s_red[t] += s_red[t + o];
    __syncthreads();
  }
  if (t == 0) out[0] = (float)(s_red[0] / (double)n);
}

// out [V][V][V][C] (z,y,x,c) fp32
__global__ void latent_gather_kernel(const float* __restrict__ feats, const int* __restrict__ grid, int gd, int gh, int gw,
                                     float minx, float miny, float minz, float shx, float shy, float shz, float voxel, int V,
                                     float vol_len, int C, float* __restrict__ out) {
  const long total = (long)V * V * V * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    const int pt = (int)(idx / C);
    const int ix = pt % V, iy = (pt / V) % V, iz = pt / (V * V);
    const float X = linspace_at(-vol_len, vol_len, V, ix), Y = linspace_at(-vol_len, vol_len, V, iy),
                Z = linspace_at(-vol_len, vol_len, V, iz);
    const float gx = (X - minx) / voxel / shx * 2.f - 1.f;
    const float gy = (Y - miny) / voxel / shy * 2.f - 1.f;
    const float gz = (Z - minz) / voxel / shz * 2.f - 1.f;
    const float px = (gx + 1.f) * 0.5f * (float)(gw - 1), py = (gy + 1.f) * 0.5f * (float)(gh - 1),
                pz = (gz + 1.f) * 0.5f * (float)(gd - 1);
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const float tx = px - fx, ty = py - fy, tz = pz - fz;
    float acc = 0.f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
      const int xx = x0 + bx, yy = y0 + by, zz = z0 + bz;
      if (xx < 0 || xx > gw - 1 || yy < 0 || yy > gh - 1 || zz < 0 || zz > gd - 1) continue;
      const int row = grid[((long)zz * gh + yy) * gw + xx];
      if (row < 0) continue;
      const float wgt = (bx ? tx : 1.f - tx) * (by ? ty : 1.f - ty) * (bz ? tz : 1.f - tz);
      acc += wgt * feats[(long)row * C + c];
    }
    out[idx] = acc;
  }
}

// rows [n][C] of the active sites -> dense [C][gd][gh][gw] (spconv's .dense(): zero where no site is active)
__global__ void sparse_densify_kernel(const float* __restrict__ feats, const int* __restrict__ grid, long nvox, int C,
                                      float* __restrict__ out) {
  const long total = nvox * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long vox = idx % nvox;
    const int c = (int)(idx / nvox);
    const int row = grid[vox];
    out[idx] = row < 0 ? 0.f : feats[(long)row * C + c];
  }
}

// vol [V][V][V][C] fp32 -> out [TN][D][S][S][C] fp16 ; 16 threads per point, C/16 channels each (C = 64 -> 4)
__global__ __launch_bounds__(256) void frustum_gather_kernel(const float* __restrict__ vol, const ViewCam* __restrict__ cams,
                                                             const int* __restrict__ view_idx, int TN, int D, int S, int V,
                                                             float vol_len, int persp, half_t* __restrict__ out) {
  constexpr int C = 64;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long pt = gid >> 4;
  const int cq = (int)(gid & 15) * 4;
  const long npts = (long)TN * D * S * S;
  if (pt >= npts) return;
  const int x = (int)(pt % S), y = (int)((pt / S) % S), d = (int)((pt / ((long)S * S)) % D), tv = (int)(pt / ((long)S * S * D));
  const ViewCam cam = cams[view_idx[tv]];
  const float depth = linspace_at(0.f, 1.f, D, d) * (cam.far_ - cam.near_) + cam.near_;
  float wx, wy, wz;
  if (persp) {
    const float a = (float)x * depth, b = (float)y * depth, c = depth;
    wx = cam.Pinv[0] * a + cam.Pinv[1] * b + cam.Pinv[2] * c + cam.Pinv[3];
    wy = cam.Pinv[4] * a + cam.Pinv[5] * b + cam.Pinv[6] * c + cam.Pinv[7];
    wz = cam.Pinv[8] * a + cam.Pinv[9] * b + cam.Pinv[10] * c + cam.Pinv[11];
  } else {
    const float gx = 2.f * (float)x / (float)(S - 1) - 1.f, gy = 2.f * (float)y / (float)(S - 1) - 1.f;
    const float a = cam.Kinv[0] * gx + cam.Kinv[1] * gy + cam.Kinv[2];
    const float b = cam.Kinv[3] * gx + cam.Kinv[4] * gy + cam.Kinv[5];
    const float c = depth;
    wx = cam.Pinv[0] * a + cam.Pinv[1] * b + cam.Pinv[2] * c + cam.Pinv[3];
    wy = cam.Pinv[4] * a + cam.Pinv[5] * b + cam.Pinv[6] * c + cam.Pinv[7];
    wz = cam.Pinv[8] * a + cam.Pinv[9] * b + cam.Pinv[10] * c + cam.Pinv[11];
  }
  const float px = (wx / vol_len + 1.f) * 0.5f * (float)(V - 1), py = (wy / vol_len + 1.f) * 0.5f * (float)(V - 1),
              pz = (wz / vol_len + 1.f) * 0.5f * (float)(V - 1);
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const float tx = px - fx, ty = py - fy, tz = pz - fz;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
    const int xx = x0 + bx, yy = y0 + by, zz = z0 + bz;
    if (xx < 0 || xx > V - 1 || yy < 0 || yy > V - 1 || zz < 0 || zz > V - 1) continue;
    const float wgt = (bx ? tx : 1.f - tx) * (by ? ty : 1.f - ty) * (bz ? tz : 1.f - tz);
    const float4 f = *(const float4*)(vol + (((long)zz * V + yy) * V + xx) * C + cq);
    acc.x += wgt * f.x;
    acc.y += wgt * f.y;
    acc.z += wgt * f.z;
    acc.w += wgt * f.w;
  }
  h4 o;
  o[0] = (half_t)acc.x; o[1] = (half_t)acc.y; o[2] = (half_t)acc.z; o[3] = (half_t)acc.w;
  *(h4*)(out + pt * C + cq) = o;
}

}  // namespace

int launch_vertex_gather(const float* feats, const ViewCam* cams, const int* view_idx, int n_views, const float* verts, int Nv, int V,
                         float vol_len, int fsize, int persp, float* out, hipStream_t s) {
  hipLaunchKernelGGL(vertex_gather_kernel, dim3(cdiv(n_views * Nv, 256)), dim3(256), 0, s, feats, cams, view_idx, n_views, verts,
                     Nv, V, vol_len, fsize, persp, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_fuse_views(const float* vf, int n_views, int Nv, int total_views, const float* w, const float* b, float* out,
                      int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(fuse_views_kernel, dim3(cdiv(Nv * 16, 256)), dim3(256), 0, s, vf, n_views, Nv, total_views, w, b,
                     out, accumulate);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// wp: the layer's B-fragment pack (launch_sparse_w_frag) or null; channel counts the matrix-core kernel does not take (not
// multiples of 16, more than 64) go through the one-site-per-workgroup kernel on `w` ([27][Cin][Cout]).
int launch_sparse_conv(const float* in, const int* nbr, int n_out, int Cin, int Cout, const float* w, const float* wp, const float* scale,
                       const float* shift, float* out, hipStream_t s, int mask_nonrep) {
  if (Cout > 256 || n_out <= 0) return n_out <= 0 ? 0 : mvd_fail("sparse_conv: Cout > 256");
  const dim3 blk(256);
#define MVD_SP(CI, CO, NBW)                                                                                                   \
  if (wp && Cin == CI && Cout == CO) {                                                                                        \
    hipLaunchKernelGGL((sparse_mfma_kernel<CI, CO, NBW>), dim3(cdiv(n_out, 16), CO / 16 / NBW), blk, 0, s, in, nbr, n_out, wp, scale, \
                       shift, mask_nonrep, out);                                                                              \
    HIP_CHECK_RET(hipGetLastError());                                                                                        \
    return 0;                                                                                                                \
  }
  MVD_SP(16, 16, 1) MVD_SP(16, 32, 2) MVD_SP(32, 16, 1) MVD_SP(32, 32, 2) MVD_SP(32, 64, 1) MVD_SP(64, 32, 1) MVD_SP(64, 64, 1)
#undef MVD_SP
  if (mask_nonrep || !w) return mvd_fail("sparse_conv: no matrix-core kernel for these channel counts");
  hipLaunchKernelGGL(sparse_conv_kernel, dim3(n_out), dim3(256), 0, s, in, nbr, n_out, Cin, Cout, w,
                     scale, shift, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
bool sparse_mfma_takes(int Cin, int Cout) {
  return (Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 16 || Cout == 32 || Cout == 64) && !(Cin == 16 && Cout == 64) &&
         !(Cin == 64 && Cout == 16);
}
int launch_sparse_w_frag(const float* w, int Cl_in, int Cl_out, int transposed, int flip, float* dst, hipStream_t s) {
  const long total = (long)27 * Cl_in * Cl_out;
  hipLaunchKernelGGL(sparse_w_frag_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, w, Cl_in, Cl_out, transposed, flip, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// sparse conv weight in one of the checkpoint layouts (engine_weights.hip: build_sparse_layer) -> [27][Cin][Cout]
int launch_sparse_w_pack(const float* src, int Cin, int Cout, int layout, float* dst, hipStream_t s) {
  const long total = (long)27 * Cin * Cout;
  hipLaunchKernelGGL(sparse_w_pack_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, src, Cin, Cout, layout, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// eval-mode BatchNorm1d folded into the conv epilogue: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
int launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int C, float* scale,
                   float* shift, hipStream_t s) {
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, gamma, beta, mean, var, eps, C, scale, shift);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// y (may be x) = relu(batchnorm(x)); stats_out: optional [2][C] (mean | rstd)
// rmean / rvar (optional): running statistics, updated with `momentum` as nn.BatchNorm1d does in train mode
int launch_bn_rows_relu(const float* x, float* y, int n, int C, const float* gamma, const float* beta, float eps, float* stats_out,
                        hipStream_t s, float* rmean, float* rvar, float momentum) {
  if (n <= 0 || C <= 0) return 0;
  const char* loop_env = getenv("MVD_BN_LOOP");  // A/B + the bit-identity test: the looped form
  const bool loop = loop_env && loop_env[0] == '1';
  if (n <= 256 * 24 && !loop)
    hipLaunchKernelGGL(bn_rows_relu_reg_kernel<24>, dim3(C), dim3(256), 0, s, x, y, n, C, gamma, beta, eps, stats_out, rmean, rvar, momentum);
  else if (n <= 256 * 48 && !loop)
    hipLaunchKernelGGL(bn_rows_relu_reg_kernel<48>, dim3(C), dim3(256), 0, s, x, y, n, C, gamma, beta, eps, stats_out, rmean, rvar, momentum);
  else
    hipLaunchKernelGGL(bn_rows_relu_kernel, dim3(C), dim3(256), 0, s, x, y, n, C, gamma, beta, eps, stats_out, rmean, rvar, momentum);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_mse(const float* a, const float* b, size_t n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(mse_kernel, dim3(1), dim3(1024), 0, s, a, b, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_sparse_densify(const float* feats, const int* grid, long nvox, int C, float* out, hipStream_t s) {
  const long total = nvox * C;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(sparse_densify_kernel, dim3(blocks), dim3(256), 0, s, feats, grid, nvox, C, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int launch_latent_gather(const float* feats, const int* grid, int gd, int gh, int gw, const float* min_xyz,
                         const int* out_sh, float voxel, int V, float vol_len, float* out, hipStream_t s) {
  // min_xyz / out_sh are HOST pointers (step-invariant mesh metadata); out_sh is (d,h,w) = (z,y,x)
  const int C = 64;
  size_t total = (size_t)V * V * V * C;
  int blocks = (int)((total + 255) / 256);
  hipLaunchKernelGGL(latent_gather_kernel, dim3(blocks), dim3(256), 0, s, feats, grid, gd, gh, gw, min_xyz[0], min_xyz[1],
                     min_xyz[2], (float)out_sh[2], (float)out_sh[1], (float)out_sh[0], voxel, V, vol_len, C, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int launch_frustum_gather(const float* vol, const ViewCam* cams, const int* view_idx, int TN, int D, int S, int V,
                          float vol_len, int persp, half_t* out, hipStream_t s) {
  const long threads = (long)TN * D * S * S * 16;
  hipLaunchKernelGGL(frustum_gather_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, vol, cams, view_idx, TN,
                     D, S, V, vol_len, persp, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
