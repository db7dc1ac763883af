// Backward kernels of the mesh conditioner (SpatialVolumeNet, morphable_diffusion.py:151-320; networks network.py): the
// adjoints of the gathers of k_cond.hip (same index arithmetic, scatter instead of gather), the sparse voxel CNN's layers in
// train mode (conv dgrad / wgrad through the neighbour tables, BatchNorm1d with batch statistics + ReLU), the view fusion, and
// the transposing im2col for the frustum network's 3-D convolutions.
// The three scatters use hardware fp32 atomic adds (unsafeAtomicAdd: the returning CAS loop of plain atomicAdd is 30x slower here) (unordered: the conditioner's gradients are reproducible to rounding, not bit
// for bit -- the UNet's are); everything else has fixed summation orders.
#include "common.h"

namespace {

inline int gridn(size_t n, int cap = 8192) {
  size_t g = (n + 255) / 256;
  return (int)(g > (size_t)cap ? (size_t)cap : (g < 1 ? 1 : g));
}

__device__ __forceinline__ float linspace_at(float a, float b, int n, int i) {
  const float step = (b - a) / (float)(n - 1);
  return i < n / 2 ? a + step * (float)i : b - step * (float)(n - 1 - i);
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const half_t* p) { return (float)*p; }

// adjoint of frustum_gather_kernel: d_vol[corner][c] += w * d_out[point][c]   (d_out fp32 [TN*D*S*S][64])
__global__ __launch_bounds__(256) void frustum_scatter_kernel(const float* __restrict__ d_out, const ViewCam* __restrict__ cams,
                                                              const int* __restrict__ view_idx, int TN, int D, int S, int V, float vol_len,
                                                              int persp, float* __restrict__ d_vol) {
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
  const float4 g = *(const float4*)(d_out + pt * C + cq);
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
    const int xx = x0 + bx, yy = y0 + by, zz = z0 + bz;
    if (xx < 0 || xx > V - 1 || yy < 0 || yy > V - 1 || zz < 0 || zz > V - 1) continue;
    const float wgt = (bx ? tx : 1.f - tx) * (by ? ty : 1.f - ty) * (bz ? tz : 1.f - tz);
    float* o = d_vol + (((long)zz * V + yy) * V + xx) * C + cq;
    unsafeAtomicAdd(o + 0, wgt * g.x);
    unsafeAtomicAdd(o + 1, wgt * g.y);
    unsafeAtomicAdd(o + 2, wgt * g.z);
    unsafeAtomicAdd(o + 3, wgt * g.w);
  }
}

// adjoint of latent_gather_kernel: d_rows[row][c] += w * d_vol[point][c]
__global__ void latent_scatter_kernel(const float* __restrict__ d_vol, const int* __restrict__ grid, int gd, int gh, int gw, float minx,
                                      float miny, float minz, float shx, float shy, float shz, float voxel, int V, float vol_len, int C,
                                      float* __restrict__ d_rows) {
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
    const float g = d_vol[idx];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const int bx = corner & 1, by = (corner >> 1) & 1, bz = corner >> 2;
      const int xx = x0 + bx, yy = y0 + by, zz = z0 + bz;
      if (xx < 0 || xx > gw - 1 || yy < 0 || yy > gh - 1 || zz < 0 || zz > gd - 1) continue;
      const int row = grid[((long)zz * gh + yy) * gw + xx];
      if (row < 0) continue;
      const float wgt = (bx ? tx : 1.f - tx) * (by ? ty : 1.f - ty) * (bz ? tz : 1.f - tz);
      unsafeAtomicAdd(d_rows + (long)row * C + c, wgt * g);
    }
  }
}

// adjoint of vertex_gather_kernel: d_feats[view][pixel][c] += w2 * d_out[view][vertex][c].  One workgroup per view accumulates the
// whole S x S x 16 image in LDS (ds_add_f32: thousands of vertices land on the same few hundred pixels) and writes it once.
// gridDim.y workgroups share a view (each takes a slice of the vertices, its own LDS image, and adds it to HBM with fp32 atomics).
__global__ __launch_bounds__(1024) void vertex_scatter_kernel(const float* __restrict__ d_out, const ViewCam* __restrict__ cams,
                                                              const int* __restrict__ view_idx, const float* __restrict__ verts, int Nv, int V,
                                                              float vol_len, int S, int persp, float* __restrict__ d_feats) {
  extern __shared__ float s_img[];  // [16][S*S] channel-major: the lanes of a wave hit different pixels, i.e. different banks
  const int SS = S * S;
  const int view = blockIdx.x;
  const ViewCam cam = cams[view_idx[view]];
  for (int i = threadIdx.x; i < S * S * 16; i += blockDim.x) s_img[i] = 0.f;
  __syncthreads();
  // thread = (vertex, corner): 8 threads share a vertex
  const int per = (Nv + gridDim.y - 1) / gridDim.y, v_beg = blockIdx.y * per, v_end = min(Nv, v_beg + per);
  for (int job = v_beg * 8 + threadIdx.x; job < v_end * 8; job += blockDim.x) {
    const int vi = job >> 3, corner = job & 7;
    float fr[3];
    int lo[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float g = verts[vi * 3 + a] / vol_len;
      const float pos = (g + 1.0f) * 0.5f * (float)(V - 1);
      const float f = floorf(pos);
      lo[a] = (int)f;
      fr[a] = pos - f;
    }
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
    const float* gsrc = d_out + ((long)view * Nv + vi) * 16;
    float g16[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) g16[c] = gsrc[c];
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
      const int xx = x0 + (tap & 1), yy = y0 + (tap >> 1);
      if (xx < 0 || xx > S - 1 || yy < 0 || yy > S - 1) continue;
      const float w2 = ((tap & 1) ? tx : 1.f - tx) * ((tap >> 1) ? ty : 1.f - ty) * w3;
      float* o = s_img + yy * S + xx;
#pragma unroll
      for (int c = 0; c < 16; ++c) unsafeAtomicAdd(o + c * SS, w2 * g16[c]);
    }
  }
  __syncthreads();
  float* fv = d_feats + (long)view * S * S * 16;
  for (int i = threadIdx.x; i < S * S * 16; i += blockDim.x) {
    const float v = s_img[(i & 15) * SS + (i >> 4)];
    if (v != 0.f) unsafeAtomicAdd(fv + i, v);
  }
}

// ---- view fusion (SMPLFeatureExtractor, network.py:41-72): fused[v][co] = sum_ci w[co][ci] mean_view vf[view][v][ci] + b[co] ----
// d_vf[view][v][ci] = (1 / total_views) sum_co w[co][ci] d_fused[v][co]  (the same for every view)
__global__ void fuse_bwd_dvf_kernel(const float* __restrict__ d_fused, const float* __restrict__ w, int n_views, int Nv, int total_views,
                                    float* __restrict__ d_vf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Nv * 16) return;
  const int v = idx >> 4, ci = idx & 15;
  float acc = 0.f;
#pragma unroll
  for (int co = 0; co < 16; ++co) acc += w[co * 16 + ci] * d_fused[(long)v * 16 + co];
  acc /= (float)total_views;
  for (int view = 0; view < n_views; ++view) d_vf[((long)view * Nv + v) * 16 + ci] = acc;
}
// dw[co][ci] += sum_v d_fused[v][co] mean_view vf[.][v][ci];  db[co] += sum_v d_fused[v][co].
// A workgroup takes FW_CH vertices: their view mean m[v][16] and gradient rows g[v][16] go to LDS with row-contiguous 16-byte
// loads (the one-workgroup-per-(co, ci) form walked the vertices with a 64-byte stride, 16 views each: 148 us per sample), then
// thread (co, ci) sums its 256-vertex partial from LDS; partials per chunk, added in chunk order by fuse_bwd_w_reduce_kernel.
constexpr int FW_CH = 256;
__global__ __launch_bounds__(256) void fuse_bwd_w_kernel(const float* __restrict__ d_fused, const float* __restrict__ vf, int n_views, int Nv,
                                                         int total_views, float* __restrict__ part) {
  __shared__ float s_m[FW_CH][17], s_g[FW_CH][17];  // (+1: the (co, ci) readers of a row do not share a bank)
  const int t = threadIdx.x, v0 = blockIdx.x * FW_CH;
  for (int e = t; e < FW_CH * 4; e += 256) {
    const int vl = e >> 2, q = e & 3, v = v0 + vl;
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f), g = m;
    if (v < Nv) {
      for (int view = 0; view < n_views; ++view) {
        const float4 x = *(const float4*)(vf + ((long)view * Nv + v) * 16 + 4 * q);
        m.x += x.x; m.y += x.y; m.z += x.z; m.w += x.w;
      }
      g = *(const float4*)(d_fused + (long)v * 16 + 4 * q);
    }
    const float inv = 1.0f / (float)total_views;
    s_m[vl][4 * q] = m.x * inv; s_m[vl][4 * q + 1] = m.y * inv; s_m[vl][4 * q + 2] = m.z * inv; s_m[vl][4 * q + 3] = m.w * inv;
    s_g[vl][4 * q] = g.x; s_g[vl][4 * q + 1] = g.y; s_g[vl][4 * q + 2] = g.z; s_g[vl][4 * q + 3] = g.w;
  }
  __syncthreads();
  const int co = t >> 4, ci = t & 15;
  float a = 0.f, b = 0.f;
#pragma unroll 8
  for (int vl = 0; vl < FW_CH; ++vl) {
    const float g = s_g[vl][co];
    a += g * s_m[vl][ci];
    b += g;
  }
  part[(long)blockIdx.x * 272 + t] = a;
  if (ci == 0) part[(long)blockIdx.x * 272 + 256 + co] = b;
}
__global__ __launch_bounds__(256) void fuse_bwd_w_reduce_kernel(const float* __restrict__ part, int nchunk, float* __restrict__ dw,
                                                                float* __restrict__ db) {
  const int t = threadIdx.x;
  float a = 0.f, b = 0.f;
  for (int c = 0; c < nchunk; ++c) {
    a += part[(long)c * 272 + t];
    if (t < 16) b += part[(long)c * 272 + 256 + t];
  }
  dw[t] += a;
  if (t < 16) db[t] += b;
}

// ---- sparse voxel CNN, train mode -------------------------------------------------------------------------------------
// BatchNorm1d (batch statistics over the n active rows, biased variance) + ReLU backward, rows [n][C] with C dividing 256.
// xraw = the conv output, stats = [mean | rstd] of the forward pass, dy = gradient w.r.t. relu(bn(xraw)); writes d xraw over dy
// (in place), dgamma / dbeta accumulated.  Two launches over chunks of 128 rows, every access a full row segment (the
// one-workgroup-per-channel form walked columns with a stride of C floats: 44 -> 12 us): per-chunk sums of du and du * xhat, then every workgroup adds the partials in chunk order
// and applies; workgroup 0 accumulates the parameter gradients.
constexpr int BNB_CH = 128;
__global__ __launch_bounds__(256) void bn_rows_bwd_sums_kernel(const float* __restrict__ xraw, const float* __restrict__ dy, int n, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ stats, float* __restrict__ part) {
  __shared__ float s_a[256], s_b[256];
  const int RL = 256 / C, t = threadIdx.x, c = t % C, rl = t / C;
  const int r0 = blockIdx.x * BNB_CH, r1 = min(n, r0 + BNB_CH);
  float s1 = 0.f, s2 = 0.f;
  if (rl < RL) {
    const float mean = stats[c], rstd = stats[C + c], g = gamma[c], b = beta[c];
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += RL) {
      const float xh = (xraw[(long)r * C + c] - mean) * rstd;
      const float du = (g * xh + b) > 0.f ? dy[(long)r * C + c] : 0.f;
      s1 += du;
      s2 += du * xh;
    }
  }
  s_a[t] = s1;
  s_b[t] = s2;
  __syncthreads();
  if (rl == 0) {
    for (int i = 1; i < RL; ++i) {
      s1 += s_a[i * C + c];
      s2 += s_b[i * C + c];
    }
    part[((long)blockIdx.x * 2) * C + c] = s1;
    part[((long)blockIdx.x * 2 + 1) * C + c] = s2;
  }
}
__global__ __launch_bounds__(256) void bn_rows_bwd_apply_kernel(const float* __restrict__ xraw, float* __restrict__ dy, int n, int C,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ stats, const float* __restrict__ part, int nchunk,
                                                                float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_1[256], s_2[256];
  const int t = threadIdx.x;
  if (t < C) {
    float a = 0.f, q = 0.f;
    for (int k = 0; k < nchunk; ++k) {
      a += part[((long)k * 2) * C + t];
      q += part[((long)k * 2 + 1) * C + t];
    }
    if (blockIdx.x == 0) {
      dgamma[t] += q;
      dbeta[t] += a;
    }
    s_1[t] = a / (float)n;
    s_2[t] = q / (float)n;
  }
  __syncthreads();
  const long e0 = (long)blockIdx.x * BNB_CH * C, e1 = min((long)n * C, e0 + (long)BNB_CH * C);
#pragma unroll 8
  for (long e = e0 + t; e < e1; e += 256) {
    const int c = (int)(e % C);
    const float rstd = stats[C + c], g = gamma[c];
    const float xh = (xraw[e] - stats[c]) * rstd;
    const float du = (g * xh + beta[c]) > 0.f ? dy[e] : 0.f;
    dy[e] = g * rstd * (du - s_1[c] - xh * s_2[c]);
  }
}

// d_in[nbr[site][k]][ci] += sum_co w[k][ci][co] d_out[site][co]: one workgroup per output site
__global__ __launch_bounds__(256) void sparse_conv_dgrad_kernel(const float* __restrict__ d_out, const int* __restrict__ nbr, int Cin,
                                                                int Cout, const float* __restrict__ w, float* __restrict__ d_in) {
  __shared__ int s_nb[27];
  __shared__ float s_g[256];
  const int site = blockIdx.x, t = threadIdx.x;
  if (t < 27) s_nb[t] = nbr[(long)site * 27 + t];
  if (t < Cout) s_g[t] = d_out[(long)site * Cout + t];
  __syncthreads();
  const int ci = t % Cin, p = t / Cin, P = 256 / Cin;
  if (p >= P) return;
  for (int k = p; k < 27; k += P) {
    const int nb = s_nb[k];
    if (nb < 0) continue;
    const float* wk = w + ((long)k * Cin + ci) * Cout;
    float acc = 0.f;
    for (int co = 0; co < Cout; ++co) acc += wk[co] * s_g[co];
    unsafeAtomicAdd(d_in + (long)nb * Cin + ci, acc);
  }
}

// dw[k][ci][co] = sum_site in[nbr[site][k]][ci] d_out[site][co]: grid (27, ceil(Cin*Cout / 256), site chunks); every chunk of
// SP_CHUNK sites writes its partial product to part[chunk][27][Cin][Cout], summed in chunk order by sparse_wgrad_reduce_kernel.
// Only ~30 % of a surface mesh's (site, tap) pairs are active: each wave first compacts the active pairs of its quarter of the
// chunk into LDS (ballot + prefix popcount: site order is kept, so the sum order is fixed), then every thread walks the four
// lists -- a third of the iterations, none of them a skipped one, four independent loads in flight.
constexpr int SP_CHUNK = 256;
__global__ __launch_bounds__(256) void sparse_conv_wgrad_kernel(const float* __restrict__ in, const int* __restrict__ nbr,
                                                                const float* __restrict__ d_out, int n_out, int Cin, int Cout,
                                                                float* __restrict__ part) {
  __shared__ int s_nb[4][SP_CHUNK / 4], s_site[4][SP_CHUNK / 4], s_cnt[4];
  const int k = blockIdx.x, e = blockIdx.y * 256 + threadIdx.x, chunk = blockIdx.z;
  const int s0 = chunk * SP_CHUNK, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < SP_CHUNK / 256; ++j) {
    const int site = s0 + wave * (SP_CHUNK / 4) + j * 64 + lane;
    const int nb = site < n_out ? nbr[(long)site * 27 + k] : -1;
    const unsigned long long m = __ballot(nb >= 0);
    if (nb >= 0) {
      const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
      s_nb[wave][pos] = nb;
      s_site[wave][pos] = site;
    }
    cnt += __popcll(m);
  }
  if (lane == 0) s_cnt[wave] = cnt;
  __syncthreads();
  if (e >= Cin * Cout) return;
  const int ci = e / Cout, co = e - ci * Cout;
  const float* pin = in + ci;
  const float* pd = d_out + co;
  float acc = 0.f;
  for (int w = 0; w < 4; ++w) {
    const int n = s_cnt[w];
    int i = 0;
    for (; i + 4 <= n; i += 4) {
      const float a0 = pin[(long)s_nb[w][i] * Cin], a1 = pin[(long)s_nb[w][i + 1] * Cin], a2 = pin[(long)s_nb[w][i + 2] * Cin],
                  a3 = pin[(long)s_nb[w][i + 3] * Cin];
      const float b0 = pd[(long)s_site[w][i] * Cout], b1 = pd[(long)s_site[w][i + 1] * Cout], b2 = pd[(long)s_site[w][i + 2] * Cout],
                  b3 = pd[(long)s_site[w][i + 3] * Cout];
      acc += a0 * b0;
      acc += a1 * b1;
      acc += a2 * b2;
      acc += a3 * b3;
    }
    for (; i < n; ++i) acc += pin[(long)s_nb[w][i] * Cin] * pd[(long)s_site[w][i] * Cout];
  }
  part[(((long)chunk * 27 + k) * Cin + ci) * Cout + co] = acc;
}
// The same weight gradient on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): a wave owns one tap of one chunk of
// SP_CHUNK output sites and the whole Cin x Cout tile; the K dimension of the MFMA is the list of ACTIVE (site, neighbour) pairs of
// that tap (compacted with ballot + prefix popcount: site order is kept, so the summation order is fixed), four pairs per
// instruction: lane (m = lane & 15, q = lane >> 4) supplies in[nbr of pair q][16 a + m] as A and d_out[site of pair q][16 b + m] as
// B -- both are 64-byte row segments, no transposes.  Partials per chunk as before, summed in chunk order by the reduce kernel.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void sparse_wgrad_mfma_kernel(const float* __restrict__ in, const int* __restrict__ nbr,
                                                                const float* __restrict__ d_out, int n_out, float* __restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int CA = CIN / 16, CB = COUT / 16, G = 4;  // G groups of four pairs in flight
  __shared__ int s_i[4][SP_CHUNK], s_o[4][SP_CHUNK];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.x * 4 + wave, chunk = blockIdx.y, s0 = chunk * SP_CHUNK;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < SP_CHUNK / 64; ++j) {
    const int site = s0 + j * 64 + lane;
    const int nb = (k < 27 && site < n_out) ? nbr[(long)site * 27 + k] : -1;
    const unsigned long long mk = __ballot(nb >= 0);
    if (nb >= 0) {
      const int pos = cnt + __popcll(mk & ((1ull << lane) - 1ull));
      s_i[wave][pos] = nb;
      s_o[wave][pos] = site;
    }
    cnt += __popcll(mk);
  }
  __syncthreads();
  if (k >= 27) return;
  const int m = lane & 15, q = lane >> 4;
  f32x4 acc[CA][CB];
#pragma unroll
  for (int a = 0; a < CA; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < cnt; base += 4 * G) {
    float av[G][CA], bv[G][CB];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int p = base + 4 * g + q;
      const bool ok = p < cnt;
      const int i = ok ? s_i[wave][p] : 0, o = ok ? s_o[wave][p] : 0;
#pragma unroll
      for (int a = 0; a < CA; ++a) av[g][a] = ok ? in[(long)i * CIN + a * 16 + m] : 0.f;
#pragma unroll
      for (int b = 0; b < CB; ++b) bv[g][b] = ok ? d_out[(long)o * COUT + b * 16 + m] : 0.f;
    }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int a = 0; a < CA; ++a)
#pragma unroll
        for (int b = 0; b < CB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g][a], bv[g][b], acc[a][b], 0, 0, 0);
  }
  float* dst = part + ((long)chunk * 27 + k) * CIN * COUT;
#pragma unroll
  for (int a = 0; a < CA; ++a)
#pragma unroll
    for (int b = 0; b < CB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(long)(a * 16 + 4 * q + r) * COUT + b * 16 + m] = acc[a][b][r];
#endif
}

// inverse of a strided layer's table: inv[i][k] = the output site that reads input row i at tap k (at most one), else -1
__global__ void sparse_inverse_table_kernel(const int* __restrict__ nbr_down, int n_out, int* __restrict__ inv) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n_out * 27; e += (long)gridDim.x * blockDim.x) {
    const int i = nbr_down[e];
    if (i >= 0) inv[(long)i * 27 + (int)(e % 27)] = (int)(e / 27);
  }
}
// Several vertices in one voxel: the later ones' rows equal the representative's (the one every neighbour lookup returns, the
// centre tap included), so their output gradients belong to the representative's row.  After this the level's table is symmetric
// over the rows that carry gradient, which is what the gather-form data-gradient needs.
__global__ void sparse_fold_dups_kernel(float* __restrict__ d, const int* __restrict__ nbr, int n, int C) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (long)n * C; e += (long)gridDim.x * blockDim.x) {
    const int s = (int)(e / C), c = (int)(e - (long)s * C);
    const int r = nbr[(long)s * 27 + 13];
    if (r != s && r >= 0) {
      unsafeAtomicAdd(d + (long)r * C + c, d[e]);
      d[e] = 0.f;
    }
  }
}

__global__ void sparse_wgrad_reduce_kernel(const float* __restrict__ part, int nchunk, long n, float* __restrict__ dw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < nchunk; ++c) a += part[(long)c * n + i];
    dw[i] = a;
  }
}

// packed [27][Cin][Cout] gradient -> added to the parameter's own layout (engine_weights.hip: build_sparse_layer)
__global__ void sparse_w_unpack_add_kernel(const float* __restrict__ pk, int Cin, int Cout, int layout, float* __restrict__ dst) {
  const long total = (long)27 * Cin * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout), ci = (int)((i / Cout) % Cin), k = (int)(i / ((long)Cout * Cin));
    const long d = layout == 0 ? ((long)co * Cin + ci) * 27 + k : layout == 1 ? ((long)co * 27 + k) * Cin + ci : i;
    dst[d] += pk[i];
  }
}

// ---- transposing im2col of a 3-D conv (k3, pad 1, stride s): dst[(ci*27 + tap)][r] = X[b, zo*s+kz-1, yo*s+ky-1, xo*s+kx-1, ci] ----
template <typename T>
__global__ __launch_bounds__(256) void im2colT3d_kernel(const T* __restrict__ src, long ld, int B, int D, int H, int W, int C, int stride,
                                                        int Do, int Ho, int Wo, half_t* __restrict__ dst, int Rp) {
  __shared__ half_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int kz = tap / 9 - 1, ky = (tap / 3) % 3 - 1, kx = tap % 3 - 1;
  const int R = B * Do * Ho * Wo;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int rr = i >> 6, cc = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    half_t v = (half_t)0;
    if (r < R && c < C) {
      const int xo = r % Wo, yo = (r / Wo) % Ho, zo = (r / (Wo * Ho)) % Do, b = r / (Wo * Ho * Do);
      const int z = zo * stride + kz, y = yo * stride + ky, x = xo * stride + kx;
      if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) v = (half_t)ldf(src + ((((long)b * D + z) * H + y) * W + x) * ld + c);
    }
    tile[rr][cc] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int cc = i >> 6, rr = i & 63;
    const int r = r0 + rr, c = c0 + cc;
    if (c < C && r < Rp) dst[((long)c * 27 + tap) * Rp + r] = tile[rr][cc];
  }
}

// with 8- / 16-byte accesses (see im2colT_vec_kernel, k_bwd.hip): four channels per gather, eight rows per store; same values
template <typename T>
__global__ __launch_bounds__(256) void im2colT3d_vec_kernel(const T* __restrict__ src, long ld, int B, int D, int H, int W, int C, int stride,
                                                            int Do, int Ho, int Wo, half_t* __restrict__ dst, int Rp) {
  __shared__ half_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tap = blockIdx.z;
  const int kz = tap / 9 - 1, ky = (tap / 3) % 3 - 1, kx = tap % 3 - 1;
  const int R = B * Do * Ho * Wo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int rr = i >> 4, cc = (i & 15) * 4;
    const int r = r0 + rr, c = c0 + cc;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < R && c < C) {
      const int xo = r % Wo, yo = (r / Wo) % Ho, zo = (r / (Wo * Ho)) % Do, b = r / (Wo * Ho * Do);
      const int z = zo * stride + kz, y = yo * stride + ky, x = xo * stride + kx;
      if (z >= 0 && z < D && y >= 0 && y < H && x >= 0 && x < W) {
        const T* p = src + ((((long)b * D + z) * H + y) * W + x) * ld + c;
        if constexpr (sizeof(T) == 4) {
          const float4 q = *(const float4*)p;
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
          const h4 q = *(const h4*)p;
          v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[rr][cc + e] = (half_t)v[e];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = threadIdx.x + 256 * j;
    const int cc = i >> 3, g = i & 7;
    const int c = c0 + cc;
    if (c >= C) continue;
    h8 hv;
#pragma unroll
    for (int k = 0; k < 8; ++k) hv[k] = tile[g * 8 + k][cc];
    *(h8*)(dst + ((long)c * 27 + tap) * Rp + r0 + g * 8) = hv;
  }
}

// out[r][k] (+)= sum_n g[r][n] w[n][k]  (w fp16 [N][K], the LinW pack) for a handful of rows (FiLM / step-MLP adjoints).
// One workgroup = 16 consecutive k x 16 lanes over n, combined through LDS in lane order.
__global__ __launch_bounds__(256) void small_linear_bwd_kernel(const float* __restrict__ g, long ldg, int N, const half_t* __restrict__ w,
                                                               int K, float* __restrict__ out, long ldo, int accum) {
  __shared__ float s_p[16][17];
  const int r = blockIdx.y, kk = threadIdx.x & 15, nl = threadIdx.x >> 4, k = blockIdx.x * 16 + kk;
  float acc = 0.f;
  if (k < K)
    for (int n = nl; n < N; n += 16) acc += g[r * ldg + n] * (float)w[(long)n * K + k];
  s_p[nl][kk] = acc;
  __syncthreads();
  if (nl == 0 && k < K) {
    float a = 0.f;
    for (int i = 0; i < 16; ++i) a += s_p[i][kk];
    out[r * ldo + k] = accum ? out[r * ldo + k] + a : a;
  }
}

}  // namespace

int cbwd_frustum_scatter(const float* d_out, const ViewCam* cams, const int* view_idx, int TN, int D, int S, int V, float vol_len, int persp,
                         float* d_vol, hipStream_t s) {
  const long threads = (long)TN * D * S * S * 16;
  hipLaunchKernelGGL(frustum_scatter_kernel, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, d_out, cams, view_idx, TN, D, S, V,
                     vol_len, persp, d_vol);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_latent_scatter(const float* d_vol, const int* grid, int gd, int gh, int gw, const float* min_xyz, const int* out_sh, float voxel,
                        int V, float vol_len, float* d_rows, hipStream_t s) {
  const int C = 64;
  const size_t total = (size_t)V * V * V * C;
  hipLaunchKernelGGL(latent_scatter_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, s, d_vol, grid, gd, gh, gw, min_xyz[0],
                     min_xyz[1], min_xyz[2], (float)out_sh[2], (float)out_sh[1], (float)out_sh[0], voxel, V, vol_len, C, d_rows);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_vertex_scatter(const float* d_out, const ViewCam* cams, const int* view_idx, int n_views, const float* verts, int Nv, int V,
                        float vol_len, int S, int persp, float* d_feats, hipStream_t s) {
  const int lds = S * S * 16 * (int)sizeof(float);
  if (lds > 160 * 1024) return mvd_fail("vertex scatter: feature map too large for the LDS accumulation");
  static bool attr_done[MVD_MAX_DEVICES] = {false};
  bool& attr_set = attr_done[mvd_current_device()];
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)vertex_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(vertex_scatter_kernel, dim3(n_views, 16), dim3(1024), lds, s, d_out, cams, view_idx, verts, Nv, V, vol_len, S, persp,
                     d_feats);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// part: scratch of cbwd_fuse_scratch_floats(Nv) floats (only read when dw / db are given)
int cbwd_fuse(const float* d_fused, const float* vf, const float* w, int n_views, int Nv, int total_views, float* d_vf, float* dw, float* db,
              float* part, hipStream_t s) {
  hipLaunchKernelGGL(fuse_bwd_dvf_kernel, dim3(cdiv(Nv * 16, 256)), dim3(256), 0, s, d_fused, w, n_views, Nv, total_views, d_vf);
  if (dw && db) {
    if (!part) return mvd_fail("fuse backward: scratch for the weight-gradient partials missing");
    const int nchunk = cdiv(Nv, FW_CH);
    hipLaunchKernelGGL(fuse_bwd_w_kernel, dim3(nchunk), dim3(256), 0, s, d_fused, vf, n_views, Nv, total_views, part);
    hipLaunchKernelGGL(fuse_bwd_w_reduce_kernel, dim3(1), dim3(256), 0, s, part, nchunk, dw, db);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_fuse_scratch_floats(int Nv) { return cdiv(Nv, FW_CH) * 272; }
int cbwd_bn_scratch_floats(int n, int C) { return cdiv(n, BNB_CH) * 2 * C; }
// stats: [mean | rstd] from launch_bn_rows_relu; scratch: cbwd_bn_scratch_floats(n, C) floats
int cbwd_bn_rows_relu(const float* xraw, float* dy, int n, int C, const float* gamma, const float* beta, const float* stats,
                      float* scratch, float* dgamma, float* dbeta, hipStream_t s) {
  if (n <= 0 || C <= 0) return 0;
  if (C > 256 || 256 % C) return mvd_fail("bn_rows_relu backward: the channel count must divide 256");
  const int nchunk = cdiv(n, BNB_CH);
  hipLaunchKernelGGL(bn_rows_bwd_sums_kernel, dim3(nchunk), dim3(256), 0, s, xraw, dy, n, C, gamma, beta, stats, scratch);
  hipLaunchKernelGGL(bn_rows_bwd_apply_kernel, dim3(nchunk), dim3(256), 0, s, xraw, dy, n, C, gamma, beta, stats, scratch, nchunk, dgamma, dbeta);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_sparse_wgrad_chunks(int n_out) { return cdiv(n_out, SP_CHUNK); }
// dw_part: scratch of cbwd_sparse_wgrad_chunks(n_out) * 27 * Cin * Cout floats
int cbwd_sparse_conv(const float* in, const int* nbr, const float* d_out, int n_out, int Cin, int Cout, const float* w, float* d_in,
                     float* dw_packed, float* dw_part, hipStream_t s) {
  if (n_out <= 0) return 0;
  if (Cin > 256 || Cout > 256) return mvd_fail("sparse conv backward: channel count > 256");
  if (d_in) hipLaunchKernelGGL(sparse_conv_dgrad_kernel, dim3(n_out), dim3(256), 0, s, d_out, nbr, Cin, Cout, w, d_in);
  if (dw_packed) {
    const int nchunk = cdiv(n_out, SP_CHUNK);
    const long n = (long)27 * Cin * Cout;
    hipLaunchKernelGGL(sparse_conv_wgrad_kernel, dim3(27, cdiv(Cin * Cout, 256), nchunk), dim3(256), 0, s, in, nbr, d_out, n_out, Cin, Cout,
                       dw_part);
    hipLaunchKernelGGL(sparse_wgrad_reduce_kernel, dim3(gridn((size_t)n)), dim3(256), 0, s, dw_part, nchunk, n, dw_packed);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
// Matrix-core form (channel counts sparse_mfma_takes): weight gradient only; the data gradient is launch_sparse_conv on the
// layer's transposed fragments (SparseLayerW::wd) -- no atomics, no zero fill.
int cbwd_sparse_wgrad_mfma(const float* in, const int* nbr, const float* d_out, int n_out, int Cin, int Cout, float* dw_packed,
                           float* dw_part, hipStream_t s) {
  if (n_out <= 0) return 0;
  const int nchunk = cdiv(n_out, SP_CHUNK);
  const long n = (long)27 * Cin * Cout;
  const dim3 grid(7, nchunk), blk(256);
#define MVD_SPW(CI, CO)                                                                                                    \
  if (Cin == CI && Cout == CO)                                                                                             \
    hipLaunchKernelGGL((sparse_wgrad_mfma_kernel<CI, CO>), grid, blk, 0, s, in, nbr, d_out, n_out, dw_part);               \
  else
  MVD_SPW(16, 16) MVD_SPW(16, 32) MVD_SPW(32, 32) MVD_SPW(32, 64) MVD_SPW(64, 64)
  return mvd_fail("sparse wgrad: no matrix-core kernel for these channel counts");
#undef MVD_SPW
  hipLaunchKernelGGL(sparse_wgrad_reduce_kernel, dim3(gridn((size_t)n)), dim3(256), 0, s, dw_part, nchunk, n, dw_packed);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_sparse_inverse_table(const int* nbr_down, int n_out, int n_in, int* inv, hipStream_t s) {
  HIP_CHECK_RET(hipMemsetAsync(inv, 0xFF, (size_t)n_in * 27 * sizeof(int), s));
  if (n_out > 0) hipLaunchKernelGGL(sparse_inverse_table_kernel, dim3(gridn((size_t)n_out * 27)), dim3(256), 0, s, nbr_down, n_out, inv);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_sparse_fold_dups(float* d, const int* nbr, int n, int C, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(sparse_fold_dups_kernel, dim3(gridn((size_t)n * C)), dim3(256), 0, s, d, nbr, n, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_sparse_w_unpack_add(const float* pk, int Cin, int Cout, int layout, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(sparse_w_unpack_add_kernel, dim3(gridn((size_t)27 * Cin * Cout)), dim3(256), 0, s, pk, Cin, Cout, layout, dst);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_im2colT3d(const void* src, int src_f32, long ld, int B, int D, int H, int W, int C, int stride, half_t* dst, int Rp, hipStream_t s) {
  const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if (Rp < B * Do * Ho * Wo) return mvd_fail("im2colT3d: bad shape");
  dim3 grid(cdiv(Rp, 64), cdiv(C, 64), 27);
  static const bool scalar_only = getenv("MVD_STAGE_SCALAR") != nullptr;
  if (!scalar_only && !(C & 3) && !(ld & 3) && !(Rp & 63) && !((uintptr_t)src & 15) && !((uintptr_t)dst & 15)) {
    if (src_f32) hipLaunchKernelGGL(im2colT3d_vec_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, B, D, H, W, C, stride, Do, Ho, Wo, dst, Rp);
    else hipLaunchKernelGGL(im2colT3d_vec_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, B, D, H, W, C, stride, Do, Ho, Wo, dst, Rp);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (src_f32) hipLaunchKernelGGL(im2colT3d_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ld, B, D, H, W, C, stride, Do, Ho, Wo, dst, Rp);
  else hipLaunchKernelGGL(im2colT3d_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)src, ld, B, D, H, W, C, stride, Do, Ho, Wo, dst, Rp);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int cbwd_small_linear_bwd(const float* g, long ldg, int rows, int N, const half_t* w, int K, float* out, long ldo, int accum, hipStream_t s) {
  hipLaunchKernelGGL(small_linear_bwd_kernel, dim3(cdiv(K, 16), rows), dim3(256), 0, s, g, ldg, N, w, K, out, ldo, accum);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
