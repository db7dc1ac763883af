// Mesh conditioner + per-view frustum network executors (reference SpatialVolumeNet,
// ldm/models/diffusion/morphable_diffusion.py:151-320; networks in ldm/models/diffusion/network.py).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <unordered_map>

#include "engine.h"
#include <chrono>
#include <string>
#include <thread>

namespace {

// ---- 4x4 helpers (host, double) -------------------------------------------------------------------
bool inv4(const double* m, double* out) {
  double a[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = m[i * 4 + j];
      a[i][j + 4] = i == j ? 1.0 : 0.0;
    }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
    if (fabs(a[piv][col]) < 1e-300) return false;
    if (piv != col)
      for (int j = 0; j < 8; ++j) std::swap(a[piv][j], a[col][j]);
    const double d = a[col][col];
    for (int j = 0; j < 8; ++j) a[col][j] /= d;
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double fct = a[r][col];
        for (int j = 0; j < 8; ++j) a[r][j] -= fct * a[col][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = a[i][j + 4];
  return true;
}

void mul4(const double* a, const double* b, double* o) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      o[i * 4 + j] = s;
    }
}

inline long key3(int z, int y, int x) { return ((long)z << 40) | ((long)y << 20) | (long)x; }

// open-addressing map voxel key -> first row with that key (linear probing, power-of-two table): the rule book does
// ~30 lookups per active site, std::unordered_map made that the most expensive part of a mesh upload
struct VoxelMap {
  std::vector<long> keys;
  std::vector<int> vals;
  size_t mask = 0;
  void reset(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    keys.assign(cap, -1);
    vals.resize(cap);
    mask = cap - 1;
  }
  static size_t hash(long k) { return (size_t)((unsigned long)k * 0x9E3779B97F4A7C15ul >> 20); }
  void insert_first(long k, int v) {  // keeps the FIRST value of a key
    size_t i = hash(k) & mask;
    while (keys[i] != -1) {
      if (keys[i] == k) return;
      i = (i + 1) & mask;
    }
    keys[i] = k;
    vals[i] = v;
  }
  int find(long k) const {
    size_t i = hash(k) & mask;
    while (keys[i] != -1) {
      if (keys[i] == k) return vals[i];
      i = (i + 1) & mask;
    }
    return -1;
  }
};

}  // namespace

// ----------------------------------------------------------------------------------------------------
// mvd_set_cameras: construct_project_matrix (utils.py:46-69), the inverse used by create_target_volume
// (utils.py:79-153) and near/far from the camera distance (morphable_diffusion.py:281-299), once per sample.
// (two halves: cams_build validates and converts on the host -- nothing of the context changes --, cams_commit uploads; a batch
//  of samples validates ALL of its samples before the first commit: engine_set_samples)
static int cams_build(mvd_ctx* c, const float* K, const float* RT, int N, std::vector<ViewCam>& cams) {
  cams.assign(N, ViewCam());
  const double ratio = (double)(c->v.input_image_size / 8) / (double)c->v.input_image_size;
  for (int i = 0; i < N; ++i) {
    const float* k = K + i * 16;
    const float* rt = RT + i * 12;
    double RT4[16], K4[16], P4[16], Pinv[16];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 4; ++b) RT4[a * 4 + b] = rt[a * 4 + b];
    RT4[12] = RT4[13] = RT4[14] = 0;
    RT4[15] = 1;
    for (int a = 0; a < 16; ++a) K4[a] = k[a];
    ViewCam& v = cams[i];
    memset(&v, 0, sizeof(v));
    if (c->v.projection == 0) {
      double KS[16] = {0};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) KS[a * 4 + b] = K4[a * 4 + b] * (a < 2 ? ratio : 1.0);
      KS[15] = 1;
      mul4(KS, RT4, P4);  // rows 0..2 = diag(r,r,1) K3 [R|t]; row 3 = (0,0,0,1)
      if (!inv4(P4, Pinv)) return mvd_fail("set_cameras: singular projection matrix");
      for (int a = 0; a < 12; ++a) {
        v.P[a] = (float)P4[a];
        v.Pinv[a] = (float)Pinv[a];
      }
    } else {
      double Kinv[16], RTinv[16];
      mul4(K4, RT4, P4);
      if (!inv4(K4, Kinv) || !inv4(RT4, RTinv)) return mvd_fail("set_cameras: singular K or RT");
      for (int a = 0; a < 12; ++a) {
        v.P[a] = (float)P4[a];
        v.Pinv[a] = (float)RTinv[a];
      }
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) v.Kinv[a * 3 + b] = (float)Kinv[a * 4 + b];
    }
    double pos[3];
    for (int a = 0; a < 3; ++a) pos[a] = -(RT4[0 * 4 + a] * RT4[3] + RT4[1 * 4 + a] * RT4[7] + RT4[2 * 4 + a] * RT4[11]);
    const double dist = sqrt(pos[0] * pos[0] + pos[1] * pos[1] + pos[2] * pos[2]);
    v.near_ = (float)dist - c->v.frustum_volume_length;
    v.far_ = (float)dist + c->v.frustum_volume_length;
  }
  return 0;
}
static int cams_commit(mvd_ctx* c, const std::vector<ViewCam>& cams, hipStream_t s) {
  const int N = (int)cams.size();
  // upload: pinned host image -> device table by ONE stream-ordered copy; both grow only when N does
  mvd_ctx::CamStage& st = c->cam_stage;
  if (N > st.cap) {
    ViewCam *dn = nullptr, *hn = nullptr;
    HIP_CHECK_RET(hipMalloc((void**)&dn, N * sizeof(ViewCam)));
    if (hipHostMalloc((void**)&hn, N * sizeof(ViewCam)) != hipSuccess) {
      hipFree(dn);
      return mvd_fail("set_cameras: pinned allocation failed");
    }
    HIP_CHECK_RET(hipDeviceSynchronize());  // nothing in flight may still read the table being replaced
    if (c->cams) hipFree(c->cams);
    if (st.h) hipHostFree(st.h);
    c->cams = dn;
    st.h = hn;
    st.cap = N;
  }
  if (!st.staged) HIP_CHECK_RET(hipEventCreateWithFlags(&st.staged, hipEventDisableTiming));
  else HIP_CHECK_RET(hipEventSynchronize(st.staged));  // the previous upload has left the host image
  memcpy(st.h, cams.data(), N * sizeof(ViewCam));
  HIP_CHECK_RET(hipMemcpyAsync(c->cams, st.h, N * sizeof(ViewCam), hipMemcpyHostToDevice, s));
  HIP_CHECK_RET(hipEventRecord(st.staged, s));
  c->n_cams = N;
  return 0;
}
int engine_set_cameras(mvd_ctx* c, const float* K, const float* RT, int N, hipStream_t s) {
  std::vector<ViewCam> cams;
  RET_IF(cams_build(c, K, RT, N, cams));
  return cams_commit(c, cams, s);
}

// ----------------------------------------------------------------------------------------------------
// mvd_set_mesh: the "rulebook" spconv would build per call (SubMConv3d k3 / SparseConv3d k3 s2 p1), built
// once per mesh on the host because coord/out_sh/bounds are step-invariant (SURVEY gotcha G15).
void mesh_free(MeshTables& m) {
  hipFree(m.pool);
  if (m.h_pool) hipHostFree(m.h_pool);
  for (int i = 0; i < 2; ++i) hipFree(m.feat[i]);
  if (m.staged) hipEventDestroy(m.staged);
  m = MeshTables();
}

namespace {
// The rule book of one mesh on the host: what spconv's indice generation produces per call (SubMConv3d k3 / SparseConv3d k3 s2
// p1 on three levels), plus the coarse index grid.
struct HostMesh {
  int n_sites[3], shapes[3][3], max_sites = 0;
  std::vector<int> nbr_subm[3], nbr_down[2], grid;
  // scratch
  std::vector<std::array<int, 3>> sites, osites;
  std::vector<int> dense;        // voxel -> first row (dense path)
  std::vector<unsigned char> mark;
  VoxelMap idx;                  // (hash path: grids of more than 2^25 cells)
};

// Pure host code (no context, no HIP): safe to run for several samples on several threads.
// Lookups go through a dense voxel -> row grid of the level (4 bytes per cell; a 100^3 head grid is 4 MB) when it has at
// most 2^25 cells, through the open-addressing map otherwise; the strided level's output sites are enumerated in (z,y,x)
// order either way.
int host_mesh_build(const int32_t* coord, const int32_t* out_sh, int Nv, HostMesh& h, size_t dense_limit = (size_t)1 << 25) {
  for (int i = 0; i < Nv; ++i)
    for (int a = 0; a < 3; ++a)
      if (coord[i * 3 + a] < 0 || coord[i * 3 + a] >= out_sh[a]) return mvd_fail("set_mesh: voxel coordinate outside out_sh");
  for (int a = 0; a < 3; ++a)
    if (out_sh[a] <= 0 || out_sh[a] >= (1 << 20)) return mvd_fail("set_mesh: out_sh out of range");
  auto& sites = h.sites;
  auto& osites = h.osites;
  sites.resize(Nv);
  for (int i = 0; i < Nv; ++i) sites[i] = {coord[i * 3], coord[i * 3 + 1], coord[i * 3 + 2]};
  int shape[3] = {out_sh[0], out_sh[1], out_sh[2]};
  h.max_sites = Nv;
  for (int lvl = 0; lvl < 3; ++lvl) {
    const int ns = (int)sites.size();
    const size_t cells = (size_t)shape[0] * shape[1] * shape[2];
    const bool use_dense = cells <= dense_limit;
    // several vertices in one voxel: the FIRST one is the voxel's representative (spconv's hash table also keeps one row per
    // voxel and resolves every neighbour lookup, the centre tap included, through it; which row wins there is a race).  Later
    // duplicates still get an output row, identical to the representative's.
    if (use_dense) {
      h.dense.assign(cells, -1);
      for (int i = 0; i < ns; ++i) {
        int& d = h.dense[((size_t)sites[i][0] * shape[1] + sites[i][1]) * shape[2] + sites[i][2]];
        if (d < 0) d = i;
      }
    } else {
      h.idx.reset(ns);
      for (int i = 0; i < ns; ++i) h.idx.insert_first(key3(sites[i][0], sites[i][1], sites[i][2]), i);
    }
    const int s0 = shape[0], s1 = shape[1], s2 = shape[2];
    auto find = [&](int z, int y, int x) -> int {
      if ((unsigned)z >= (unsigned)s0 || (unsigned)y >= (unsigned)s1 || (unsigned)x >= (unsigned)s2) return -1;
      return use_dense ? h.dense[((size_t)z * s1 + y) * s2 + x] : h.idx.find(key3(z, y, x));
    };
    h.n_sites[lvl] = ns;
    for (int a = 0; a < 3; ++a) h.shapes[lvl][a] = shape[a];
    std::vector<int>& nbr = h.nbr_subm[lvl];
    nbr.resize((size_t)ns * 27);
    for (int i = 0; i < ns; ++i) {
      int* o = &nbr[(size_t)i * 27];
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) *o++ = find(sites[i][0] + dz, sites[i][1] + dy, sites[i][2] + dx);
    }
    if (lvl == 2) {
      h.grid.assign(cells, -1);
      for (int i = 0; i < ns; ++i) h.grid[((size_t)sites[i][0] * shape[1] + sites[i][1]) * shape[2] + sites[i][2]] = i;
      break;
    }
    // strided conv to the next level: input i feeds output o = (i + 1 - k) / 2 for every tap k that makes it integral and in
    // range; the output sites are the distinct ones, in (z,y,x) order
    const int oshape[3] = {(shape[0] - 1) / 2 + 1, (shape[1] - 1) / 2 + 1, (shape[2] - 1) / 2 + 1};
    const size_t ocells = (size_t)oshape[0] * oshape[1] * oshape[2];
    osites.clear();
    auto outputs_of = [&](const std::array<int, 3>& st, auto&& emit) {
      for (int kz = 0; kz < 3; ++kz) {
        const int nz = st[0] + 1 - kz;
        if ((nz & 1) || nz < 0 || nz / 2 >= oshape[0]) continue;
        for (int ky = 0; ky < 3; ++ky) {
          const int ny = st[1] + 1 - ky;
          if ((ny & 1) || ny < 0 || ny / 2 >= oshape[1]) continue;
          for (int kx = 0; kx < 3; ++kx) {
            const int nx = st[2] + 1 - kx;
            if ((nx & 1) || nx < 0 || nx / 2 >= oshape[2]) continue;
            emit(nz / 2, ny / 2, nx / 2);
          }
        }
      }
    };
    if (ocells <= dense_limit) {
      h.mark.assign(ocells, 0);
      for (auto& st : sites) outputs_of(st, [&](int oz, int oy, int ox) { h.mark[((size_t)oz * oshape[1] + oy) * oshape[2] + ox] = 1; });
      for (int oz = 0; oz < oshape[0]; ++oz)
        for (int oy = 0; oy < oshape[1]; ++oy) {
          const unsigned char* row = &h.mark[((size_t)oz * oshape[1] + oy) * oshape[2]];
          for (int ox = 0; ox < oshape[2]; ++ox)
            if (row[ox]) osites.push_back({oz, oy, ox});
        }
    } else {
      std::vector<long> keys;
      for (auto& st : sites) outputs_of(st, [&](int oz, int oy, int ox) { keys.push_back(key3(oz, oy, ox)); });
      std::sort(keys.begin(), keys.end());
      keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
      for (long k : keys) osites.push_back({(int)(k >> 40), (int)((k >> 20) & 0xFFFFF), (int)(k & 0xFFFFF)});
    }
    std::vector<int>& dn = h.nbr_down[lvl];
    dn.resize(osites.size() * 27);
    for (size_t o = 0; o < osites.size(); ++o) {
      int* d = &dn[o * 27];
      for (int kz = 0; kz < 3; ++kz)
        for (int ky = 0; ky < 3; ++ky)
          for (int kx = 0; kx < 3; ++kx) *d++ = find(2 * osites[o][0] - 1 + kz, 2 * osites[o][1] - 1 + ky, 2 * osites[o][2] - 1 + kx);
    }
    sites.swap(osites);
    for (int a = 0; a < 3; ++a) shape[a] = oshape[a];
    h.max_sites = std::max(h.max_sites, (int)sites.size());
  }
  return 0;
}

// The tables of `h` become the ACTIVE slot's mesh: ONE copy in the order of stream `s` out of the slot's pinned host image
// (launches enqueued on `s` before this call still read the old tables, later ones the new tables).  A failure (allocation)
// leaves the active mesh exactly as it was.
int mesh_commit(mvd_ctx* c, HostMesh& h, const float* vertices, const int32_t* out_sh, const float* bounds, int Nv, hipStream_t s) {
  std::vector<int>*nbr_subm = h.nbr_subm, *nbr_down = h.nbr_down;
  std::vector<int>& grid = h.grid;
  const int* n_sites = h.n_sites;
  const int(*shapes)[3] = h.shapes;
  const int max_sites = h.max_sites;
  // ---- pool layout (ints; every table 64-byte aligned): verts | nbr_subm[0..2] | nbr_down[0..1] | grid2 ----
  auto up16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
  size_t off[7], total = 0;
  const size_t lens[7] = {(size_t)Nv * 3, nbr_subm[0].size(), nbr_subm[1].size(), nbr_subm[2].size(), nbr_down[0].size(),
                          nbr_down[1].size(), grid.size()};
  for (int i = 0; i < 7; ++i) {
    off[i] = total;
    total += up16(std::max<size_t>(lens[i], 1));
  }
  MeshTables& m = c->mesh;
  if (!c->volume) {
    const int V = c->v.spatial_volume_size;
    if (hipMalloc((void**)&c->volume, (size_t)V * V * V * 64 * sizeof(float)) != hipSuccess)
      return mvd_fail("set_mesh: volume allocation failed");
  }
  if (total > m.pool_cap || (size_t)max_sites * 64 > m.feat_cap) {  // grow (new first, swap on success)
    const size_t ncap = std::max(total + total / 4, m.pool_cap), nfeat = std::max((size_t)max_sites * 64 * 5 / 4, m.feat_cap);
    int *dp = nullptr, *hp = nullptr;
    float* f[2] = {nullptr, nullptr};
    bool ok = hipMalloc((void**)&dp, ncap * sizeof(int)) == hipSuccess && hipHostMalloc((void**)&hp, ncap * sizeof(int)) == hipSuccess &&
              hipMalloc((void**)&f[0], nfeat * sizeof(float)) == hipSuccess && hipMalloc((void**)&f[1], nfeat * sizeof(float)) == hipSuccess;
    if (!ok) {
      hipFree(dp);
      if (hp) hipHostFree(hp);
      hipFree(f[0]);
      hipFree(f[1]);
      return mvd_fail("set_mesh: table allocation failed");
    }
    HIP_CHECK_RET(hipDeviceSynchronize());  // nothing in flight may still read the tables being replaced
    hipFree(m.pool);
    if (m.h_pool) hipHostFree(m.h_pool);
    hipFree(m.feat[0]);
    hipFree(m.feat[1]);
    m.pool = dp;
    m.h_pool = hp;
    m.feat[0] = f[0];
    m.feat[1] = f[1];
    m.pool_cap = ncap;
    m.feat_cap = nfeat;
  }
  if (!m.staged) HIP_CHECK_RET(hipEventCreateWithFlags(&m.staged, hipEventDisableTiming));
  else HIP_CHECK_RET(hipEventSynchronize(m.staged));  // the previous upload has left the host image
  memcpy(m.h_pool + off[0], vertices, (size_t)Nv * 3 * sizeof(float));
  const std::vector<int>* src[6] = {&nbr_subm[0], &nbr_subm[1], &nbr_subm[2], &nbr_down[0], &nbr_down[1], &grid};
  for (int i = 0; i < 6; ++i)
    if (!src[i]->empty()) memcpy(m.h_pool + off[i + 1], src[i]->data(), src[i]->size() * sizeof(int));
  // readers on other streams: the volume build of the caller's communication stream reads these tables; the hand-over event it
  // registered (mvd_set_volume_ready_event) is recorded behind that build, so the upload waits for it (an upload on
  // another stream than the one that ran the last step could otherwise overtake the last reader)
  if (c->vol_ready) HIP_CHECK_RET(hipStreamWaitEvent(s, c->vol_ready, 0));
  HIP_CHECK_RET(hipMemcpyAsync(m.pool, m.h_pool, total * sizeof(int), hipMemcpyHostToDevice, s));
  HIP_CHECK_RET(hipEventRecord(m.staged, s));
  m.Nv = Nv;
  m.verts = (float*)(m.pool + off[0]);
  for (int l = 0; l < 3; ++l) {
    m.nbr_subm[l] = m.pool + off[1 + l];
    m.n_sites[l] = n_sites[l];
    for (int a = 0; a < 3; ++a) m.shape[l][a] = shapes[l][a];
  }
  for (int l = 0; l < 2; ++l) m.nbr_down[l] = m.pool + off[4 + l];
  m.grid2 = m.pool + off[6];
  for (int a = 0; a < 3; ++a) {
    m.min_xyz[a] = bounds[a];
    m.out_sh[a] = out_sh[a];
  }
  return 0;
}

}  // namespace

// Transactional: validation and the host-side build touch nothing of the active mesh; a failed call (bad coordinates,
// allocation failure) leaves it exactly as it was.
int engine_set_mesh(mvd_ctx* c, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds, int Nv,
                    hipStream_t s) {
  static const bool timing = getenv("MVD_MESH_TIMING") != nullptr;  // development aid: host phases of this call on stderr
  const auto t0 = std::chrono::steady_clock::now();
  static thread_local HostMesh h;
  RET_IF(host_mesh_build(coord, out_sh, Nv, h));
  const auto t1 = std::chrono::steady_clock::now();
  RET_IF(mesh_commit(c, h, vertices, out_sh, bounds, Nv, s));
  if (timing)
    fprintf(stderr, "[set_mesh] Nv %d sites %d/%d/%d: build %.2f ms, stage + enqueue %.2f ms\n", Nv, h.n_sites[0], h.n_sites[1],
            h.n_sites[2], std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return 0;
}

// host-only probes of the rule book (tests without a GPU): build, then read the tables of the last build on this thread
static thread_local HostMesh t_probe;
int engine_rulebook_build(const int32_t* coord, const int32_t* out_sh, int Nv, int force_hash, int32_t* n_sites, int64_t* lens) {
  RET_IF(host_mesh_build(coord, out_sh, Nv, t_probe, force_hash ? 0 : (size_t)1 << 25));
  for (int l = 0; l < 3; ++l) {
    n_sites[l] = t_probe.n_sites[l];
    lens[l] = (int64_t)t_probe.nbr_subm[l].size();
  }
  lens[3] = (int64_t)t_probe.nbr_down[0].size();
  lens[4] = (int64_t)t_probe.nbr_down[1].size();
  lens[5] = (int64_t)t_probe.grid.size();
  return 0;
}
int engine_rulebook_table(int which, int32_t* out) {
  const std::vector<int>* v = which < 3 ? &t_probe.nbr_subm[which] : which < 5 ? &t_probe.nbr_down[which - 3] : &t_probe.grid;
  if (which < 0 || which > 5) return mvd_fail("mvd_rulebook_table: table index 0..5");
  if (!v->empty()) memcpy(out, v->data(), v->size() * sizeof(int));
  return 0;
}

int engine_select_sample(mvd_ctx* c, int slot);
// mvd_set_samples_async: the tables of B samples (a training step's new batch) -- the rule books are built on B host threads,
// then committed slot by slot in the order of stream `s`.  Validation of ALL samples precedes the first commit.
int engine_set_samples(mvd_ctx* c, int B, const int* slots, const float* const* vertices, const int32_t* const* coord,
                       const int32_t* const* out_sh, const float* const* bounds, const int* Nv, const float* const* K,
                       const float* const* RT, int N, hipStream_t s) {
  static const bool timing = getenv("MVD_MESH_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  static thread_local std::vector<HostMesh> hs;  // scratch kept between steps
  if ((int)hs.size() < B) hs.resize(B);
  HostMesh* const hp = hs.data();  // (a thread_local name inside a worker's lambda would be the WORKER's instance)
  std::vector<int> rc(B, 0);
  std::vector<std::string> err(B);
  {
    std::vector<std::thread> th;
    for (int i = 1; i < B; ++i)
      th.emplace_back([&, i]() {
        rc[i] = host_mesh_build(coord[i], out_sh[i], Nv[i], hp[i]);
        if (rc[i]) err[i] = mvd_last_error();  // thread-local message: carry it to the caller's thread
      });
    rc[0] = host_mesh_build(coord[0], out_sh[0], Nv[0], hp[0]);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < B; ++i)
    if (rc[i]) return i ? mvd_fail(err[i].c_str()) : rc[i];
  // every sample's cameras are validated and converted before the first slot is touched (a singular K / RT in sample 3 must not
  // leave samples 0-2 committed), and the caller's slot is active again on every exit path
  std::vector<std::vector<ViewCam>> cams(B);
  for (int i = 0; i < B; ++i) RET_IF(cams_build(c, K[i], RT[i], N, cams[i]));
  const auto t1 = std::chrono::steady_clock::now();
  struct SlotGuard {
    mvd_ctx* c;
    int slot;
    ~SlotGuard() { engine_select_sample(c, slot); }
  } guard{c, c->cur_slot};
  for (int i = 0; i < B; ++i) {
    RET_IF(engine_select_sample(c, slots[i]));
    RET_IF(mesh_commit(c, hp[i], vertices[i], out_sh[i], bounds[i], Nv[i], s));
    RET_IF(cams_commit(c, cams[i], s));
  }
  if (timing)
    fprintf(stderr, "[set_samples] %d samples: build (threads) %.2f ms, commit %.2f ms\n", B,
            std::chrono::duration<double, std::milli>(t1 - t0).count(),
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
  return 0;
}

// mvd_select_sample: park the active tables, activate another slot (empty slots start with no mesh / no cameras)
int engine_select_sample(mvd_ctx* c, int slot) {
  if (slot < 0 || slot >= 64) return mvd_fail("mvd_select_sample: slot out of range (0..63)");
  if (slot == c->cur_slot) return 0;
  const int need = std::max(slot, c->cur_slot) + 1;
  if ((int)c->slots.size() < need) c->slots.resize(need);
  mvd_ctx::SampleSlot& cur = c->slots[c->cur_slot];
  cur.mesh = c->mesh;
  cur.cams = c->cams;
  cur.n_cams = c->n_cams;
  cur.cam_stage = c->cam_stage;
  cur.volume = c->volume;
  mvd_ctx::SampleSlot& nxt = c->slots[slot];
  c->mesh = nxt.mesh;
  c->cams = nxt.cams;
  c->n_cams = nxt.n_cams;
  c->cam_stage = nxt.cam_stage;
  c->volume = nxt.volume;
  nxt = mvd_ctx::SampleSlot();  // the active copy is the owner now
  c->cur_slot = slot;
  return 0;
}

// ----------------------------------------------------------------------------------------------------
// NoisyTargetViewEncoder (network.py:181-207) for n_local views + fused unprojection/vertex gather + this
// rank's share of the view mean (morphable_diffusion.py:203-231).
// the whole 2-D encoder as one launch (k_enc.hip) applies: 32 x 32 latents, the reference's 8 -> 16 -> ... -> 16 channel plan
bool engine_encoder_is_fused(const mvd_ctx* c) {
  static const bool no_fused_enc = getenv("MVD_NO_FUSED_ENC") != nullptr;
  bool fused_enc = !no_fused_enc && c->has_cond && c->u.image_size == 32 && c->enc_init.Cin == 8 && c->enc_init.N == 16 &&
                   c->enc_init.taps == 9 && !c->enc_init.xp && c->enc_final.N == 16 && c->enc_final.Cin == 16 && !c->enc_final.xp;
  for (int i = 0; i < 3 && fused_enc; ++i)
    fused_enc = c->enc_blocks[i].c1.Cin == 16 && c->enc_blocks[i].c1.N == 16 && !c->enc_blocks[i].c1.xp &&
                c->enc_blocks[i].c2.Cin == 16 && c->enc_blocks[i].c2.N == 16 && !c->enc_blocks[i].c2.xp;
  return fused_enc;
}

// pre_own (optional, n_local * 48 floats): the FiLM rows in storage of the caller's instead of the shared workspace -- with it
// (and the fused encoder) the call touches no workspace memory, i.e. it may run on a stream beside a UNet pass of the same context
int engine_target_encoder(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed, int n_local, float* feats,
                          hipStream_t s, float* pre_own) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  WsScope ws_scope(c);
  const int S = c->u.image_size, HW = S * S, rows = n_local * HW, td = c->v.time_dim, vd = c->v.view_dim;
  const bool fused_enc = engine_encoder_is_fused(c);
  if (pre_own && !fused_enc) return mvd_fail("engine_target_encoder: caller-owned scratch needs the one-launch encoder");
  float *x8 = nullptr, *h = nullptr, *h2 = nullptr, *r1 = nullptr, *pre = pre_own;
  half_t* a = nullptr;
  if (!fused_enc) {
    x8 = ws_alloc<float>(c, (size_t)rows * 8);
    h = ws_alloc<float>(c, (size_t)rows * 16);
    h2 = ws_alloc<float>(c, (size_t)rows * 16);
    r1 = ws_alloc<float>(c, (size_t)rows * 16);
    a = ws_alloc<half_t>(c, (size_t)rows * 16);
    WS_CHECK(x8 && h && h2 && r1 && a);
  }
  if (!pre) pre = ws_alloc<float>(c, (size_t)n_local * 48);
  WS_CHECK(pre);
  // x + time_embed(t) + view_embed(v) of all three blocks in two launches (the step embedding is shared by
  // all views of the sample): pre[v][16*i + c]
  RET_IF(launch_small_linear(t_embed, td, -n_local, td, c->enc_t.w, c->enc_t.bias, 48, ACT_NONE, pre, 48, 0, s));
  RET_IF(launch_small_linear(v_embed, vd, n_local, vd, c->enc_v.w, c->enc_v.bias, 48, ACT_NONE, pre, 48, 1, s));
  if (fused_enc) {  // the whole encoder in one launch, one workgroup per view (k_enc.hip)
    const half_t* w[8];
    const float *bias[8], *gamma[7], *beta[7];
    int cin[8];
    const ConvW* cw[8] = {&c->enc_init, &c->enc_blocks[0].c1, &c->enc_blocks[0].c2, &c->enc_blocks[1].c1, &c->enc_blocks[1].c2,
                          &c->enc_blocks[2].c1, &c->enc_blocks[2].c2, &c->enc_final};
    const NormW* nw[7] = {&c->enc_blocks[0].n1, &c->enc_blocks[0].n2, &c->enc_blocks[1].n1, &c->enc_blocks[1].n2,
                          &c->enc_blocks[2].n1, &c->enc_blocks[2].n2, &c->enc_final_norm};
    for (int i = 0; i < 8; ++i) {
      w[i] = cw[i]->w;
      bias[i] = cw[i]->bias;
      cin[i] = cw[i]->Cin;
    }
    for (int i = 0; i < 7; ++i) {
      gamma[i] = nw[i]->g;
      beta[i] = nw[i]->b;
    }
    return launch_target_encoder(x_noisy, pre, n_local, w, bias, cin, gamma, beta, feats, s);
  }
  RET_IF(launch_nchw_to_nhwc(x_noisy, n_local, 4, HW, x8, 8, 8, s));
  GemmArgs g;
  // no split-K anywhere in the encoder: a view's features must not depend on how many views share the launch (the sharded
  // step is bit-identical to the single-GPU one only if every per-view quantity is)
  g.a = x8; g.a_f32 = 1; g.lda = 8; g.w = &c->enc_init; g.out = h; g.ldc = 16; g.force_splitk = 1;
  RET_IF(run_conv2d(c, g, n_local, S, S, 1, 0, s));
  float* cur = h;
  float* nxt = h2;
  for (int i = 0; i < 3; ++i) {
    const EncBlockW& e = c->enc_blocks[i];
    RET_IF(run_group_norm(c, cur, 16, n_local, HW, e.n1, 8, 1e-5f, ACT_SILU, pre + 16 * i, a, 16, s, 48));
    g = GemmArgs();
    g.a = a; g.lda = 16; g.w = &e.c1; g.out = r1; g.ldc = 16; g.force_splitk = 1;
    RET_IF(run_conv2d(c, g, n_local, S, S, 1, 0, s));
    RET_IF(run_group_norm(c, r1, 16, n_local, HW, e.n2, 8, 1e-5f, ACT_SILU, nullptr, a, 16, s));
    g = GemmArgs();
    g.a = a; g.lda = 16; g.w = &e.c2; g.out = nxt; g.ldc = 16; g.resid = cur; g.ldr = 16; g.force_splitk = 1;
    RET_IF(run_conv2d(c, g, n_local, S, S, 1, 0, s));
    std::swap(cur, nxt);
  }
  RET_IF(run_group_norm(c, cur, 16, n_local, HW, c->enc_final_norm, 8, 1e-5f, ACT_SILU, nullptr, a, 16, s));
  g = GemmArgs();
  g.a = a; g.lda = 16; g.w = &c->enc_final; g.out = feats; g.ldc = 16; g.force_splitk = 1;
  return run_conv2d(c, g, n_local, S, S, 1, 0, s);
}

int engine_vertex_features(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed,
                           const int32_t* view_idx_dev, int n_local, int add_bias, float* fused_out, hipStream_t s,
                           float* vf_out) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  if (!c->mesh.Nv || !c->cams) return mvd_fail("mvd_set_mesh / mvd_set_cameras must be called first");
  WsScope ws_scope(c);
  const int S = c->u.image_size, rows = n_local * S * S;
  // With the one-launch encoder and a caller-provided vf_out the stage runs out of a scratch of the context's own (grown on
  // demand, never shrunk) instead of the shared workspace: it may then be enqueued on the communication stream while the UNet of
  // the same context runs on the caller's (mvd_vertex_features_stream_safe; the step's head leaves the critical path)
  float *feats = nullptr, *pre_own = nullptr;
  if (vf_out && engine_encoder_is_fused(c)) {
    const size_t need = (size_t)rows * 16 + (size_t)n_local * 48;
    if (c->enc_scratch_cap < need) {
      if (c->enc_scratch) HIP_CHECK_RET(hipFree(c->enc_scratch));
      c->enc_scratch = nullptr;
      c->enc_scratch_cap = 0;
      HIP_CHECK_RET(hipMalloc((void**)&c->enc_scratch, need * sizeof(float)));
      c->enc_scratch_cap = need;
    }
    feats = c->enc_scratch;
    pre_own = c->enc_scratch + (size_t)rows * 16;
  } else {
    feats = ws_alloc<float>(c, (size_t)rows * 16);
  }
  // vf_out: the per-view vertex features themselves [n_local][Nv][16] (the all-gather variant of the view exchange)
  float* vf = vf_out ? vf_out : ws_alloc<float>(c, (size_t)n_local * c->mesh.Nv * 16);
  WS_CHECK(feats && vf);
  RET_IF(engine_target_encoder(c, x_noisy, t_embed, v_embed, n_local, feats, s, pre_own));
  RET_IF(launch_vertex_gather(feats, c->cams, view_idx_dev, n_local, c->mesh.verts, c->mesh.Nv, c->v.spatial_volume_size,
                              c->v.spatial_volume_length, S, c->v.projection == 0, vf, s));
  if (fused_out)
    RET_IF(launch_fuse_views(vf, n_local, c->mesh.Nv, c->v.num_views, c->fuse_w, add_bias ? c->fuse_b : nullptr, fused_out,
                             0, s));
  return 0;
}

// SMPLFeatureExtractor (network.py:41-72) on the per-view features of ALL views, summed in view order: the same
// arithmetic, in the same order, whether the views were computed on one GPU or gathered from several
int engine_fuse_vertex_features(mvd_ctx* c, const float* vf_all, int n_views, float* fused_out, hipStream_t s) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  if (!c->mesh.Nv) return mvd_fail("mvd_set_mesh must be called first");
  if (n_views != c->v.num_views) return mvd_fail("mvd_fuse_vertex_features: expects the features of all num_views views");
  return launch_fuse_views(vf_all, n_views, c->mesh.Nv, c->v.num_views, c->fuse_w, c->fuse_b, fused_out, 0, s);
}

// SparseConvNet (network.py:74-96): fused [Nv,16] -> feature rows of the coarsest level's active sites [n_sites[2]][64]
// (*rows_out points into the mesh's ping-pong buffers)
int engine_sparse_net(mvd_ctx* c, const float* fused, hipStream_t s, bool bn_batch_stats, const float** rows_out) {
  MeshTables& m = c->mesh;
  if (!m.Nv) return mvd_fail("mvd_set_mesh must be called first");
  const float* in = fused;
  int lvl = 0, pp = 0;
  for (int i = 0; i < 9; ++i) {
    const SparseLayerW& L = c->sparse[i];
    const int* nbr;
    int n_out;
    if (L.strided) {
      nbr = m.nbr_down[lvl];
      ++lvl;
      n_out = m.n_sites[lvl];
    } else {
      nbr = m.nbr_subm[lvl];
      n_out = m.n_sites[lvl];
    }
    float* out = m.feat[pp];
    if (bn_batch_stats) {  // train mode: raw conv, then BatchNorm1d(eps 1e-3) on the statistics of the active rows + ReLU
      RET_IF(launch_sparse_conv(in, nbr, n_out, L.cin, L.cout, L.w, L.wp, nullptr, nullptr, out, s));
      RET_IF(launch_bn_rows_relu(out, out, n_out, L.cout, L.gamma, L.beta, 1e-3f, nullptr, s, L.rmean, L.rvar, 0.01f));
    } else {
      RET_IF(launch_sparse_conv(in, nbr, n_out, L.cin, L.cout, L.w, L.wp, L.scale, L.shift, out, s));
    }
    in = out;
    pp ^= 1;
  }
  if (bn_batch_stats && c->sparse[0].rmean) ++c->bn_train_calls;
  *rows_out = in;
  return 0;
}

// + latent-code volume gather (morphable_diffusion.py:232-257)
int engine_volume_from_fused(mvd_ctx* c, const float* fused, hipStream_t s, bool bn_batch_stats) {
  MeshTables& m = c->mesh;
  const float* in = nullptr;
  RET_IF(engine_sparse_net(c, fused, s, bn_batch_stats, &in));
  RET_IF(launch_latent_gather(in, m.grid2, m.shape[2][0], m.shape[2][1], m.shape[2][2], m.min_xyz, m.out_sh,
                              c->v.voxel_size, c->v.spatial_volume_size, c->v.spatial_volume_length, c->volume, s));
  return 0;
}

// construct_view_frustum_volume (morphable_diffusion.py:265-320): frustum gather + FrustumTV3DNet
// (network.py:313-347).  Outputs stay channels-last fp32 in the workspace (caller owns the mark).
// FrustumTV3DNet (network.py:313-347) on TN gathered frustum volumes; the caller fills `gath` [TN][D0*S0*S0][64] fp16 through
// `gather` and `pre` [TN][film_total] (x + t_conv(t) + v_conv(v) of all nine blocks) through `film`.  The launch order --
// gather, conv0, FiLM projections -- is the one the side-stream determinism runs of DESIGN section 4 were made with.
template <typename Gather, typename Film>
static int frustum_net(mvd_ctx* c, int TN, FrustumOut* out, hipStream_t s, bool half0, Gather&& gather, Film&& film) {
  const int D0 = c->v.frustum_volume_depth, S0 = c->v.input_image_size / 8;
  const int* fd = c->v.frustum_dims;
  int D[4], S[4];
  for (int l = 0; l < 4; ++l) {
    D[l] = l ? (D[l - 1] - 1) / 2 + 1 : D0;
    S[l] = l ? (S[l - 1] - 1) / 2 + 1 : S0;
  }
  auto vox = [&](int l) { return (size_t)TN * D[l] * S[l] * S[l]; };
  // outputs (persist for the caller)
  float* x[4];
  for (int l = 0; l < 4; ++l) {
    x[l] = ws_alloc<float>(c, vox(l) * fd[l]);
    WS_CHECK(x[l]);
    out->lvl[l] = x[l];
  }
  // the level-0 volume is only ever an MFMA operand of the UNet's context projections: fp16 on request
  half_t* x0h = nullptr;
  if (half0) {
    x0h = ws_alloc<half_t>(c, vox(0) * fd[0]);
    WS_CHECK(x0h);
    out->lvl0_half = x0h;
    out->lvl[0] = nullptr;
  }
  WsScope ws_scope(c);
  half_t* gath = ws_alloc<half_t>(c, vox(0) * 64);
  float* tmp = ws_alloc<float>(c, vox(1) * fd[1]);  // largest intermediate (conv1 output == level-1 size)
  half_t* a = ws_alloc<half_t>(c, vox(0) * fd[0]);
  float* pre = ws_alloc<float>(c, (size_t)TN * c->film_total);
  float* up = ws_alloc<float>(c, vox(0) * fd[0]);
  WS_CHECK(gath && tmp && a && pre && up);
  RET_IF(gather(gath));
  static const bool dbg_sum = getenv("MVD_DEBUG_SUM") != nullptr;
  if (dbg_sum) {
    const int V = c->v.spatial_volume_size;
    c->dbg.push_back({"volume", c->volume, (size_t)V * V * V * 64 * 4});
    c->dbg.push_back({"gath", gath, vox(0) * 64 * 2});
    c->dbg.push_back({"film", pre, (size_t)TN * c->film_total * 4});
    c->dbg.push_back({"x1", x[1], vox(1) * fd[1] * 4});
    c->dbg.push_back({"x2", x[2], vox(2) * fd[2] * 4});
    c->dbg.push_back({"x3", x[3], vox(3) * fd[3] * 4});
  }
  GemmArgs g;
  g.a = gath; g.lda = 64; g.w = &c->fr_conv0; g.out = x[0]; g.ldc = fd[0];
  RET_IF(run_conv3d(c, g, TN, D0, S0, S0, 1, s));
  const int FT = c->film_total;
  RET_IF(film(pre));
  // down path: conv{1,3,5} stride 2, conv{2,4,6} stride 1
  for (int l = 0; l < 3; ++l) {
    const FrustumBlockW& b1 = c->fr_blocks[2 * l];
    const FrustumBlockW& b2 = c->fr_blocks[2 * l + 1];
    RET_IF(run_group_norm(c, x[l], fd[l], TN, D[l] * S[l] * S[l], b1.gn, 8, 1e-5f, ACT_SILU, pre + c->film_off[2 * l], a,
                          fd[l], s, FT));
    g = GemmArgs();
    g.a = a; g.lda = fd[l]; g.w = &b1.conv; g.out = tmp; g.ldc = fd[l + 1];
    RET_IF(run_conv3d(c, g, TN, D[l], S[l], S[l], 2, s));
    RET_IF(run_group_norm(c, tmp, fd[l + 1], TN, D[l + 1] * S[l + 1] * S[l + 1], b2.gn, 8, 1e-5f, ACT_SILU,
                          pre + c->film_off[2 * l + 1], a, fd[l + 1], s, FT));
    g = GemmArgs();
    g.a = a; g.lda = fd[l + 1]; g.w = &b2.conv; g.out = x[l + 1]; g.ldc = fd[l + 1];
    RET_IF(run_conv3d(c, g, TN, D[l + 1], S[l + 1], S[l + 1], 1, s));
  }
  // up path: x_l = up(x_{l+1}) + x_l  (in place on x_l through the residual epilogue)
  for (int l = 2; l >= 0; --l) {
    const FrustumBlockW& u = c->fr_up[2 - l];
    RET_IF(run_group_norm(c, x[l + 1], fd[l + 1], TN, D[l + 1] * S[l + 1] * S[l + 1], u.gn, 8, 1e-5f, ACT_SILU,
                          pre + c->film_off[6 + (2 - l)], a, fd[l + 1], s, FT));
    g = GemmArgs();
    g.a = a; g.lda = fd[l + 1]; g.w = &u.conv; g.out = x[l]; g.ldc = fd[l]; g.resid = x[l]; g.ldr = fd[l];
    if (l == 0 && x0h) {
      g.out = x0h;
      g.out_f32 = 0;
    }
    RET_IF(run_convT3d(c, g, TN, D[l + 1], S[l + 1], S[l + 1], s));
  }
  return 0;
}

int engine_frustum(mvd_ctx* c, const float* t_embed, const float* v_embed, const int32_t* view_idx_dev, int TN,
                   FrustumOut* out, hipStream_t s, bool half0) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  if (!c->volume || !c->cams) return mvd_fail("volume / cameras not set");
  // the volume may have been produced on another stream (mvd_set_volume_ready_event): its first reader waits here
  if (c->vol_ready) HIP_CHECK_RET(hipStreamWaitEvent(s, c->vol_ready, 0));
  const int D0 = c->v.frustum_volume_depth, S0 = c->v.input_image_size / 8, td = c->v.time_dim, vd = c->v.view_dim;
  return frustum_net(c, TN, out, s, half0, [&](half_t* gath) -> int {
    return launch_frustum_gather(c->volume, c->cams, view_idx_dev, TN, D0, S0, c->v.spatial_volume_size,
                                 c->v.spatial_volume_length, c->v.projection == 0, gath, s);
  }, [&](float* pre) -> int {
    // x + t_conv(t) + v_conv(v) (network.py:294,308) of all nine blocks in two launches: a per-(view, channel)
    // constant that the GroupNorm kernels fold in (pre[v][film_off[i] + c])
    const int FT = c->film_total;
    RET_IF(launch_small_linear(t_embed, td, -TN, td, c->film_t.w, c->film_t.bias, FT, ACT_NONE, pre, FT, 0, s));
    return launch_small_linear(v_embed, vd, TN, vd, c->film_v.w, c->film_v.bias, FT, ACT_NONE, pre, FT, 1, s);
  });
}

int engine_select_sample(mvd_ctx* c, int slot);
int engine_frustum_multi(mvd_ctx* c, int B, const int* slots, const float* t_embed, const float* v_embed,
                         const int32_t* view_idx_dev, int TN, FrustumOut* out, hipStream_t s, bool half0) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  if (c->vol_ready) HIP_CHECK_RET(hipStreamWaitEvent(s, c->vol_ready, 0));
  const int D0 = c->v.frustum_volume_depth, S0 = c->v.input_image_size / 8, td = c->v.time_dim, vd = c->v.view_dim;
  const int back = c->cur_slot;
  const int r = frustum_net(c, B * TN, out, s, half0, [&](half_t* gath) -> int {
    const size_t per = (size_t)TN * D0 * S0 * S0 * 64;
    for (int b = 0; b < B; ++b) {
      RET_IF(engine_select_sample(c, slots[b]));
      if (!c->cams || !c->volume) return mvd_fail("mvd_denoise_views_batch: a slot has no cameras / no volume");
      RET_IF(launch_frustum_gather(c->volume, c->cams, view_idx_dev, TN, D0, S0, c->v.spatial_volume_size,
                                   c->v.spatial_volume_length, c->v.projection == 0, gath + (size_t)b * per, s));
    }
    return 0;
  }, [&](float* pre) -> int {
    const int FT = c->film_total;
    for (int b = 0; b < B; ++b) {  // the step embedding is shared by the views of a sample
      float* pb = pre + (size_t)b * TN * FT;
      RET_IF(launch_small_linear(t_embed + (size_t)b * td, td, -TN, td, c->film_t.w, c->film_t.bias, FT, ACT_NONE, pb, FT, 0, s));
      RET_IF(launch_small_linear(v_embed + (size_t)b * TN * vd, vd, TN, vd, c->film_v.w, c->film_v.bias, FT, ACT_NONE, pb, FT, 1, s));
    }
    return 0;
  });
  const int r2 = engine_select_sample(c, back);
  return r ? r : r2;
}
// One target view for each of B samples (training_step: morphable_diffusion.py:496-518 with TN = 1): every sample's frustum is
// gathered from ITS 32^3 volume with ITS cameras (slots[b]), then the network runs once with the B volumes as its batch.
// volumes: channels-last [B][V^3][64]; t_embed [B][time_dim]; v_rows [B][view_dim] (the target view's embedding); view_idx_dev [B]
int engine_frustum_batch(mvd_ctx* c, int B, const int* slots, const float* volumes, const float* t_embed, const float* v_rows,
                         const int32_t* view_idx_dev, FrustumOut* out, hipStream_t s) {
  if (!c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  const int D0 = c->v.frustum_volume_depth, S0 = c->v.input_image_size / 8, td = c->v.time_dim, vd = c->v.view_dim;
  const int V = c->v.spatial_volume_size;
  const int back = c->cur_slot;
  const int r = frustum_net(c, B, out, s, false, [&](half_t* gath) -> int {
    const size_t per = (size_t)D0 * S0 * S0 * 64;
    for (int b = 0; b < B; ++b) {
      RET_IF(engine_select_sample(c, slots[b]));
      if (!c->cams) return mvd_fail("mvd_frustum_volumes_batch: a slot has no cameras");
      RET_IF(launch_frustum_gather(volumes + (size_t)b * V * V * V * 64, c->cams, view_idx_dev + b, 1, D0, S0, V,
                                   c->v.spatial_volume_length, c->v.projection == 0, gath + (size_t)b * per, s));
    }
    return 0;
  }, [&](float* pre) -> int {
    const int FT = c->film_total;
    RET_IF(launch_small_linear(t_embed, td, B, td, c->film_t.w, c->film_t.bias, FT, ACT_NONE, pre, FT, 0, s));
    return launch_small_linear(v_rows, vd, B, vd, c->film_v.w, c->film_v.bias, FT, ACT_NONE, pre, FT, 1, s);
  });
  const int r2 = engine_select_sample(c, back);
  return r ? r : r2;
}
