// extern "C" surface of libmvd_hip.so (declared in include/mvd.h).
#include <array>
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "engine.h"
#include "rowchain.h"

int engine_set_cameras(mvd_ctx* c, const float* K, const float* RT, int N, hipStream_t s);
int engine_set_mesh(mvd_ctx* c, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds,
                    int Nv, hipStream_t s);
int engine_select_sample(mvd_ctx* c, int slot);
int engine_set_samples(mvd_ctx* c, int B, const int* slots, const float* const* vertices, const int32_t* const* coord,
                       const int32_t* const* out_sh, const float* const* bounds, const int* Nv, const float* const* K,
                       const float* const* RT, int N, hipStream_t s);
int engine_rulebook_build(const int32_t* coord, const int32_t* out_sh, int Nv, int force_hash, int32_t* n_sites, int64_t* lens);
int engine_rulebook_table(int which, int32_t* out);
void mesh_free(MeshTables& m);

static thread_local std::string g_err;
int mvd_fail(const char* msg) {
  g_err = msg ? msg : "unknown error";
  return -1;
}
const char* mvd_error_text() { return g_err.c_str(); }

namespace {

__global__ void split_rows_to_f32_kernel(const half_t* in, int rows, int C, float* out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)rows * C; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / C, c = i % C;
    const half_t hi = in[r * 3 * C + c], lo = in[r * 3 * C + C + c], h2 = in[r * 3 * C + 2 * C + c];
    out[i] = (float)hi == (float)h2 ? (float)hi + (float)lo : __builtin_nanf("");
  }
}
__global__ void f16_to_f32_kernel(const half_t* in, float* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (float)in[i];
}
// q,k,v [rows][C] fp32 -> fp16 [rows][3C] (the layout of a fused q|k|v projection)
__global__ void pack_qkv_rows_kernel(const float* q, const float* k, const float* v, int rows, int C, half_t* qkv) {
  const long total = (long)rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i % C);
    qkv[(long)r * 3 * C + c] = (half_t)q[i];
    qkv[(long)r * 3 * C + C + c] = (half_t)k[i];
    qkv[(long)r * 3 * C + 2 * C + c] = (half_t)v[i];
  }
}
// UNetWrapper.predict_with_unconditional_scale input assembly (morphable_diffusion.py:133-146), channels-last:
// rows [0,TN) = [x | x_input / 0.18215], rows [TN,2TN) = [x | 0]
// UNet input of one sample's TN views: rows [0, TN) of `out` = the conditional copies, rows [half_off, half_off + TN) the
// unconditional ones (copies == 2); half_off = TN for a single sample, B * TN when B samples share the UNet pass
__global__ void build_cfg_input_kernel(const float* x_noisy, const float* x_input, int TN, int HW, int copies, float* out,
                                       long half_off) {
  const long total = (long)copies * TN * HW * 8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i & 7);
    const long bp = i >> 3;
    const int p = (int)(bp % HW), b = (int)(bp / HW);
    const int v = b % TN, half = b / TN;
    float val;
    if (ch < 4) val = x_noisy[((long)v * 4 + ch) * HW + p];
    else val = half == 0 ? x_input[(long)(ch - 4) * HW + p] / 0.18215f : 0.f;
    out[(((long)half * half_off + v) * HW + p) * 8 + ch] = val;
  }
}
__global__ void build_cfg_context_kernel(const float* clip, int TN, int dim, int copies, float* ctx, int64_t* t, int64_t step,
                                         long half_off) {
  const int total = copies * TN * dim;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int b = i / dim, j = i - b * dim;
    const long row = (long)(b / TN) * half_off + (b % TN);
    ctx[row * dim + j] = b < TN ? clip[j] : 0.f;
    if (j == 0) t[row] = step;
  }
}
__global__ void fill_pattern_f16_kernel(half_t* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (half_t)(((float)(h & 0xFFFF) / 32768.0f) - 1.0f);
  }
}

inline hipStream_t S(void* s) { return (hipStream_t)s; }
inline int nblk(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

}  // namespace

int bwd_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, float wd, int step,
              float inv_scale, const int* skip, hipStream_t s);
int bwd_finite_check(const float* g, size_t n, int* flag, hipStream_t s);

int bwd_attention(const half_t* qkv, int ld3, const half_t* o, const half_t* dO, int ldo, half_t* dqkv, int ldd, float* lse, float* delta,
                  int B, int T, int heads, int d, hipStream_t s);
int bwd_group_norm(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                   const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* dg_part,
                   float* db_part, float* dpre_part, hipStream_t s);
int bwd_sum_rows_add(const float* part, int R, int C, long ldp, float* out, int accum, hipStream_t s);
int bwd_gn_slabs(int B, int G, int rows);
int bwd_group_norm_slab(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                        const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* part,
                        float* part2, float* dg_part, float* db_part, float* dpre_part, int S, hipStream_t s);
int bwd_ln_max_blocks();
int bwd_layer_norm(const float* x, long ld, const float* dy, long ldy, int rows, int C, const float* gamma, float eps, float* dx,
                   long lddx, int accum, float* part, int* nblk, hipStream_t s);
int bwd_cast_rows(const float* src, long ld, long rows, int C, int Cp, half_t* dst, hipStream_t s, int split = 0);
extern "C" {

const char* mvd_last_error(void) { return g_err.c_str(); }
const char* mvd_compute_dtype(void) { return MVD_DTYPE_NAME; }

int mvd_create(const mvd_unet_config* ucfg, const mvd_volume_config* vcfg, int device, size_t workspace_bytes,
               mvd_ctx** out) {
  if (!ucfg || !vcfg || !out) return mvd_fail("mvd_create: null argument");
  HIP_CHECK_RET(hipSetDevice(device));
  mvd_ctx* c = new mvd_ctx();
  c->u = *ucfg;
  c->v = *vcfg;
  c->device = device;
  c->use_halo = getenv("MVD_NO_HALO") == nullptr;
  if (workspace_bytes == 0) workspace_bytes = (size_t)8 << 30;
  hipError_t e = hipMalloc((void**)&c->ws.base, workspace_bytes);
  if (e != hipSuccess) {
    delete c;
    return mvd_fail("mvd_create: workspace allocation failed");
  }
  c->ws.size = workspace_bytes;
  *out = c;
  return 0;
}

void mvd_destroy(mvd_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  mvd_comm_destroy(c);
  for (auto& kv : c->raw)
    if (!c->param_index.count(kv.first)) hipFree(kv.second.d);  // master parameters live in the arena
  float* arenas[4] = {c->arena_p, c->arena_g, c->arena_m, c->arena_v};
  for (int i = 0; i < 4; ++i)
    if (arenas[i] && c->arena_owned[i]) hipFree(arenas[i]);
  hipFree(c->found_inf);
  for (void* p : c->owned) hipFree(p);
  mesh_free(c->mesh);
  hipFree(c->cams);
  auto free_stage = [](mvd_ctx::CamStage& st) {
    if (st.h) hipHostFree(st.h);
    if (st.staged) hipEventDestroy(st.staged);
  };
  free_stage(c->cam_stage);
  for (auto& sl : c->slots) {
    mesh_free(sl.mesh);
    hipFree(sl.cams);
    free_stage(sl.cam_stage);
    hipFree(sl.volume);
  }
  hipFree(c->volume);
  hipFree(c->ws.base);
  for (hipEvent_t ev : c->probe_ev) hipEventDestroy(ev);
  for (auto& gb : c->buckets)
    if (gb.ev) hipEventDestroy(gb.ev);
  for (int i = 0; i < 4; ++i) {
    if (c->bstreams[i]) hipStreamDestroy(c->bstreams[i]);
    if (c->bevents[i]) hipEventDestroy(c->bevents[i]);
  }
  if (c->bev_after) hipEventDestroy(c->bev_after);
  for (hipEvent_t ev : c->ev_grad_sync)
    if (ev) hipEventDestroy(ev);
  for (auto& cc : c->cond_const)
    if (cc.k) hipFree(cc.k);
  if (c->enc_scratch) hipFree(c->enc_scratch);
  for (hipEvent_t ev : c->ev_cond) hipEventDestroy(ev);
  for (hipEvent_t ev : {c->ev_fork, c->ev_join, c->ev_join2, c->ev_ctx, c->ev_emb0, c->ev_emb})
    if (ev) hipEventDestroy(ev);
  if (c->side) hipStreamDestroy(c->side);
  if (c->side2) hipStreamDestroy(c->side2);
  for (hipEvent_t ev : {c->ev_s2_fork, c->ev_s2_join})
    if (ev) hipEventDestroy(ev);
  delete c;
}

int mvd_upload_weight(mvd_ctx* c, const char* name, const float* data, const int64_t* shape, int ndim, int on_device) {
  if (!c || !name || !data) return mvd_fail("mvd_upload_weight: null argument");
  if (c->finalized) return mvd_fail("mvd_upload_weight: weights already finalized");
  const std::string k(name);
  if (k.rfind("model.diffusion_model.", 0) != 0 && k.rfind("spatial_volume.", 0) != 0 && k.rfind("time_embed.", 0) != 0 &&
      k.rfind("first_stage_model.decoder.", 0) != 0 && k.rfind("first_stage_model.post_quant_conv.", 0) != 0 &&
      k.rfind("first_stage_model.encoder.", 0) != 0 && k.rfind("first_stage_model.quant_conv.", 0) != 0 &&
      k.rfind("clip_image_encoder.model.visual.", 0) != 0)
    return 0;  // CLIP text tower / loss / schedule buffers: not on this path
  HIP_CHECK_RET(hipSetDevice(c->device));
  RawTensor t;
  t.numel = 1;
  for (int i = 0; i < ndim; ++i) {
    t.shape.push_back(shape[i]);
    t.numel *= (size_t)shape[i];
  }
  HIP_CHECK_RET(hipMalloc((void**)&t.d, std::max<size_t>(t.numel, 1) * sizeof(float)));
  HIP_CHECK_RET(hipMemcpy(t.d, data, t.numel * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  auto it = c->raw.find(k);
  if (it != c->raw.end()) hipFree(it->second.d);
  c->raw[k] = t;
  return 0;
}

int mvd_set_precision_level(mvd_ctx* c, int level) {
  if (!c) return mvd_fail("null context");
  if (c->finalized) return mvd_fail("mvd_set_precision_level: weights already finalized");
  if (level < 0 || level > 6) return mvd_fail("mvd_set_precision_level: level must be 0..6");
  c->precision_level = level;
  return 0;
}

int mvd_set_vae_precision(mvd_ctx* c, int exact) {
  if (!c) return mvd_fail("null context");
  if (c->finalized) return mvd_fail("mvd_set_vae_precision: weights already finalized");
  c->vae_exact = exact != 0;
  return 0;
}

int mvd_finalize_weights(mvd_ctx* c) {
  if (!c) return mvd_fail("null context");
  if (c->finalized) return 0;
  return engine_finalize(c);
}

int mvd_set_mesh(mvd_ctx* c, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds, int Nv) {
  if (!c || !vertices || !coord || !out_sh || !bounds || Nv <= 0) return mvd_fail("mvd_set_mesh: bad argument");
  HIP_CHECK_RET(hipSetDevice(c->device));
  HIP_CHECK_RET(hipDeviceSynchronize());  // launches on ANY stream may still read the tables being replaced
  RET_IF(engine_set_mesh(c, vertices, coord, out_sh, bounds, Nv, nullptr));
  HIP_CHECK_RET(hipStreamSynchronize(nullptr));
  return 0;
}

int mvd_set_mesh_async(mvd_ctx* c, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds, int Nv,
                       void* stream) {
  if (!c || !vertices || !coord || !out_sh || !bounds || Nv <= 0) return mvd_fail("mvd_set_mesh_async: bad argument");
  HIP_CHECK_RET(hipSetDevice(c->device));
  return engine_set_mesh(c, vertices, coord, out_sh, bounds, Nv, S(stream));
}

int mvd_select_sample(mvd_ctx* c, int slot) {
  if (!c) return mvd_fail("null context");
  HIP_CHECK_RET(hipSetDevice(c->device));
  return engine_select_sample(c, slot);
}

int mvd_set_cameras(mvd_ctx* c, const float* K, const float* RT, int N) {
  if (!c || !K || !RT || N <= 0) return mvd_fail("mvd_set_cameras: bad argument");
  HIP_CHECK_RET(hipSetDevice(c->device));
  HIP_CHECK_RET(hipDeviceSynchronize());
  RET_IF(engine_set_cameras(c, K, RT, N, nullptr));
  HIP_CHECK_RET(hipStreamSynchronize(nullptr));
  return 0;
}

int mvd_set_samples_async(mvd_ctx* c, int B, const int* slots, const float* const* vertices, const int32_t* const* coord,
                          const int32_t* const* out_sh, const float* const* bounds, const int* Nv, const float* const* K,
                          const float* const* RT, int N, void* stream) {
  if (!c || B <= 0 || !slots || !vertices || !coord || !out_sh || !bounds || !Nv || !K || !RT || N <= 0)
    return mvd_fail("mvd_set_samples_async: bad argument");
  for (int i = 0; i < B; ++i) {
    if (!vertices[i] || !coord[i] || !out_sh[i] || !bounds[i] || !K[i] || !RT[i] || Nv[i] <= 0 || slots[i] < 0 || slots[i] >= 64)
      return mvd_fail("mvd_set_samples_async: bad argument");
    for (int j = 0; j < i; ++j)
      if (slots[j] == slots[i]) return mvd_fail("mvd_set_samples_async: a slot appears twice");
  }
  HIP_CHECK_RET(hipSetDevice(c->device));
  return engine_set_samples(c, B, slots, vertices, coord, out_sh, bounds, Nv, K, RT, N, S(stream));
}

int mvd_rulebook_build(const int32_t* coord, const int32_t* out_sh, int Nv, int force_hash, int32_t* n_sites, int64_t* lens) {
  if (!coord || !out_sh || Nv <= 0 || !n_sites || !lens) return mvd_fail("mvd_rulebook_build: bad argument");
  return engine_rulebook_build(coord, out_sh, Nv, force_hash, n_sites, lens);
}
int mvd_rulebook_table(int which, int32_t* out) {
  if (!out) return mvd_fail("mvd_rulebook_table: bad argument");
  return engine_rulebook_table(which, out);
}

int mvd_set_cameras_async(mvd_ctx* c, const float* K, const float* RT, int N, void* stream) {
  if (!c || !K || !RT || N <= 0) return mvd_fail("mvd_set_cameras_async: bad argument");
  HIP_CHECK_RET(hipSetDevice(c->device));
  return engine_set_cameras(c, K, RT, N, S(stream));
}

int mvd_embed_time(mvd_ctx* c, const int64_t* t, int B, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  WsScope ws_scope(c);
  const int td = c->v.time_dim;
  float* e0 = ws_alloc<float>(c, (size_t)B * td);
  float* e1 = ws_alloc<float>(c, (size_t)B * td);
  WS_CHECK(e0 && e1);
  RET_IF(launch_timestep_embedding(t, B, td, e0, S(stream)));
  RET_IF(launch_small_linear(e0, td, B, td, c->step_te0.w, c->step_te0.bias, td, ACT_NONE, e1, td, 0, S(stream)));
  RET_IF(launch_small_linear(e1, td, B, td, c->step_te2.w, c->step_te2.bias, td, ACT_SILU, out, td, 0, S(stream)));
  return 0;
}

int mvd_unet_forward(mvd_ctx* c, const float* x, const int64_t* timesteps, const float* context, int Bv, int n_ctx,
                     const float* src0, const float* src1, const float* src2, const float* src3, int depth0, float* out,
                     void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (n_ctx < 0 || n_ctx > Bv) return mvd_fail("mvd_unet_forward: n_ctx out of range");
  hipStream_t s = S(stream);
  const mvd_unet_config& u = c->u;
  WsScope ws_scope(c);
  const int HW = u.image_size * u.image_size;
  const int cin = u.in_channels;
  if (cin % 8) return mvd_fail("in_channels must be a multiple of 8");
  float* xn = ws_alloc<float>(c, (size_t)Bv * HW * cin);
  float* eps = ws_alloc<float>(c, (size_t)Bv * HW * u.out_channels);
  WS_CHECK(xn && eps);
  RET_IF(launch_nchw_to_nhwc(x, Bv, cin, HW, xn, cin, cin, s));
  const float* srcs[4] = {src0, src1, src2, src3};
  Ctx5 cl[4];
  for (int l = 0; l < 4 && n_ctx > 0; ++l) {
    if (!srcs[l]) return mvd_fail("mvd_unet_forward: missing source_dict level");
    const int sl = u.image_size >> l, Dl = depth0 >> l, C = u.volume_dims[l];
    float* t = ws_alloc<float>(c, (size_t)n_ctx * Dl * sl * sl * C);
    WS_CHECK(t);
    RET_IF(launch_nchw_to_nhwc(srcs[l], n_ctx, C, Dl * sl * sl, t, C, C, s));
    cl[l].p = t;
    cl[l].f32 = 1;
  }
  RET_IF(engine_unet(c, xn, cin, timesteps, context, Bv, n_ctx, depth0, cl, eps, s));
  RET_IF(launch_nhwc_to_nchw(eps, u.out_channels, Bv, u.out_channels, HW, out, s));
  return 0;
}

int mvd_unet_block(mvd_ctx* c, const char* path, const float* x, int B, int C, int H, int W, const int64_t* timesteps,
                   const float* context, const float* volume, int D, float* out, int out_capacity, int* out_shape, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized || !path || !x || !out || !out_shape || B <= 0 || C <= 0 || (C & 7) || H <= 0 || W <= 0)
    return mvd_fail("mvd_unet_block: bad argument");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int HW = H * W;
  float* xn = ws_alloc<float>(c, (size_t)B * HW * C);
  // any block's output fits: an Upsample quadruples the pixels, no block is wider than 8 * model_channels
  float* on = ws_alloc<float>(c, (size_t)B * HW * 4 * (size_t)(c->u.model_channels * 8));
  WS_CHECK(xn && on);
  RET_IF(launch_nchw_to_nhwc(x, B, C, HW, xn, C, C, s));
  float* vn = nullptr;
  if (volume) {
    int level = 0;
    for (int r = c->u.image_size; r > H; r >>= 1) ++level;
    if (level > 3 || D <= 0) return mvd_fail("mvd_unet_block: bad volume");
    const int Cc = c->u.volume_dims[level];
    vn = ws_alloc<float>(c, (size_t)B * D * HW * Cc);
    WS_CHECK(vn);
    RET_IF(launch_nchw_to_nhwc(volume, B, Cc, D * HW, vn, Cc, Cc, s));
  }
  int Co = 0, Ho = 0;
  RET_IF(engine_unet_block(c, path, xn, B, C, H, W, timesteps, context, vn, D, on, &Co, &Ho, s));
  if ((size_t)B * Co * Ho * Ho > (size_t)out_capacity) return mvd_fail("mvd_unet_block: output buffer too small");
  RET_IF(launch_nhwc_to_nchw(on, Co, B, Co, Ho * Ho, out, s));
  out_shape[0] = B; out_shape[1] = Co; out_shape[2] = Ho; out_shape[3] = Ho;
  return 0;
}

int mvd_vertex_features(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed,
                        const int32_t* view_idx, int n_local, int add_bias, float* fused_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  return engine_vertex_features(c, x_noisy, t_embed, v_embed, view_idx, n_local, add_bias, fused_out, S(stream));
}

int mvd_vertex_view_features(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed,
                             const int32_t* view_idx, int n_local, float* vf_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !vf_out) return mvd_fail("mvd_vertex_view_features: null argument");
  return engine_vertex_features(c, x_noisy, t_embed, v_embed, view_idx, n_local, 0, nullptr, S(stream), vf_out);
}

int mvd_vertex_features_stream_safe(mvd_ctx* c) {
  return (c && c->finalized && c->has_cond && engine_encoder_is_fused(c)) ? 1 : 0;
}

int mvd_fuse_vertex_features(mvd_ctx* c, const float* vf_all, int n_views, float* fused_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !vf_all || !fused_out) return mvd_fail("mvd_fuse_vertex_features: null argument");
  return engine_fuse_vertex_features(c, vf_all, n_views, fused_out, S(stream));
}

int mvd_stage_target_encoder(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed, int n_local,
                             float* feats_nchw, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !x_noisy || !t_embed || !v_embed || !feats_nchw || n_local <= 0) return mvd_fail("mvd_stage_target_encoder: bad argument");
  WsScope ws_scope(c);
  const int HW = c->u.image_size * c->u.image_size;
  float* feats = ws_alloc<float>(c, (size_t)n_local * HW * 16);
  WS_CHECK(feats);
  RET_IF(engine_target_encoder(c, x_noisy, t_embed, v_embed, n_local, feats, S(stream)));
  return launch_nhwc_to_nchw(feats, 16, n_local, 16, HW, feats_nchw, S(stream));
}

int mvd_stage_sparse_dense(mvd_ctx* c, const float* fused, int train_mode, float* dense_out, int32_t* shape_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized || !c->has_cond) return mvd_fail("spatial_volume weights not uploaded / finalized");
  if (!c->mesh.Nv) return mvd_fail("mvd_set_mesh must be called first");
  const MeshTables& m = c->mesh;
  if (shape_out) {
    shape_out[0] = c->sparse[8].cout;
    for (int a = 0; a < 3; ++a) shape_out[1 + a] = m.shape[2][a];
  }
  if (!dense_out) return 0;  // shape query
  if (!fused) return mvd_fail("mvd_stage_sparse_dense: null argument");
  const float* rows = nullptr;
  RET_IF(engine_sparse_net(c, fused, S(stream), train_mode != 0, &rows));
  return launch_sparse_densify(rows, m.grid2, (long)m.shape[2][0] * m.shape[2][1] * m.shape[2][2], c->sparse[8].cout, dense_out, S(stream));
}

int mvd_set_volume_ready_event(mvd_ctx* c, void* event) {
  if (!c) return mvd_fail("null context");
  c->vol_ready = (hipEvent_t)event;
  return 0;
}

int mvd_volume_from_fused(mvd_ctx* c, const float* fused, float* volume_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  c->vol_ready = nullptr;  // a registered hand-over event guarded the PREVIOUS volume (the caller registers a new one after this call)
  RET_IF(engine_volume_from_fused(c, fused, S(stream)));
  if (volume_out) {
    const int V = c->v.spatial_volume_size;
    RET_IF(launch_nhwc_to_nchw(c->volume, 64, 1, 64, V * V * V, volume_out, S(stream)));
  }
  return 0;
}

int mvd_volume_from_fused_train(mvd_ctx* c, const float* fused, float* volume_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  c->vol_ready = nullptr;
  RET_IF(engine_volume_from_fused(c, fused, S(stream), /*bn_batch_stats=*/true));
  if (volume_out) {
    const int V = c->v.spatial_volume_size;
    RET_IF(launch_nhwc_to_nchw(c->volume, 64, 1, 64, V * V * V, volume_out, S(stream)));
  }
  return 0;
}

int mvd_set_volume(mvd_ctx* c, const float* volume, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !volume) return mvd_fail("mvd_set_volume: null argument");
  if (!c->volume) return mvd_fail("mvd_set_volume: mvd_set_mesh must be called first");
  const int V = c->v.spatial_volume_size;
  c->vol_ready = nullptr;
  return launch_nchw_to_nhwc(volume, 1, 64, V * V * V, c->volume, 64, 64, S(stream));
}

int mvd_train_enable(mvd_ctx* c, int on) {
  if (!c) return mvd_fail("null context");
  if (c->finalized) return mvd_fail("mvd_train_enable: call it before mvd_finalize_weights");
  c->train_mode = on != 0;
  return 0;
}

int mvd_train_param_count(mvd_ctx* c) { return (c && c->train_mode && c->finalized) ? (int)c->params.size() : 0; }

int mvd_train_param_info(mvd_ctx* c, int i, char* name, size_t cap, int64_t* offset, int64_t* numel, int64_t* shape, int* ndim) {
  if (!c || !c->train_mode || !c->finalized) return mvd_fail("mvd_train_param_info: context not finalized in training mode");
  if (i < 0 || i >= (int)c->params.size()) return mvd_fail("mvd_train_param_info: index out of range");
  const mvd_ctx::ParamRec& r = c->params[i];
  if (name) {
    if (r.key.size() + 1 > cap) return mvd_fail("mvd_train_param_info: name buffer too small");
    memcpy(name, r.key.c_str(), r.key.size() + 1);
  }
  if (offset) *offset = (int64_t)r.off;
  if (numel) *numel = (int64_t)r.numel;
  if (ndim) *ndim = (int)r.shape.size();
  if (shape)
    for (size_t k = 0; k < r.shape.size() && k < 8; ++k) shape[k] = r.shape[k];
  return 0;
}

int64_t mvd_train_arena_size(mvd_ctx* c) { return (c && c->train_mode && c->finalized) ? (int64_t)c->arena_n : 0; }

int mvd_train_adopt_arena(mvd_ctx* c, int which, float* ptr, int64_t numel) {
  if (!c || !c->train_mode || !c->finalized) return mvd_fail("mvd_train_adopt_arena: context not finalized in training mode");
  if (which < 0 || which > 3 || !ptr || numel != (int64_t)c->arena_n) return mvd_fail("mvd_train_adopt_arena: bad argument");
  HIP_CHECK_RET(hipSetDevice(c->device));
  HIP_CHECK_RET(hipDeviceSynchronize());
  float** slot[4] = {&c->arena_p, &c->arena_g, &c->arena_m, &c->arena_v};
  float* old = *slot[which];
  if (old == ptr) return 0;
  if (old) HIP_CHECK_RET(hipMemcpy(ptr, old, c->arena_n * sizeof(float), hipMemcpyDeviceToDevice));
  else HIP_CHECK_RET(hipMemset(ptr, 0, c->arena_n * sizeof(float)));
  if (old && c->arena_owned[which]) hipFree(old);
  *slot[which] = ptr;
  c->arena_owned[which] = false;
  if (which == 0) {
    for (auto& r : c->params) c->raw[r.key].d = ptr + r.off;
    // biases / norm gains are read in place from the master arena (engine_weights.hip: copy_f32): every such pointer is refreshed
    RET_IF(engine_repack(c));
  }
  return 0;
}

int mvd_train_zero_grad(mvd_ctx* c, void* stream) {
  if (!c || !c->train_mode || !c->finalized) return mvd_fail("mvd_train_zero_grad: context not finalized in training mode");
  HIP_CHECK_RET(hipSetDevice(c->device));
  HIP_CHECK_RET(hipMemsetAsync(c->arena_g, 0, c->arena_n * sizeof(float), S(stream)));
  c->grad_touched[1] = c->grad_touched[2] = false;
  return 0;
}

int mvd_train_unet_step(mvd_ctx* c, const float* x, const int64_t* timesteps, const float* context, int B, const float* src0,
                        const float* src1, const float* src2, const float* src3, int depth0, const float* target, float loss_scale,
                        int recompute, float* pred_out, float* loss_out, float* dsrc0, float* dsrc1, float* dsrc2, float* dsrc3,
                        void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (!x || !timesteps || !context || !target || !pred_out || B <= 0) return mvd_fail("mvd_train_unet_step: bad argument");
  hipStream_t s = S(stream);
  const mvd_unet_config& u = c->u;
  WsScope ws_scope(c);
  const int HW = u.image_size * u.image_size, cin = u.in_channels, oc = u.out_channels;
  if (cin % 8) return mvd_fail("in_channels must be a multiple of 8");
  float* xn = ws_alloc<float>(c, (size_t)B * HW * cin);
  float* eps = ws_alloc<float>(c, (size_t)B * HW * oc);
  float* tgt = ws_alloc<float>(c, (size_t)B * HW * oc);
  WS_CHECK(xn && eps && tgt);
  RET_IF(launch_nchw_to_nhwc(x, B, cin, HW, xn, cin, cin, s));
  RET_IF(launch_nchw_to_nhwc(target, B, oc, HW, tgt, oc, oc, s));
  const float* srcs[4] = {src0, src1, src2, src3};
  float* douts[4] = {dsrc0, dsrc1, dsrc2, dsrc3};
  Ctx5 cl[4];
  float* dcl[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t nvol[4];
  for (int l = 0; l < 4; ++l) {
    if (!srcs[l]) return mvd_fail("mvd_train_unet_step: missing source_dict level");
    const int sl = u.image_size >> l, Dl = depth0 >> l, C = u.volume_dims[l];
    nvol[l] = (size_t)B * Dl * sl * sl * C;
    float* t = ws_alloc<float>(c, nvol[l]);
    WS_CHECK(t);
    RET_IF(launch_nchw_to_nhwc(srcs[l], B, C, Dl * sl * sl, t, C, C, s));
    cl[l].p = t;
    cl[l].f32 = 1;
    if (douts[l]) {
      dcl[l] = ws_alloc<float>(c, nvol[l]);
      WS_CHECK(dcl[l]);
      HIP_CHECK_RET(hipMemsetAsync(dcl[l], 0, nvol[l] * sizeof(float), s));
    }
  }
  RET_IF(engine_train_step(c, xn, cin, timesteps, context, B, depth0, cl, tgt, loss_scale, recompute, eps, loss_out, dcl, s));
  c->grad_touched[1] = true;
  RET_IF(launch_nhwc_to_nchw(eps, oc, B, oc, HW, pred_out, s));
  for (int l = 0; l < 4; ++l)
    if (douts[l]) {
      const int sl = u.image_size >> l, Dl = depth0 >> l, C = u.volume_dims[l];
      RET_IF(launch_nhwc_to_nchw(dcl[l], C, B, C, Dl * sl * sl, douts[l], s));
    }
  return 0;
}

int mvd_train_cond_backward(mvd_ctx* c, int cond_index, const float* x, const float* context, const float* d_out, int B, int H,
                            int W, int depth0, float* dx, float* dcontext, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized || !c->train_mode) return mvd_fail("mvd_train_cond_backward: context not finalized in training mode");
  if (cond_index < 0 || cond_index >= (int)c->conds.size() || !x || !context || !d_out || !dx || B <= 0)
    return mvd_fail("mvd_train_cond_backward: bad argument");
  hipStream_t s = S(stream);
  const CondW& cd = c->conds[cond_index];
  int level = 0;
  for (int r = c->u.image_size; r > H; r >>= 1) ++level;
  if (level > 3 || (c->u.image_size >> level) != H || H != W) return mvd_fail("mvd_train_cond_backward: resolution is not a UNet level");
  const int D = depth0 >> level, HW = H * W;
  WsScope ws_scope(c);
  float* xn = ws_alloc<float>(c, (size_t)B * HW * cd.dim);
  float* dn = ws_alloc<float>(c, (size_t)B * HW * cd.dim);
  float* gx = ws_alloc<float>(c, (size_t)B * HW * cd.dim);
  float* cn = ws_alloc<float>(c, (size_t)B * D * HW * cd.Cc);
  float* gc = ws_alloc<float>(c, (size_t)B * D * HW * cd.Cc);
  WS_CHECK(xn && dn && gx && cn && gc);
  RET_IF(launch_nchw_to_nhwc(x, B, cd.dim, HW, xn, cd.dim, cd.dim, s));
  RET_IF(launch_nchw_to_nhwc(d_out, B, cd.dim, HW, dn, cd.dim, cd.dim, s));
  RET_IF(launch_nchw_to_nhwc(context, B, cd.Cc, D * HW, cn, cd.Cc, cd.Cc, s));
  HIP_CHECK_RET(hipMemsetAsync(gc, 0, (size_t)B * D * HW * cd.Cc * sizeof(float), s));
  RET_IF(engine_train_cond_backward(c, cond_index, xn, cn, dn, B, H, W, level, depth0, gx, gc, s));
  c->grad_touched[1] = true;
  RET_IF(launch_nhwc_to_nchw(gx, cd.dim, B, cd.dim, HW, dx, s));
  if (dcontext) RET_IF(launch_nhwc_to_nchw(gc, cd.Cc, B, cd.Cc, D * HW, dcontext, s));
  return 0;
}

int mvd_train_conditioner_backward_batch(mvd_ctx* c, int B, const int* slots, const float* x_noisy, const int64_t* timesteps,
                                         const float* v_embed, int n_views, const int* target_index, const float* dsrc0,
                                         const float* dsrc1, const float* dsrc2, const float* dsrc3, float* dbg_dvolume,
                                         float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized || !c->train_mode) return mvd_fail("mvd_train_conditioner_backward: context not finalized in training mode");
  if (B < 1 || !slots || !x_noisy || !timesteps || !v_embed || !target_index || !dsrc0 || !dsrc1 || !dsrc2 || !dsrc3)
    return mvd_fail("mvd_train_conditioner_backward: null argument");
  for (int i = 0; i < B; ++i)
    if (slots[i] < 0 || slots[i] >= 64) return mvd_fail("mvd_train_conditioner_backward: slot out of range (0..63)");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const float* srcs[4] = {dsrc0, dsrc1, dsrc2, dsrc3};
  float* dcl[4];
  int D = c->v.frustum_volume_depth, Sz = c->v.input_image_size / 8;
  for (int l = 0; l < 4; ++l) {  // NCDHW (the reference's layout) -> channels-last, all samples
    const int C = c->v.frustum_dims[l];
    const size_t n = (size_t)B * D * Sz * Sz * C;
    dcl[l] = ws_alloc<float>(c, n);
    WS_CHECK(dcl[l]);
    RET_IF(launch_nchw_to_nhwc(srcs[l], B, C, D * Sz * Sz, dcl[l], C, C, s));
    D = (D - 1) / 2 + 1;
    Sz = (Sz - 1) / 2 + 1;
  }
  c->grad_touched[2] = true;
  return engine_train_conditioner_backward_batch(c, B, slots, x_noisy, timesteps, v_embed, n_views, target_index, dcl, dbg_dvolume,
                                                 dbg_dfused, dbg_dfeats, dbg_dtembed, s);
}

int mvd_train_conditioner_backward(mvd_ctx* c, const float* x_noisy, int64_t timestep, const float* v_embed, int n_views,
                                   int target_index, const float* dsrc0, const float* dsrc1, const float* dsrc2, const float* dsrc3,
                                   float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, void* stream) {
  if (!c) return mvd_fail("mvd_train_conditioner_backward: null argument");
  const int slot = c->cur_slot;
  return mvd_train_conditioner_backward_batch(c, 1, &slot, x_noisy, &timestep, v_embed, n_views, &target_index, dsrc0, dsrc1, dsrc2, dsrc3,
                                              dbg_dvolume, dbg_dfused, dbg_dfeats, dbg_dtembed, stream);
}

int mvd_train_get_grad(mvd_ctx* c, const char* name, float* out, size_t numel, void* stream) {
  if (!c || !name || !out) return mvd_fail("mvd_train_get_grad: null argument");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  auto it = c->param_index.find(name);
  if (it == c->param_index.end()) return mvd_fail("mvd_train_get_grad: not a parameter of this context (training mode off?)");
  const mvd_ctx::ParamRec& r = c->params[it->second];
  if (r.numel != numel) return mvd_fail("mvd_train_get_grad: size mismatch");
  HIP_CHECK_RET(hipMemcpyAsync(out, c->arena_g + r.off, numel * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
  return 0;
}

int mvd_train_get_tensor(mvd_ctx* c, const char* name, float* out, size_t numel, void* stream) {
  if (!c || !name || !out) return mvd_fail("mvd_train_get_tensor: null argument");
  if (!c->train_mode || !c->finalized) return mvd_fail("mvd_train_get_tensor: context not finalized in training mode");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  auto it = c->raw.find(name);  // parameters (their storage is the master arena) and the resident buffers
  if (it == c->raw.end()) return mvd_fail("mvd_train_get_tensor: not a resident tensor of this context");
  if (it->second.numel != numel) return mvd_fail("mvd_train_get_tensor: size mismatch");
  HIP_CHECK_RET(hipMemcpyAsync(out, it->second.d, numel * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
  return 0;
}

int64_t mvd_train_bn_calls(mvd_ctx* c) { return c ? (int64_t)c->bn_train_calls : 0; }

int mvd_train_adamw_step(mvd_ctx* c, float lr, float lr_aux, float beta1, float beta2, float eps, float weight_decay, int step,
                         float inv_scale, int finetune_unet, int* skipped_out, void* stream) {
  if (!c || !c->train_mode || !c->finalized) return mvd_fail("mvd_train_adamw_step: context not finalized in training mode");
  if (step < 1) return mvd_fail("mvd_train_adamw_step: step counts from 1");
  HIP_CHECK_RET(hipSetDevice(c->device));
  hipStream_t s = S(stream);
  for (float** a : {&c->arena_m, &c->arena_v})
    if (!*a) {
      HIP_CHECK_RET(hipMalloc((void**)a, c->arena_n * sizeof(float)));
      HIP_CHECK_RET(hipMemset(*a, 0, c->arena_n * sizeof(float)));
      c->arena_owned[a == &c->arena_m ? 2 : 3] = true;
    }
  HIP_CHECK_RET(hipMemsetAsync(c->found_inf, 0, sizeof(int), s));
  // The overflow check (a 3.7 GB read: ~1 ms) belongs to loss scaling -- GradScaler's "skip the step, halve the scale".  Without
  // a loss scale (inv_scale == 1: the bfloat16 build) the step is torch.optim.AdamW's own, which checks nothing.
  if (inv_scale != 1.0f) RET_IF(bwd_finite_check(c->arena_g, c->arena_n, c->found_inf, s));
  // the reference's parameter groups (morphable_diffusion.py:627-646): the UNet (all of it with finetune_unet, else the
  // DepthTransformers: attention.py:140-142) at lr; time_embed and spatial_volume at 10 lr (passed in as lr_aux).
  // Consecutive parameters of one group are updated by one launch.
  const std::string U = "model.diffusion_model.";
  auto group_of = [&](const std::string& k) -> int {
    if (k.rfind(U, 0) == 0) {
      if (finetune_unet) return 1;
      return (k.rfind(U + "middle_conditions.", 0) == 0 || k.rfind(U + "output_conditions.", 0) == 0) ? 1 : 0;
    }
    return 2;
  };
  size_t i = 0;
  while (i < c->params.size()) {
    const int grp = group_of(c->params[i].key);
    size_t j = i;
    while (j + 1 < c->params.size() && group_of(c->params[j + 1].key) == grp) ++j;
    if (grp && c->grad_touched[grp]) {  // a group no backward pass wrote to since zero_grad keeps its parameters AND its moments
      const size_t off = c->params[i].off, end = c->params[j].off + ((c->params[j].numel + 63) & ~(size_t)63);
      RET_IF(bwd_adamw(c->arena_p + off, c->arena_g + off, c->arena_m + off, c->arena_v + off, end - off, grp == 1 ? lr : lr_aux, beta1,
                       beta2, eps, weight_decay, step, inv_scale, c->found_inf, s));
    }
    i = j + 1;
  }
  if (skipped_out) {
    HIP_CHECK_RET(hipMemcpyAsync(skipped_out, c->found_inf, sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK_RET(hipStreamSynchronize(s));
  }
  return 0;
}

int mvd_train_repack(mvd_ctx* c) {
  if (!c) return mvd_fail("null context");
  return engine_repack(c);
}
int mvd_train_repack_async(mvd_ctx* c, void* stream) {
  if (!c) return mvd_fail("null context");
  return engine_repack(c, S(stream), true);
}

int mvd_train_grad_bucket_count(mvd_ctx* c) { return c ? c->n_buckets : 0; }
int mvd_train_grad_bucket(mvd_ctx* c, int k, int max_ranges, int64_t* offs, int64_t* lens, int* n_ranges) {
  if (!c || k < 0 || k >= c->n_buckets || !n_ranges) return mvd_fail("mvd_train_grad_bucket: no such bucket");
  const mvd_ctx::GradBucket& b = c->buckets[k];
  *n_ranges = (int)b.off.size();
  if (!offs || !lens) return 0;  // count only
  if (max_ranges < (int)b.off.size()) return mvd_fail("mvd_train_grad_bucket: range arrays too small");
  for (size_t r = 0; r < b.off.size(); ++r) {
    offs[r] = (int64_t)b.off[r];
    lens[r] = (int64_t)b.len[r];
  }
  return 0;
}
int mvd_train_grad_bucket_wait(mvd_ctx* c, int k, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || k < 0 || k >= c->n_buckets || !c->buckets[k].ev) return mvd_fail("mvd_train_grad_bucket_wait: no such bucket");
  HIP_CHECK_RET(hipStreamWaitEvent(S(stream), c->buckets[k].ev, 0));
  return 0;
}
int mvd_train_set_bucket_snapshot(mvd_ctx* c, float* arena) {
  if (!c) return mvd_fail("null context");
  c->bucket_snapshot = arena;
  return 0;
}

int mvd_mse_loss(mvd_ctx* c, const float* a, const float* b, size_t n, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !a || !b || !out || n == 0) return mvd_fail("mvd_mse_loss: bad argument");
  return launch_mse(a, b, n, out, S(stream));
}

int mvd_frustum_volumes(mvd_ctx* c, const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN,
                        float* out0, float* out1, float* out2, float* out3, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  WsScope ws_scope(c);
  FrustumOut fo;
  RET_IF(engine_frustum(c, t_embed, v_embed, view_idx, TN, &fo, S(stream)));
  float* outs[4] = {out0, out1, out2, out3};
  int D = c->v.frustum_volume_depth, Sz = c->v.input_image_size / 8;
  for (int l = 0; l < 4; ++l) {
    if (outs[l]) RET_IF(launch_nhwc_to_nchw(fo.lvl[l], c->v.frustum_dims[l], TN, c->v.frustum_dims[l], D * Sz * Sz, outs[l], S(stream)));
    D = (D - 1) / 2 + 1;
    Sz = (Sz - 1) / 2 + 1;
  }
  return 0;
}

int mvd_frustum_volumes_batch(mvd_ctx* c, int B, const int* slots, const float* volumes, const float* t_embed, const float* v_rows,
                              const int32_t* view_idx, float* out0, float* out1, float* out2, float* out3, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (B < 1 || !slots || !volumes || !t_embed || !v_rows || !view_idx) return mvd_fail("mvd_frustum_volumes_batch: bad argument");
  for (int i = 0; i < B; ++i)
    if (slots[i] < 0 || slots[i] >= 64) return mvd_fail("mvd_frustum_volumes_batch: slot out of range (0..63)");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int V = c->v.spatial_volume_size;
  float* vcl = ws_alloc<float>(c, (size_t)B * V * V * V * 64);  // NCDHW -> channels-last, all samples
  WS_CHECK(vcl);
  RET_IF(launch_nchw_to_nhwc(volumes, B, 64, V * V * V, vcl, 64, 64, s));
  FrustumOut fo;
  RET_IF(engine_frustum_batch(c, B, slots, vcl, t_embed, v_rows, view_idx, &fo, s));
  float* outs[4] = {out0, out1, out2, out3};
  int D = c->v.frustum_volume_depth, Sz = c->v.input_image_size / 8;
  for (int l = 0; l < 4; ++l) {
    if (outs[l]) RET_IF(launch_nhwc_to_nchw(fo.lvl[l], c->v.frustum_dims[l], B, c->v.frustum_dims[l], D * Sz * Sz, outs[l], s));
    D = (D - 1) / 2 + 1;
    Sz = (Sz - 1) / 2 + 1;
  }
  return 0;
}

int mvd_denoise_views_batch(mvd_ctx* c, int B, const int* slots, const float* x_noisy, const float* x_input, const float* clip,
                            const int64_t* timesteps, const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN,
                            float cfg_scale, const float* noise, float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev,
                            float dir_coef, float sigma, float* eps_out, float* x_prev, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (B < 1 || TN < 1 || (B > 1 && !slots) || !timesteps) return mvd_fail("mvd_denoise_views_batch: bad argument");
  hipStream_t s = S(stream);
  const mvd_unet_config& u = c->u;
  if (u.in_channels != 8 || u.out_channels != 4) return mvd_fail("denoise_views: expects the 8-in / 4-out latent UNet");
  WsScope ws_scope(c);
  const int HW = u.image_size * u.image_size;
  const bool cfg = cfg_scale != 1.0f;
  const int copies = cfg ? 2 : 1, Bv = copies * B * TN;
  const int td = c->v.time_dim, vd = c->v.view_dim;
  // The frustum network feeds the DepthTransformers only (middle block onwards): engine_unet enqueues it (through this
  // producer) on its side stream after the full-resolution input blocks, beside the lower-resolution ones.
  FrustumOut fo;
  Ctx5 cl[4];
  const CtxProducer produce = [&](hipStream_t ps) -> int {
    if (B == 1 && !slots) RET_IF(engine_frustum(c, t_embed, v_embed, view_idx, TN, &fo, ps, /*half0=*/true));
    else RET_IF(engine_frustum_multi(c, B, slots, t_embed, v_embed, view_idx, TN, &fo, ps, /*half0=*/true));
    for (int l = 0; l < 4; ++l) {
      cl[l].p = fo.lvl[l];
      cl[l].f32 = 1;
    }
    if (fo.lvl0_half) {
      cl[0].p = fo.lvl0_half;
      cl[0].f32 = 0;
    }
    return 0;
  };
  float* xin = ws_alloc<float>(c, (size_t)Bv * HW * 8);
  float* ctx = ws_alloc<float>(c, (size_t)Bv * u.context_dim);
  int64_t* tt = ws_alloc<int64_t>(c, (size_t)Bv);
  float* eps = ws_alloc<float>(c, (size_t)Bv * HW * 4);
  float* eps_nchw = ws_alloc<float>(c, (size_t)Bv * HW * 4);
  WS_CHECK(xin && ctx && tt && eps && eps_nchw);
  // UNet batch order (morphable_diffusion.py:133-147): every conditional row first -- sample by sample, view by view --, then
  // the unconditional copies in the same order
  const long half_off = (long)B * TN;
  for (int b = 0; b < B; ++b) {
    hipLaunchKernelGGL(build_cfg_input_kernel, dim3(nblk((size_t)copies * TN * HW * 8)), dim3(256), 0, s,
                       x_noisy + (size_t)b * TN * 4 * HW, x_input + (size_t)b * 4 * HW, TN, HW, copies, xin + (size_t)b * TN * HW * 8,
                       half_off);
    hipLaunchKernelGGL(build_cfg_context_kernel, dim3(nblk((size_t)copies * TN * u.context_dim)), dim3(256), 0, s,
                       clip + (size_t)b * u.context_dim, TN, u.context_dim, copies, ctx + (size_t)b * TN * u.context_dim,
                       tt + (size_t)b * TN, timesteps[b], half_off);
  }
  HIP_CHECK_RET(hipGetLastError());
  (void)td; (void)vd;
  RET_IF(engine_unet(c, xin, 8, tt, ctx, Bv, B * TN, c->v.frustum_volume_depth, cl, eps, s, &produce));
  RET_IF(launch_nhwc_to_nchw(eps, 4, Bv, 4, HW, eps_nchw, s));
  const size_t n = (size_t)B * TN * 4 * HW;
  RET_IF(launch_cfg_ddim(eps_nchw, cfg ? eps_nchw + n : nullptr, cfg_scale, x_noisy, noise, sqrt_one_minus_at, sqrt_at,
                         sqrt_aprev, dir_coef, sigma, eps_out, x_prev, n, s));
  return 0;
}

int mvd_denoise_views(mvd_ctx* c, const float* x_noisy, const float* x_input, const float* clip, int64_t timestep,
                      const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN, float cfg_scale,
                      const float* noise, float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev, float dir_coef,
                      float sigma, float* eps_out, float* x_prev, void* stream) {
  return mvd_denoise_views_batch(c, 1, nullptr, x_noisy, x_input, clip, &timestep, t_embed, v_embed, view_idx, TN, cfg_scale, noise,
                                 sqrt_one_minus_at, sqrt_at, sqrt_aprev, dir_coef, sigma, eps_out, x_prev, stream);
}


// ------------------------------------------------------------------------------------------ view exchange over RCCL
// the sharded step's ONE collective -- the all-gather of the per-view vertex features -- behind the C
// ABI, on a stream of the caller, with a communicator the library owns.  librccl.so is opened on first use (dlopen: the
// library itself has no link-time dependency on it, single-GPU users never load it).  The unique id travels over whatever
// side channel the host has (the Python mirror broadcasts it with torch.distributed).
namespace {
struct RcclApi {
  void* h = nullptr;
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, mvd_rccl_id, int) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*group_start)() = nullptr;
  int (*group_end)() = nullptr;
  const char* (*get_error_string)(int) = nullptr;
};
RcclApi* rccl_api() {
  // C++11 magic static: initialised exactly once, by one thread, before any caller sees it (a one-thread-per-GPU host may call
  // mvd_comm_init for several contexts at the same time)
  static const RcclApi api = [] {
    RcclApi a;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      a.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (a.h) break;
    }
    if (a.h) {
      a.get_unique_id = (int (*)(void*))dlsym(a.h, "ncclGetUniqueId");
      a.comm_init_rank = (int (*)(void**, int, mvd_rccl_id, int))dlsym(a.h, "ncclCommInitRank");
      a.comm_destroy = (int (*)(void*))dlsym(a.h, "ncclCommDestroy");
      a.all_gather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(a.h, "ncclAllGather");
      a.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(a.h, "ncclAllReduce");
      a.group_start = (int (*)())dlsym(a.h, "ncclGroupStart");
      a.group_end = (int (*)())dlsym(a.h, "ncclGroupEnd");
      a.get_error_string = (const char* (*)(int))dlsym(a.h, "ncclGetErrorString");
      if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_gather || !a.all_reduce) a.h = nullptr;
    }
    return a;
  }();
  return api.h ? const_cast<RcclApi*>(&api) : nullptr;
}
int rccl_fail(RcclApi* a, const char* what, int rc) {
  static thread_local std::string msg;
  msg = std::string(what) + ": " + (a && a->get_error_string ? a->get_error_string(rc) : "RCCL error");
  return mvd_fail(msg.c_str());
}
}  // namespace

int mvd_comm_unique_id(mvd_rccl_id* id_out) {
  RcclApi* a = rccl_api();
  if (!a) return mvd_fail("mvd_comm_unique_id: librccl.so could not be opened");
  const int rc = a->get_unique_id(id_out);
  return rc ? rccl_fail(a, "ncclGetUniqueId", rc) : 0;
}

int mvd_comm_init(mvd_ctx* c, const mvd_rccl_id* id, int rank, int world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) return mvd_fail("mvd_comm_init: bad arguments");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  RcclApi* a = rccl_api();
  if (!a) return mvd_fail("mvd_comm_init: librccl.so could not be opened");
  if (c->comm) {
    a->comm_destroy(c->comm);
    c->comm = nullptr;
  }
  const int rc = a->comm_init_rank(&c->comm, world, *id, rank);
  if (rc) {
    c->comm = nullptr;
    return rccl_fail(a, "ncclCommInitRank", rc);
  }
  c->comm_rank = rank;
  c->comm_world = world;
  return 0;
}

int mvd_comm_destroy(mvd_ctx* c) {
  if (!c || !c->comm) return 0;
  RcclApi* a = rccl_api();
  if (a) a->comm_destroy(c->comm);
  c->comm = nullptr;
  c->comm_world = 1;
  c->comm_rank = 0;
  return 0;
}

int mvd_exchange_view_features(mvd_ctx* c, const float* local, float* all, int n_local, void* stream) {
  if (!c || !c->comm) return mvd_fail("mvd_exchange_view_features: no communicator (mvd_comm_init)");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!local || !all || n_local <= 0 || c->mesh.Nv <= 0) return mvd_fail("mvd_exchange_view_features: bad arguments / no mesh is set");
  RcclApi* a = rccl_api();
  if (!a) return mvd_fail("mvd_exchange_view_features: librccl.so could not be opened");
  // `all` holds comm_world * n_local views: the step's partition is even (SyncDDIMSampler.view_range), so a rank count that does
  // not divide the configured view count, or a slice of another size, is a caller error and not a gather of some other shape
  if (c->v.num_views > 0 && (long)c->comm_world * n_local != c->v.num_views)
    return mvd_fail("mvd_exchange_view_features: comm_world * n_local does not equal the configured view count");
  const size_t count = (size_t)n_local * c->mesh.Nv * 16;  // fp32 elements this rank contributes: [n_local][Nv][16]
  const int rc = a->all_gather(local, all, count, /* ncclFloat32 */ 7, c->comm, S(stream));
  return rc ? rccl_fail(a, "ncclAllGather", rc) : 0;
}

int mvd_comm_all_reduce(mvd_ctx* c, float* buf, size_t count, void* stream) {
  if (!c || !c->comm) return mvd_fail("mvd_comm_all_reduce: no communicator (mvd_comm_init)");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!buf || count == 0) return mvd_fail("mvd_comm_all_reduce: bad arguments");
  RcclApi* a = rccl_api();
  if (!a) return mvd_fail("mvd_comm_all_reduce: librccl.so could not be opened");
  const int rc = a->all_reduce(buf, buf, count, /* ncclFloat32 */ 7, /* ncclSum */ 0, c->comm, S(stream));
  return rc ? rccl_fail(a, "ncclAllReduce", rc) : 0;
}

// DDP's reducer on the gradient arena, entirely behind the C ABI (train_morphable_diffusion.py:302-303).  phase 0, right after
// mvd_train_unet_step: every bucket's ranges are all-reduced (sum, in place) on `comm_stream`, each bucket behind its own event
// -- one host call for all 26 buckets, no Python and no torch.distributed per bucket.  phase 1, after the conditioner's backward:
// `comm_stream` waits for `stream`, reduces every arena range no bucket covered, `stream` waits for `comm_stream`, and the arena is
// scaled by 1 / world on `stream`.  The arithmetic per element is the flat all-reduce's: sum over ranks, then the scale.
int mvd_train_sync_gradients(mvd_ctx* c, int phase, void* comm_stream, void* stream) {
  if (!c || !c->comm) return mvd_fail("mvd_train_sync_gradients: no communicator (mvd_comm_init)");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c->arena_g || c->arena_n == 0) return mvd_fail("mvd_train_sync_gradients: no gradient arena (mvd_train_enable)");
  if (phase != 0 && phase != 1) return mvd_fail("mvd_train_sync_gradients: phase is 0 (start) or 1 (finish)");
  RcclApi* a = rccl_api();
  if (!a) return mvd_fail("mvd_train_sync_gradients: librccl.so could not be opened");
  hipStream_t cs = S(comm_stream), ms = S(stream);
  auto reduce = [&](size_t off, size_t n) -> int {
    const int rc = a->all_reduce(c->arena_g + off, c->arena_g + off, n, 7, 0, c->comm, cs);
    return rc ? rccl_fail(a, "ncclAllReduce", rc) : 0;
  };
  if (phase == 0) {
    for (int k = 0; k < c->n_buckets; ++k) {
      const mvd_ctx::GradBucket& b = c->buckets[k];
      if (b.ev) HIP_CHECK_RET(hipStreamWaitEvent(cs, b.ev, 0));
      for (size_t r = 0; r < b.off.size(); ++r) RET_IF(reduce(b.off[r], b.len[r]));
    }
    c->grad_sync_started = true;
    return 0;
  }
  if (!c->ev_grad_sync[0]) {
    HIP_CHECK_RET(hipEventCreateWithFlags(&c->ev_grad_sync[0], hipEventDisableTiming));
    HIP_CHECK_RET(hipEventCreateWithFlags(&c->ev_grad_sync[1], hipEventDisableTiming));
  }
  HIP_CHECK_RET(hipEventRecord(c->ev_grad_sync[0], ms));  // the conditioner's backward wrote the rest of the arena on `stream`
  HIP_CHECK_RET(hipStreamWaitEvent(cs, c->ev_grad_sync[0], 0));
  std::vector<std::pair<size_t, size_t>> spans;
  if (c->grad_sync_started)
    for (int k = 0; k < c->n_buckets; ++k)
      for (size_t r = 0; r < c->buckets[k].off.size(); ++r) spans.push_back({c->buckets[k].off[r], c->buckets[k].off[r] + c->buckets[k].len[r]});
  std::sort(spans.begin(), spans.end());
  size_t pos = 0;
  for (auto& sp : spans) {
    if (sp.first < pos) return mvd_fail("mvd_train_sync_gradients: gradient buckets overlap");
    if (sp.first > pos) RET_IF(reduce(pos, sp.first - pos));
    pos = sp.second;
  }
  if (pos < c->arena_n) RET_IF(reduce(pos, c->arena_n - pos));
  c->grad_sync_started = false;
  HIP_CHECK_RET(hipEventRecord(c->ev_grad_sync[1], cs));
  HIP_CHECK_RET(hipStreamWaitEvent(ms, c->ev_grad_sync[1], 0));
  return launch_scale_copy(c->arena_g, c->arena_n, 1.0f / (float)c->comm_world, c->arena_g, ms);
}

// ------------------------------------------------------------------------------------------ test hooks
int mvd_op_conv(mvd_ctx* c, const float* x_nchw, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout,
                int ksize, int stride, int upsample, const float* resid_nchw, float* out_nchw, int force_splitk,
                void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int cpad = (Cin + 7) / 8 * 8, taps = ksize * ksize;
  const int Hv = H << upsample, Wv = W << upsample;
  const int Ho = (Hv - 1) / stride + 1, Wo = (Wv - 1) / stride + 1;
  float* xn = ws_alloc<float>(c, (size_t)B * H * W * cpad);
  half_t* wp = ws_alloc<half_t>(c, (size_t)taps * Cout * cpad);
  float* on = ws_alloc<float>(c, (size_t)B * Ho * Wo * Cout);
  float* rn = resid_nchw ? ws_alloc<float>(c, (size_t)B * Ho * Wo * Cout) : nullptr;
  WS_CHECK(xn && wp && on);
  RET_IF(launch_nchw_to_nhwc(x_nchw, B, Cin, H * W, xn, cpad, cpad, s));
  RET_IF(launch_pack_weight(w, Cout, cpad, taps, 0, 0, wp, s, Cin));
  if (rn) RET_IF(launch_nchw_to_nhwc(resid_nchw, B, Cout, Ho * Wo, rn, Cout, Cout, s));
  if (force_splitk == -3) {  // the UNet's last layer at inference: exact fp32 on the vector ALU (launch_out_conv_f32)
    if (taps != 9 || stride != 1 || upsample || rn || cpad != Cin) return mvd_fail("op_conv: the fp32 output-head form is 3x3, stride 1, no residual");
    RET_IF(launch_out_conv_f32(xn, Cin, w, bias, Cout, B, H, W, on, Cout, s));
    RET_IF(launch_nhwc_to_nchw(on, Cout, B, Cout, Ho * Wo, out_nchw, s));
    return 0;
  }
  if (force_splitk == -2) {  // the UNet's first layer at inference: exact fp32 on the vector ALU (launch_conv_in_f32)
    if (taps != 9 || stride != 1 || upsample || rn) return mvd_fail("op_conv: the fp32 first-layer form is 3x3, stride 1, no residual");
    RET_IF(launch_conv_in_f32(xn, cpad, w, Cin, bias, Cout, B, H, W, on, Cout, s));
    RET_IF(launch_nhwc_to_nchw(on, Cout, B, Cout, Ho * Wo, out_nchw, s));
    return 0;
  }
  ConvW cw;
  cw.w = wp; cw.bias = const_cast<float*>(bias); cw.N = Cout; cw.Cin = cpad; cw.taps = taps;
  GemmArgs g;
  g.a = xn; g.a_f32 = 1; g.lda = cpad; g.w = &cw; g.out = on; g.ldc = Cout; g.resid = rn; g.ldr = Cout;
  g.use_bias = bias != nullptr; g.force_splitk = force_splitk;
  if (cpad % 64 == 0) {  // exercise the fp16-source paths (incl. the LDS-halo 3x3 kernels) the UNet uses
    half_t* xh = ws_alloc<half_t>(c, (size_t)B * H * W * cpad);
    WS_CHECK(xh);
    RET_IF(launch_f32_to_f16(xn, xh, (size_t)B * H * W * cpad, s));
    g.a = xh;
    g.a_f32 = 0;
    const int bnx = Cout % 160 == 0 ? 160 : (Cout % 128 == 0 ? 128 : 0);
    if (taps == 9 && bnx && !getenv("MVD_NO_CONV3X")) {  // ... and the conv3x form of the weights (k_conv3x.hip), as build_conv3x_streams does
      cw.wx = ws_alloc<half_t>(c, conv3x_stream_halfs(Cout, cpad, bnx));
      WS_CHECK(cw.wx);
      RET_IF(conv3x_pack(wp, Cout, cpad, bnx, cw.wx, s));
      cw.wx_bn = bnx;
    }
  }
  if (upsample && taps == 9 && stride == 1 && cpad == Cin && B * H * W >= 2048) {
    // the UNet's path for large upsample convs: four parity-folded 2x2 convs (fp32 source, as in the decoder)
    cw.w_up = ws_alloc<half_t>(c, (size_t)16 * Cout * Cin);
    WS_CHECK(cw.w_up);
    RET_IF(launch_pack_upconv_weight(w, Cout, Cin, cw.w_up, s));
    g.a = xn; g.a_f32 = 1;
    RET_IF(run_upconv2d(c, g, B, H, W, s));
  } else {
    RET_IF(run_conv2d(c, g, B, H, W, stride, upsample, s));
  }
  RET_IF(launch_nhwc_to_nchw(on, Cout, B, Cout, Ho * Wo, out_nchw, s));
  return 0;
}

int mvd_op_conv3d(mvd_ctx* c, const float* x, int B, int Cin, int D, int H, int W, const float* w, const float* bias,
                  int Cout, int stride, int transposed, const float* resid, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  if (Cin % 8) return mvd_fail("op_conv3d: Cin must be a multiple of 8");
  const int Do = transposed ? 2 * D : (D - 1) / stride + 1, Ho = transposed ? 2 * H : (H - 1) / stride + 1,
            Wo = transposed ? 2 * W : (W - 1) / stride + 1;
  float* xn = ws_alloc<float>(c, (size_t)B * D * H * W * Cin);
  half_t* wp = ws_alloc<half_t>(c, (size_t)27 * Cout * Cin);
  float* on = ws_alloc<float>(c, (size_t)B * Do * Ho * Wo * Cout);
  WS_CHECK(xn && wp && on);
  RET_IF(launch_nchw_to_nhwc(x, B, Cin, D * H * W, xn, Cin, Cin, s));
  RET_IF(launch_pack_weight(w, Cout, Cin, 27, transposed, 0, wp, s, Cin));
  if (resid) RET_IF(launch_nchw_to_nhwc(resid, B, Cout, Do * Ho * Wo, on, Cout, Cout, s));
  ConvW cw;
  cw.w = wp; cw.bias = const_cast<float*>(bias); cw.N = Cout; cw.Cin = Cin; cw.taps = 27;
  GemmArgs g;
  g.a = xn; g.a_f32 = 1; g.lda = Cin; g.w = &cw; g.out = on; g.ldc = Cout; g.use_bias = bias != nullptr;
  if (resid) { g.resid = on; g.ldr = Cout; }
  if (Cin % 64 == 0) {  // fp16-source path (LDS-DMA implicit GEMM), as in the frustum network's inner layers
    half_t* xh = ws_alloc<half_t>(c, (size_t)B * D * H * W * Cin);
    WS_CHECK(xh);
    RET_IF(launch_f32_to_f16(xn, xh, (size_t)B * D * H * W * Cin, s));
    g.a = xh;
    g.a_f32 = 0;
  }
  if (transposed) RET_IF(run_convT3d(c, g, B, D, H, W, s));
  else RET_IF(run_conv3d(c, g, B, D, H, W, stride, s));
  RET_IF(launch_nhwc_to_nchw(on, Cout, B, Cout, Do * Ho * Wo, out, s));
  return 0;
}

int mvd_op_linear(mvd_ctx* c, const float* a, int M, int K, const float* w, const float* bias, int N, int geglu,
                  const float* resid, int a_half, int force_splitk, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  if (K % 8) return mvd_fail("op_linear: K must be a multiple of 8");
  half_t* wp = ws_alloc<half_t>(c, (size_t)N * K);
  float* bp = ws_alloc<float>(c, (size_t)N);
  WS_CHECK(wp && bp);
  RET_IF(launch_pack_weight(w, N, K, 1, 0, geglu, wp, s, K));
  if (bias) {
    if (geglu) RET_IF(launch_permute_geglu_bias(bias, N, bp, s));
    else HIP_CHECK_RET(hipMemcpyAsync(bp, bias, (size_t)N * 4, hipMemcpyDeviceToDevice, s));
  }
  ConvW cw;
  cw.w = wp; cw.bias = bias ? bp : nullptr; cw.N = N; cw.Cin = K; cw.taps = 1;
  GemmArgs g;
  g.a = a; g.a_f32 = 1; g.lda = K; g.w = &cw; g.out = out; g.ldc = geglu ? N / 2 : N; g.geglu = geglu;
  g.use_bias = bias != nullptr;
  g.force_splitk = force_splitk;
  if (resid) {
    g.resid = resid; g.resid_f32 = 1; g.ldr = N;
  }
  if (a_half) {
    half_t* ah = ws_alloc<half_t>(c, (size_t)M * K);
    WS_CHECK(ah);
    RET_IF(launch_f32_to_f16(a, ah, (size_t)M * K, s));
    g.a = ah; g.a_f32 = 0;
  }
  RET_IF(run_linear(c, g, 1, M, s));
  return 0;
}

int mvd_op_group_norm(mvd_ctx* c, const float* x_nchw, int B, int C, int HW, int groups, const float* gamma,
                      const float* beta, float eps, int act, float* out_nchw, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  float* xn = ws_alloc<float>(c, (size_t)B * HW * C);
  half_t* y = ws_alloc<half_t>(c, (size_t)B * HW * C);
  float* yf = ws_alloc<float>(c, (size_t)B * HW * C);
  WS_CHECK(xn && y && yf);
  RET_IF(launch_nchw_to_nhwc(x_nchw, B, C, HW, xn, C, C, s));
  NormW n;
  n.g = const_cast<float*>(gamma); n.b = const_cast<float*>(beta); n.C = C;
  RET_IF(run_group_norm(c, xn, C, B, HW, n, groups, eps, act, nullptr, y, C, s));
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk((size_t)B * HW * C)), dim3(256), 0, s, y, yf, (size_t)B * HW * C);
  RET_IF(launch_nhwc_to_nchw(yf, C, B, C, HW, out_nchw, s));
  return 0;
}

int mvd_op_layer_norm(mvd_ctx* c, const float* x, int rows, int C, const float* gamma, const float* beta, float* out,
                      void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  half_t* y = ws_alloc<half_t>(c, (size_t)rows * C);
  WS_CHECK(y);
  RET_IF(launch_layernorm(x, rows, C, gamma, beta, 1e-5f, y, s));
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk((size_t)rows * C)), dim3(256), 0, s, y, out, (size_t)rows * C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int mvd_op_attention(mvd_ctx* c, const float* q, const float* k, const float* v, int B, int T, int heads, int d, float* out,
                     void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int C = heads * d, rows = B * T;
  half_t* qkv = ws_alloc<half_t>(c, (size_t)rows * 3 * C);
  half_t* o = ws_alloc<half_t>(c, (size_t)rows * C);
  WS_CHECK(qkv && o);
  hipLaunchKernelGGL(pack_qkv_rows_kernel, dim3(nblk((size_t)rows * C)), dim3(256), 0, s, q, k, v, rows, C, qkv);
  RET_IF(launch_attention(qkv, 3 * C, qkv + 2 * C, 3 * C, o, C, B, T, heads, d, s));
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk((size_t)rows * C)), dim3(256), 0, s, o, out, (size_t)rows * C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// backward-kernel hooks (tests/test_gpu_train_ops.py): each against torch.autograd of the same op
int mvd_op_attention_bwd(mvd_ctx* c, const float* q, const float* k, const float* v, const float* d_out, int B, int T, int heads, int d,
                         float* dq, float* dk, float* dv, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int C = heads * d, rows = B * T;
  half_t* qkv = ws_alloc<half_t>(c, (size_t)rows * 3 * C);
  half_t* o = ws_alloc<half_t>(c, (size_t)rows * C);
  half_t* do16 = ws_alloc<half_t>(c, (size_t)rows * C);
  half_t* dqkv = ws_alloc<half_t>(c, (size_t)rows * 3 * C);
  float* lse = ws_alloc<float>(c, (size_t)B * heads * T);
  float* delta = ws_alloc<float>(c, (size_t)B * heads * T);
  float* tmp = ws_alloc<float>(c, (size_t)rows * 3 * C);
  WS_CHECK(qkv && o && do16 && dqkv && lse && delta && tmp);
  hipLaunchKernelGGL(pack_qkv_rows_kernel, dim3(nblk((size_t)rows * C)), dim3(256), 0, s, q, k, v, rows, C, qkv);
  RET_IF(launch_attention(qkv, 3 * C, qkv + 2 * C, 3 * C, o, C, B, T, heads, d, s));
  RET_IF(bwd_cast_rows(d_out, C, rows, C, C, do16, s));
  RET_IF(bwd_attention(qkv, 3 * C, o, do16, C, dqkv, 3 * C, lse, delta, B, T, heads, d, s));
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk((size_t)rows * 3 * C)), dim3(256), 0, s, dqkv, tmp, (size_t)rows * 3 * C);
  HIP_CHECK_RET(hipGetLastError());
  float* outs[3] = {dq, dk, dv};
  for (int i = 0; i < 3; ++i)
    HIP_CHECK_RET(hipMemcpy2DAsync(outs[i], (size_t)C * 4, tmp + (size_t)i * C, (size_t)3 * C * 4, (size_t)C * 4, rows, hipMemcpyDeviceToDevice, s));
  return 0;
}

// x, dy, dx: [B, rows, C] channels-last
int mvd_op_group_norm_bwd(mvd_ctx* c, const float* x, const float* dy, int B, int rows, int C, int groups, const float* gamma,
                          const float* beta, float eps, int act, float* dx, float* dgamma, float* dbeta, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  // the slabbed form the training step uses; MVD_GN_BWD_ONE_BLOCK=1: the one-workgroup-per-(sample, group) kernel
  static const bool one_block = getenv("MVD_GN_BWD_ONE_BLOCK") != nullptr;
  const int S = one_block ? 1 : bwd_gn_slabs(B, groups, rows);
  float* dg = ws_alloc<float>(c, (size_t)B * S * C);
  float* db = ws_alloc<float>(c, (size_t)B * S * C);
  float* part = ws_alloc<float>(c, (size_t)B * groups * S * 4);
  WS_CHECK(dg && db && part);
  if (one_block)
    RET_IF(bwd_group_norm(x, C, nullptr, 0, dy, C, B, rows, C, groups, gamma, beta, eps, act, dx, C, 0, dg, db, nullptr, s));
  else
    RET_IF(bwd_group_norm_slab(x, C, nullptr, 0, dy, C, B, rows, C, groups, gamma, beta, eps, act, dx, C, 0, part,
                               part + (size_t)B * groups * S * 2, dg, db, nullptr, S, s));
  RET_IF(bwd_sum_rows_add(dg, B * S, C, C, dgamma, 0, s));
  return bwd_sum_rows_add(db, B * S, C, C, dbeta, 0, s);
}

int mvd_op_layer_norm_bwd(mvd_ctx* c, const float* x, const float* dy, int rows, int C, const float* gamma, float* dx, float* dgamma,
                          float* dbeta, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  float* part = ws_alloc<float>(c, (size_t)bwd_ln_max_blocks() * 2 * C);
  WS_CHECK(part);
  int nb = 0;
  RET_IF(bwd_layer_norm(x, C, dy, C, rows, C, gamma, 1e-5f, dx, C, 0, part, &nb, s));
  RET_IF(bwd_sum_rows_add(part, nb, C, 2 * C, dgamma, 0, s));
  return bwd_sum_rows_add(part + C, nb, C, 2 * C, dbeta, 0, s);
}

int mvd_bench_conv(mvd_ctx* c, int B, int C, int H, int W, int Cout, int iters, float* ms_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const size_t na = (size_t)B * H * W * C, nw = (size_t)9 * Cout * C;
  half_t* a = ws_alloc<half_t>(c, na);
  half_t* w = ws_alloc<half_t>(c, nw);
  float* o = ws_alloc<float>(c, (size_t)B * H * W * Cout);
  WS_CHECK(a && w && o);
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk(na)), dim3(256), 0, s, a, na, 17u);
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk(nw)), dim3(256), 0, s, w, nw, 91u);
  ConvW cw;
  cw.w = w; cw.N = Cout; cw.Cin = C; cw.taps = 9;
  const int bnx = Cout % 160 == 0 ? 160 : (Cout % 128 == 0 ? 128 : 0);
  if (C % 64 == 0 && bnx && !getenv("MVD_NO_CONV3X")) {
    cw.wx = ws_alloc<half_t>(c, conv3x_stream_halfs(Cout, C, bnx));
    WS_CHECK(cw.wx);
    RET_IF(conv3x_pack(w, Cout, C, bnx, cw.wx, s));
    cw.wx_bn = bnx;
  }
  GemmArgs g;
  g.a = a; g.lda = C; g.w = &cw; g.out = o; g.ldc = Cout; g.use_bias = false;
  RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));  // warm-up
  hipEvent_t e0, e1;
  HIP_CHECK_RET(hipEventCreate(&e0));
  HIP_CHECK_RET(hipEventCreate(&e1));
  HIP_CHECK_RET(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));
  HIP_CHECK_RET(hipEventRecord(e1, s));
  HIP_CHECK_RET(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_out = ms / (float)iters;
  return 0;
}

int mvd_vae_decode(mvd_ctx* c, const float* z, int B, int h, int w, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (!z || !out || B <= 0) return mvd_fail("mvd_vae_decode: bad argument");
  return engine_vae_decode(c, z, B, h, w, out, S(stream));
}

int mvd_vae_encode(mvd_ctx* c, const float* x, int B, int H, int W, float* moments, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (!x || !moments || B <= 0) return mvd_fail("mvd_vae_encode: bad argument");
  return engine_vae_encode(c, x, B, H, W, moments, S(stream));
}

int mvd_clip_encode(mvd_ctx* c, const float* x, int B, int H, int W, float* out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c || !c->finalized) return mvd_fail("weights not finalized");
  if (!x || !out || B <= 0) return mvd_fail("mvd_clip_encode: bad argument");
  return engine_clip_encode(c, x, B, H, W, out, S(stream));
}

int mvd_clip_embed_dim(mvd_ctx* c) { return (c && c->finalized && c->clip.present) ? c->clip.embed : 0; }

int mvd_probe_config(mvd_ctx* c, int mode, const char* family, int stride) {
  if (!c) return mvd_fail("mvd_probe_config: null context");
  if (mode < 0 || mode > 2 || (mode == 2 && !family)) return mvd_fail("mvd_probe_config: bad mode");
  c->probe_mode = mode;
  c->probe_only = family ? family : "";
  c->probe_stride = stride > 1 ? stride : 1;
  if (mode) {  // a new measurement starts
    c->probe_used = 0;
    c->probe_counter = 0;
    c->probe_fam.clear();
    c->probe_empty.clear();
    c->probe_null.clear();
  }
  return 0;
}

int mvd_probe_report(mvd_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap < 3) return mvd_fail("mvd_probe_report: bad argument");
  if (hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  std::string out = "[";
  bool first = true;
  for (auto& kv : c->probe_fam) {
    double ms = 0.0;
    for (size_t slot : kv.second.ev) {
      HIP_CHECK_RET(hipEventSynchronize(c->probe_ev[slot + 1]));
      float t = 0.f;
      HIP_CHECK_RET(hipEventElapsedTime(&t, c->probe_ev[slot], c->probe_ev[slot + 1]));
      ms += t;
    }
    char line[512];
    snprintf(line, sizeof line,
             "%s{\"family\": \"%s\", \"launches\": %ld, \"sampled\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, "
             "\"all_flops\": %.6e, \"all_bytes\": %.6e}",
             first ? "" : ", ", kv.first.c_str(), kv.second.launches, kv.second.sampled, ms, kv.second.s_flops, kv.second.s_bytes,
             kv.second.flops, kv.second.bytes);
    out += line;
    first = false;
  }
  if (!c->probe_empty.empty()) {  // what a bracket costs by itself (survey mode), as a pseudo-family
    double ms = 0.0;
    for (size_t e : c->probe_empty) {
      HIP_CHECK_RET(hipEventSynchronize(c->probe_ev[e + 1]));
      float t = 0.f;
      HIP_CHECK_RET(hipEventElapsedTime(&t, c->probe_ev[e], c->probe_ev[e + 1]));
      ms += t;
    }
    char line[256];
    snprintf(line, sizeof line,
             "%s{\"family\": \"(empty bracket)\", \"launches\": %zu, \"sampled\": %zu, \"ms\": %.6f, \"flops\": 0, \"bytes\": 0, "
             "\"all_flops\": 0, \"all_bytes\": 0}",
             first ? "" : ", ", c->probe_empty.size(), c->probe_empty.size(), ms);
    out += line;
  }
  if (!c->probe_null.empty()) {  // event overhead + dispatch latency of a launch (survey mode), as a pseudo-family
    double ms = 0.0;
    for (size_t e : c->probe_null) {
      HIP_CHECK_RET(hipEventSynchronize(c->probe_ev[e + 1]));
      float t = 0.f;
      HIP_CHECK_RET(hipEventElapsedTime(&t, c->probe_ev[e], c->probe_ev[e + 1]));
      ms += t;
    }
    char line[256];
    snprintf(line, sizeof line,
             "%s{\"family\": \"(null-kernel bracket)\", \"launches\": %zu, \"sampled\": %zu, \"ms\": %.6f, \"flops\": 0, \"bytes\": 0, "
             "\"all_flops\": 0, \"all_bytes\": 0}",
             out.size() > 1 ? ", " : "", c->probe_null.size(), c->probe_null.size(), ms);
    out += line;
  }
  out += "]";
  if (out.size() + 1 > cap) return mvd_fail("mvd_probe_report: buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}

int mvd_bench_linear(mvd_ctx* c, int M, int K, int N, int flags, int iters, float* ms_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const size_t na = (size_t)M * K, nw = (size_t)N * K, no = (size_t)M * N;
  const bool cold = flags & 32;  // evict the operands (L2 + memory-side cache) before every timed launch
  const size_t flush_bytes = cold ? (size_t)768 << 20 : 0;
  half_t* a = ws_alloc<half_t>(c, na);
  half_t* w = ws_alloc<half_t>(c, nw);
  float* o = ws_alloc<float>(c, no);
  float* r = ws_alloc<float>(c, no);
  float* bias = ws_alloc<float>(c, (size_t)N + 64 * (size_t)N);
  char* flush = cold ? ws_alloc<char>(c, flush_bytes) : nullptr;
  float* a32 = (flags & 64) ? ws_alloc<float>(c, na) : nullptr;  // fp32 activations (converted while staged)
  WS_CHECK(a && w && o && r && bias && (!cold || flush) && (!(flags & 64) || a32));
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk(na)), dim3(256), 0, s, a, na, 17u);
  if (a32) hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(na)), dim3(256), 0, s, a, a32, na);
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk(nw)), dim3(256), 0, s, w, nw, 91u);
  HIP_CHECK_RET(hipMemsetAsync(r, 0, no * sizeof(float), s));
  HIP_CHECK_RET(hipMemsetAsync(bias, 0, ((size_t)N + 64 * (size_t)N) * sizeof(float), s));
  ConvW cw;
  cw.w = w; cw.N = N; cw.Cin = K; cw.taps = 1; cw.bias = bias;
  GemmArgs g;
  g.a = a; g.lda = K; g.w = &cw; g.out = o; g.use_bias = (flags & 8) != 0;
  if (a32) { g.a = a32; g.a_f32 = 1; }
  g.geglu = (flags & 4) ? 1 : 0;
  g.ldc = g.geglu ? N / 2 : N;
  g.out_f32 = (flags & 2) ? 0 : 1;
  if (flags & 1) { g.resid = r; g.resid_f32 = 1; g.ldr = N; }
  const int nb = M % 32 == 0 ? 32 : 1;  // samples (per-sample bias rows)
  if (flags & 16) { g.rowbias = bias + N; g.rb_ld = N; }
  RET_IF(run_linear(c, g, nb, M, s));  // warm-up
  hipEvent_t e0, e1;
  HIP_CHECK_RET(hipEventCreate(&e0));
  HIP_CHECK_RET(hipEventCreate(&e1));
  float total = 0.f;
  if (cold) {
    for (int i = 0; i < iters; ++i) {
      HIP_CHECK_RET(hipMemsetAsync(flush, i & 0xFF, flush_bytes, s));
      HIP_CHECK_RET(hipEventRecord(e0, s));
      RET_IF(run_linear(c, g, nb, M, s));
      HIP_CHECK_RET(hipEventRecord(e1, s));
      HIP_CHECK_RET(hipEventSynchronize(e1));
      float ms = 0.f;
      HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
      total += ms;
    }
  } else {
    HIP_CHECK_RET(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) RET_IF(run_linear(c, g, nb, M, s));
    HIP_CHECK_RET(hipEventRecord(e1, s));
    HIP_CHECK_RET(hipEventSynchronize(e1));
    HIP_CHECK_RET(hipEventElapsedTime(&total, e0, e1));
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_out = total / (float)iters;
  return 0;
}

int mvd_bench_group_norm(mvd_ctx* c, int B, int C, int HW, int groups, int flags, int iters, float* ms_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const size_t n = (size_t)B * HW * C;
  const int split = flags & 1, wo = split ? 3 : 1;
  float* x = ws_alloc<float>(c, n);
  half_t* tmp = ws_alloc<half_t>(c, n);
  half_t* y = ws_alloc<half_t>(c, n * wo);
  float* gb = ws_alloc<float>(c, (size_t)2 * C);
  WS_CHECK(x && tmp && y && gb);
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk(n)), dim3(256), 0, s, tmp, n, 29u);
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(n)), dim3(256), 0, s, tmp, x, n);
  hipLaunchKernelGGL(fill_pattern_f16_kernel, dim3(nblk((size_t)2 * C)), dim3(256), 0, s, tmp, (size_t)2 * C, 5u);
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk((size_t)2 * C)), dim3(256), 0, s, tmp, gb, (size_t)2 * C);
  NormW nw;
  nw.g = gb; nw.b = gb + C; nw.C = C;
  RET_IF(run_group_norm(c, x, C, B, HW, nw, groups, 1e-5f, ACT_SILU, nullptr, y, C * wo, s, 0, split));  // warm-up
  // flags & 2: the APPLY pass alone (statistics given): what a GroupNorm costs once its statistics come out of the producer's epilogue
  float* partial = nullptr;
  int nslabs = 0;
  if (flags & 2) {
    partial = ws_alloc<float>(c, (size_t)B * gn_max_slabs() * groups * 2);
    WS_CHECK(partial);
    RET_IF(launch_gn_stats(x, C, B, HW, C, groups, nullptr, 0, partial, &nslabs, s));
  }
  hipEvent_t e0, e1;
  HIP_CHECK_RET(hipEventCreate(&e0));
  HIP_CHECK_RET(hipEventCreate(&e1));
  HIP_CHECK_RET(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) {
    if (flags & 2) RET_IF(launch_gn_apply(x, C, B, HW, C, groups, nullptr, 0, partial, nslabs, nw.g, nw.b, 1e-5f, ACT_SILU, y, C * wo, s, split));
    else RET_IF(run_group_norm(c, x, C, B, HW, nw, groups, 1e-5f, ACT_SILU, nullptr, y, C * wo, s, 0, split));
  }
  HIP_CHECK_RET(hipEventRecord(e1, s));
  HIP_CHECK_RET(hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms_out = ms / (float)iters;
  return 0;
}

// The row-chain kernel (k_rowchain.hip) on fp32 operands in the reference's layouts: packs the weight stream, runs the kernel
// once (iters > 0: `iters` more times between two events -> *ms_out = mean milliseconds) and returns the result in fp32.
int mvd_op_st_tail(mvd_ctx* c, int C, int rows, int T, const float* ao, const float* xin, const float* rowbias, const float* w_ao,
                   const float* b_ao, const float* ln_g, const float* ln_b, const float* w1, const float* b1, const float* w2,
                   const float* b2, const float* w_po, const float* b_po, const float* resid, float* out, int flags, int iters,
                   float* ms_out, void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int has_ao = flags & 1, has_po = (flags & 2) ? ((flags & 8) ? 2 : 1) : 0, split = (flags & 4) && !has_po;
  if (!rowchain_takes(C, rows, T)) return mvd_fail("op_st_tail: shape not supported by the row-chain kernel");
  const size_t n = (size_t)rows * C;
  half_t* stream_w = ws_alloc<half_t>(c, rowchain_stream_halfs(C, has_ao, has_po));
  float* tmp = ws_alloc<float>(c, (size_t)8 * C);
  half_t* aoh = has_ao ? ws_alloc<half_t>(c, n) : nullptr;
  half_t* oh = has_po ? nullptr : ws_alloc<half_t>(c, n * (split ? 3 : 1));
  WS_CHECK(stream_w && tmp && (!has_ao || aoh) && (has_po || oh));
  RcWeights w;
  w.w_ao = w_ao; w.ln_g = ln_g; w.ln_b = ln_b; w.w1 = w1; w.b1 = b1; w.w2 = w2; w.b2 = b2; w.w_po = w_po;
  RET_IF(rowchain_pack(w, C, has_ao, has_po, tmp, stream_w, s));
  if (has_ao) RET_IF(launch_f32_to_f16(ao, aoh, n, s));
  RowChain p;
  memset(&p, 0, sizeof p);
  p.stream = stream_w; p.rows = rows; p.T = T;
  p.ao = aoh; p.ld_ao = C; p.xin = xin; p.ld_x = C; p.b_ao = b_ao; p.rowbias = rowbias; p.rb_ld = C;
  p.b_po = b_po; p.resid = resid; p.ld_r = C;
  p.out = has_po ? (void*)out : (void*)oh; p.ld_o = has_po ? C : (split ? 3 * C : C); p.out_split = split ? C : 0;
  RET_IF(launch_rowchain(p, C, has_ao, has_po, s));
  if (iters > 0) {
    hipEvent_t e0, e1;
    HIP_CHECK_RET(hipEventCreate(&e0));
    HIP_CHECK_RET(hipEventCreate(&e1));
    HIP_CHECK_RET(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) RET_IF(launch_rowchain(p, C, has_ao, has_po, s));
    HIP_CHECK_RET(hipEventRecord(e1, s));
    HIP_CHECK_RET(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (ms_out) *ms_out = ms / (float)iters;
  }
  if (!has_po) {
    if (split) {  // [hi | lo | hi] rows -> hi + lo (NaN where the third block differs from the first)
      hipLaunchKernelGGL(split_rows_to_f32_kernel, dim3(nblk(n)), dim3(256), 0, s, oh, rows, C, out);
    } else {
      hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(n)), dim3(256), 0, s, oh, out, n);
    }
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// The row-head kernel (k_rowchain.hip: proj_in -> t0, LayerNorm1, q | k | v) on fp32 operands in the reference's layouts, C = 320.
int mvd_op_st_head(mvd_ctx* c, int rows, int xp, const float* n0, const float* w_pi, const float* b_pi, const float* ln_g, const float* ln_b,
                   const float* w_q, const float* w_k, const float* w_v, float* t0_out, float* qkv_out, int iters, float* ms_out,
                   void* stream) {
  if (c && hipSetDevice(c->device) != hipSuccess) return mvd_fail("hipSetDevice failed");
  if (!c) return mvd_fail("null context");
  hipStream_t s = S(stream);
  WsScope ws_scope(c);
  const int C = RH_C;
  const size_t n = (size_t)rows * C;
  half_t* stream_w = ws_alloc<half_t>(c, rowhead_stream_halfs(xp));
  float* tmp = ws_alloc<float>(c, (size_t)3 * C);
  half_t* n0h = ws_alloc<half_t>(c, n * (xp ? 3 : 1));
  half_t* qkvh = ws_alloc<half_t>(c, 3 * n);
  WS_CHECK(stream_w && tmp && n0h && qkvh);
  RhWeights w;
  w.w_pi = w_pi; w.ln_g = ln_g; w.ln_b = ln_b; w.w_q = w_q; w.w_k = w_k; w.w_v = w_v;
  RET_IF(rowhead_pack(w, xp, tmp, stream_w, s));
  if (xp) RET_IF(launch_rows_f32_to_f16_split(n0, C, rows, C, n0h, s));  // [hi | lo | hi] rows, as the GroupNorm's split mode writes
  else RET_IF(launch_f32_to_f16(n0, n0h, n, s));
  RowHead p;
  p.stream = stream_w; p.rows = rows; p.n0 = n0h; p.ld_n0 = xp ? 3 * C : C; p.b_pi = b_pi; p.t0 = t0_out; p.ld_t0 = C; p.qkv = qkvh; p.ld_qkv = 3 * C;
  RET_IF(launch_rowhead(p, xp, s));
  if (iters > 0) {
    hipEvent_t e0, e1;
    HIP_CHECK_RET(hipEventCreate(&e0));
    HIP_CHECK_RET(hipEventCreate(&e1));
    HIP_CHECK_RET(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) RET_IF(launch_rowhead(p, xp, s));
    HIP_CHECK_RET(hipEventRecord(e1, s));
    HIP_CHECK_RET(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (ms_out) *ms_out = ms / (float)iters;
  }
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3(nblk(3 * n)), dim3(256), 0, s, qkvh, qkv_out, 3 * n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // extern "C"
