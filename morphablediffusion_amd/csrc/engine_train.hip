// Training step of the multi-view UNet: forward with a tape, MSE loss, and the backward pass through
// EVERY block of DepthWiseAttention (reference: training_step morphable_diffusion.py:520-549 from `self.model(...)` on, then
// loss.backward(); blocks openaimodel.py:256-276 (ResBlock), modules/attention.py:265-336 (SpatialTransformer /
// BasicTransformerBlock), attention.py:49-84 (DepthTransformer), openaimodel.py:110-160 (Up / Downsample), :717-721 (head)).
//
// * dgrad runs on the forward MFMA kernels with the adjoint weights ConvW::wT (transposed, tap-flipped fp16 packs derived
//   from the forward packs at finalize / repack time);
// * wgrad is a plain MFMA GEMM dW[Cout][Cin*taps] += dY^T col(X) whose operands are made K-contiguous by transposing casts
//   (k_bwd.hip: tcast / im2colT); its output layout IS the reference's parameter layout, accumulated straight into the
//   gradient arena;
// * norms / GEGLU / softmax / reductions are the fp32 kernels of k_bwd.hip;
// * the DepthTransformers are differentiated in their UNFOLDED form (to_q / to_k / to_v / to_out separately, as the
//   reference's parameters are), re-computed from the block input with the master weights; their GEMMs go through `tgemm`
//   (fp32 in HBM -> fp16 operands -> MFMA);
// * master parameters, gradients and Adam moments are flat fp32 arenas with one layout (sorted by state_dict key): the
//   reference's two optimiser groups (morphable_diffusion.py:627-646) are two contiguous ranges, the DDP gradient
//   all-reduce (train_morphable_diffusion.py:302-303) is ONE collective on one buffer.
// Activation gradients carry the caller's loss scale (fp16 MFMA operands); nothing here un-scales -- the optimiser does.
#include <string.h>

#include <algorithm>
#include <chrono>

#include "engine.h"

// k_bwd.hip
int bwd_tcast(const void* src, int src_f32, long ld, int R, int C, half_t* dst, int Rp, hipStream_t s, int split = 0, half_t* rows16 = nullptr,
              int Cp = 0);
int bwd_im2colT(const void* src, int src_f32, long ld, int B, int H, int W, int C, int stride, int ups, half_t* dst, int Rp, hipStream_t s);
int bwd_cast_rows(const float* src, long ld, long rows, int C, int Cp, half_t* dst, hipStream_t s, int split = 0);
int bwd_pack_dgrad(const half_t* w, int taps, int N, int ldw, int Cl, int Np, half_t* wT, hipStream_t s, int flip = 1);
int bwd_group_norm(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                   const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* dg_part,
                   float* db_part, float* dpre_part, hipStream_t s);
int bwd_sum_rows_add(const float* part, int R, int C, long ldp, float* out, int accum, hipStream_t s);
int bwd_sum_rows_multi(int n, const float* const* part, const int* R, const int* C, const long* ldp, float* const* out, const int* accum,
                       hipStream_t s);
int bwd_gn_slabs(int B, int G, int rows);
int bwd_group_norm_fwd(const float* x, long ld, int B, int rows, int C, int G, const float* gamma, const float* beta, float eps, int act,
                       float* y, long ldy, float* part, int S, hipStream_t s);
int bwd_group_norm_slab(const float* x, long ld, const float* pre, int pld, const float* dy, long ldy, int B, int rows, int C, int G,
                        const float* gamma, const float* beta, float eps, int act, float* dx, long lddx, int accum, float* part,
                        float* part2, float* dg_part, float* db_part, float* dpre_part, int S, hipStream_t s);
int bwd_outer_add(const float* a, long lda, const float* b, long ldb, float* C, int M, int N, int K, hipStream_t s);
int bwd_colsum_samples(const void* v, int v_f32, long ld, int B, int rows, int C, float* out, long ldo, hipStream_t s);
int bwd_ln_max_blocks();
int bwd_layer_norm(const float* x, long ld, const float* dy, long ldy, int rows, int C, const float* gamma, float eps, float* dx,
                   long lddx, int accum, float* part, int* nblk, hipStream_t s);
int bwd_geglu(const half_t* pre, const float* dgg, long ldg, long rows, int N, half_t* dpre, hipStream_t s);
int bwd_geglu_unpermute_add(const float* src, int N, int C, float* dst, hipStream_t s);
int bwd_add_views(float* out, long ldo, const float* a, long lda, const float* b, long ldb, long rows, int C, int accum, hipStream_t s);
int bwd_upsample2(const float* dup, int B, int H, int W, int C, float* dx, long lddx, int accum, hipStream_t s);
int bwd_col2im3(const float* dcol, int B, int H, int W, int C, int stride, float* dx, long lddx, int accum, hipStream_t s);
int bwd_silu_inplace(float* g, const float* u, size_t n, hipStream_t s);
int bwd_attention(const half_t* qkv, int ld3, const half_t* o, const half_t* dO, int ldo, half_t* dqkv, int ldd, float* lse, float* delta,
                  int B, int T, int heads, int d, hipStream_t s);
int bwd_silu_fwd(const float* u, float* out, size_t n, hipStream_t s);
// k_cond_bwd.hip
int cbwd_frustum_scatter(const float* d_out, const ViewCam* cams, const int* view_idx, int TN, int D, int S, int V, float vol_len, int persp,
                         float* d_vol, hipStream_t s);
int cbwd_latent_scatter(const float* d_vol, const int* grid, int gd, int gh, int gw, const float* min_xyz, const int* out_sh, float voxel,
                        int V, float vol_len, float* d_rows, hipStream_t s);
int cbwd_vertex_scatter(const float* d_out, const ViewCam* cams, const int* view_idx, int n_views, const float* verts, int Nv, int V,
                        float vol_len, int S, int persp, float* d_feats, hipStream_t s);
int cbwd_fuse_scratch_floats(int Nv);
int cbwd_fuse(const float* d_fused, const float* vf, const float* w, int n_views, int Nv, int total_views, float* d_vf, float* dw, float* db,
              float* part,
              hipStream_t s);
int cbwd_bn_scratch_floats(int n, int C);
int cbwd_bn_rows_relu(const float* xraw, float* dy, int n, int C, const float* gamma, const float* beta, const float* stats,
                      float* scratch, float* dgamma, float* dbeta, hipStream_t s);
int cbwd_sparse_wgrad_chunks(int n_out);
int cbwd_sparse_conv(const float* in, const int* nbr, const float* d_out, int n_out, int Cin, int Cout, const float* w, float* d_in,
                     float* dw_packed, float* dw_part, hipStream_t s);
int cbwd_sparse_w_unpack_add(const float* pk, int Cin, int Cout, int layout, float* dst, hipStream_t s);
int cbwd_sparse_wgrad_mfma(const float* in, const int* nbr, const float* d_out, int n_out, int Cin, int Cout, float* dw_packed,
                           float* dw_part, hipStream_t s);
int cbwd_sparse_inverse_table(const int* nbr_down, int n_out, int n_in, int* inv, hipStream_t s);
int cbwd_sparse_fold_dups(float* d, const int* nbr, int n, int C, hipStream_t s);
int cbwd_im2colT3d(const void* src, int src_f32, long ld, int B, int D, int H, int W, int C, int stride, half_t* dst, int Rp, hipStream_t s);
int cbwd_small_linear_bwd(const float* g, long ldg, int rows, int N, const half_t* w, int K, float* out, long ldo, int accum, hipStream_t s);
// k_train.hip
int train_im2col3(const float* X, int B, int H, int W, int C, float* col, hipStream_t s);
int train_col2im3(const float* dcol, int B, int H, int W, int C, float* dX, hipStream_t s);
int train_perm_w3(const float* src, int N, int C, int to_mat, float* dst, hipStream_t s);
int train_depth_fwd(const float* q, const float* k, const float* v, int R, int HW, int D, int hn, int hd, float scale, float* attn,
                    float* z, hipStream_t s);
int train_depth_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dz, int R, int HW, int D, int hn,
                    int hd, float scale, float* dq, float* dk, float* dv, hipStream_t s);
int train_add_inplace(float* a, const float* b, size_t n, hipStream_t s);
int train_copy_rows(const float* src, int ld, long rows, int C, float* dst, hipStream_t s);
int train_add_bias_rows(float* x, long rows, int C, const float* bias, hipStream_t s);

// ---------------------------------------------------------------------------------------------------------------------
// arenas
// ---------------------------------------------------------------------------------------------------------------------
bool engine_hot_key(const std::string& k) {
  return k.rfind("model.diffusion_model.", 0) == 0 || k.rfind("spatial_volume.", 0) == 0 || k.rfind("time_embed.", 0) == 0;
}
namespace {
inline bool hot_key(const std::string& k) { return engine_hot_key(k); }
inline int up8(int n) { return (n + 7) & ~7; }
inline int up64(int n) { return (n + 63) & ~63; }
}  // namespace

// finalize in training mode: the uploaded fp32 tensors of the hot path become the master parameters, laid out back to back
// (each aligned to 64 floats) in key order
int engine_train_setup(mvd_ctx* c) {
  c->params.clear();
  c->param_index.clear();
  size_t off = 0;
  for (auto& kv : c->raw) {
    if (!hot_key(kv.first)) continue;
    // BatchNorm running statistics are buffers, not parameters: they stay outside the arena (and outside the optimiser)
    const std::string& k = kv.first;
    const bool buffer = k.size() > 13 && (k.compare(k.size() - 13, 13, ".running_mean") == 0 || k.compare(k.size() - 12, 12, ".running_var") == 0);
    if (buffer) continue;
    mvd_ctx::ParamRec r;
    r.key = k;
    r.off = off;
    r.numel = kv.second.numel;
    r.shape = kv.second.shape;
    c->param_index[k] = c->params.size();
    c->params.push_back(r);
    off += (kv.second.numel + 63) & ~(size_t)63;
  }
  c->arena_n = off;
  if (!off) return mvd_fail("training mode: no trainable tensors were uploaded");
  HIP_CHECK_RET(hipMalloc((void**)&c->arena_p, off * sizeof(float)));
  HIP_CHECK_RET(hipMalloc((void**)&c->arena_g, off * sizeof(float)));
  c->arena_owned[0] = c->arena_owned[1] = true;
  HIP_CHECK_RET(hipMemset(c->arena_p, 0, off * sizeof(float)));
  HIP_CHECK_RET(hipMemset(c->arena_g, 0, off * sizeof(float)));
  HIP_CHECK_RET(hipMalloc((void**)&c->found_inf, sizeof(int)));
  HIP_CHECK_RET(hipMemset(c->found_inf, 0, sizeof(int)));
  for (auto& r : c->params) {
    RawTensor& t = c->raw[r.key];
    HIP_CHECK_RET(hipMemcpy(c->arena_p + r.off, t.d, r.numel * sizeof(float), hipMemcpyDeviceToDevice));
    hipFree(t.d);
    t.d = c->arena_p + r.off;
  }
  return 0;
}

float* engine_grad(mvd_ctx* c, const std::string& key) {
  auto it = c->param_index.find(key);
  return it == c->param_index.end() ? nullptr : c->arena_g + c->params[it->second].off;
}
const float* engine_master(mvd_ctx* c, const std::string& key) {
  auto it = c->param_index.find(key);
  return it == c->param_index.end() ? nullptr : c->arena_p + c->params[it->second].off;
}

// adjoint weights of every GEMM of the UNet trunk (ResBlocks, SpatialTransformers, conv_in / down / up, the output conv)
namespace {
int make_wT(mvd_ctx* c, ConvW& w, int flip = 1) {
  if (!w.w) return 0;
  const int Cl = w.cin_l > 0 ? w.cin_l : w.Cin;
  w.Np = up8(w.N);
  RET_IF(engine_dmalloc(c, (void**)&w.wT, (size_t)w.taps * Cl * w.Np * sizeof(half_t)));
  engine_build_rotate(c);
  return bwd_pack_dgrad(w.w, w.taps, w.N, w.Cin, Cl, w.Np, w.wT, c->bs, flip);
}
}  // namespace
int engine_build_dgrad(mvd_ctx* c) {
  for (auto& r : c->res) {
    RET_IF(make_wT(c, r.c1));
    RET_IF(make_wT(c, r.c2));
    if (r.has_skip) RET_IF(make_wT(c, r.skip));
  }
  for (auto& t : c->st)
    for (ConvW* w : {&t.proj_in, &t.qkv, &t.attn_out, &t.ff1, &t.ff2, &t.proj_out}) RET_IF(make_wT(c, *w));
  for (auto& w : c->convs) RET_IF(make_wT(c, w));
  RET_IF(make_wT(c, c->out_conv));
  return 0;
}
// ... and of the conditioner's dense convolutions.  A strided conv's adjoint is launched as a transposed conv and vice versa
// (their tap tables carry the o = 2 i - 1 + k relation), so those packs are transposed but not tap-flipped.
int engine_build_dgrad_cond(mvd_ctx* c) {
  for (int i = 0; i < 3; ++i) {
    RET_IF(make_wT(c, c->enc_blocks[i].c1));
    RET_IF(make_wT(c, c->enc_blocks[i].c2));
  }
  RET_IF(make_wT(c, c->enc_final));
  RET_IF(make_wT(c, c->fr_conv0));
  for (int i = 0; i < 6; ++i) RET_IF(make_wT(c, c->fr_blocks[i].conv, c->fr_blocks[i].stride == 1 ? 1 : 0));
  for (int i = 0; i < 3; ++i) RET_IF(make_wT(c, c->fr_up[i].conv, 0));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// backward machinery
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct Bwd {
  mvd_ctx* c;
  hipStream_t s;
  int B;
  Fwd* f;
  TrainTape* tape;
  float* demb = nullptr;  // [B][emb_total]: dL/d(emb_layers output) of every ResBlock
  float* da2 = nullptr;   // [B][a2_total]: dL/d(attn2 output) of every SpatialTransformer
  float* dsrc[4] = {nullptr, nullptr, nullptr, nullptr};
};

struct Opnd {  // a GEMM operand in HBM
  const void* p;
  int f32;   // 1 fp32, 0 fp16
  long ld;
  int trans;  // A: 0 = [M][K], 1 = stored [K][M];  B: 1 = stored [N][K] (weight layout), 0 = stored [K][N]
};

// fp16 [rows][Kp] image of an operand: `stored_rows` = the operand is stored [rows][K] (row-major, stride ld), otherwise
// [K][rows].  A row-major fp16 operand is used in place (need_dense: only when its rows are back to back).
// split (1: [hi | lo | hi], 2: [hi | hi | lo], fp32 operands only): rows of 3 Kp halfs, see k_bwd.hip tcast_kernel.
int operand16(mvd_ctx* c, const Opnd& o, int rows, int K, int Kp, bool stored_rows, bool need_dense, int split, const half_t** out,
              int* ld, hipStream_t s) {
  if (stored_rows && !o.f32) {
    if (split) return mvd_fail("tgemm: extended precision needs fp32 operands");
    if (o.ld % 8 || K != Kp || (((uintptr_t)o.p) & 15) || (need_dense && o.ld != Kp))
      return mvd_fail("tgemm: a row-major fp16 operand needs K % 8 == 0, ld % 8 == 0 (dense for the weight side)");
    *out = (const half_t*)o.p;
    *ld = (int)o.ld;
    return 0;
  }
  const int w = split ? 3 * Kp : Kp;
  half_t* d = ws_alloc<half_t>(c, (size_t)rows * w);
  WS_CHECK(d);
  if (stored_rows) RET_IF(bwd_cast_rows((const float*)o.p, o.ld, rows, K, Kp, d, s, split));
  else RET_IF(bwd_tcast(o.p, o.f32, o.ld, K, rows, d, Kp, s, split));  // stored [K][rows] -> [rows][Kp]
  *out = d;
  *ld = w;
  return 0;
}

// C[M][N] (ldc) (+)= op(A)[M][K] op(B)[K][N] on the MFMA GEMM kernels.  A.trans: stored [K][M]; B.trans: stored [N][K]
// (the weight layout), otherwise [K][N].  xp: extended precision -- both operands split into fp16 hi + lo parts, three
// products accumulated in fp32 (a_hi b_hi + a_lo b_hi + a_hi b_lo) by concatenation along K: ~2^-22 relative operand error
// instead of 2^-11, at three times the MFMA work.  Used where the backward pass is ill-conditioned (the DepthTransformers'
// softmax over nearly uniform depth weights, ReLU masks re-derived from re-computed activations).
int tgemm(mvd_ctx* c, Opnd A, Opnd Bm, float* C, int ldc, int M, int N, int K, bool accum, hipStream_t s, bool xp = false) {
  if (M <= 0 || N <= 0 || K <= 0) return mvd_fail("tgemm: empty problem");
  WsScope scope(c, WS_TEMP);
  const int Kp = up8(K);
  if (xp && (!A.f32 || !Bm.f32)) xp = false;
  const half_t *a16, *b16;
  int lda, ldb;
  RET_IF(operand16(c, A, M, K, Kp, !A.trans, false, xp ? 1 : 0, &a16, &lda, s));
  RET_IF(operand16(c, Bm, N, K, Kp, Bm.trans != 0, true, xp ? 2 : 0, &b16, &ldb, s));
  ConvW w;
  w.w = const_cast<half_t*>(b16);
  w.N = N;
  w.Cin = xp ? 3 * Kp : Kp;
  w.taps = 1;
  GemmArgs g;
  g.a = a16; g.lda = lda; g.w = &w; g.out = C; g.ldc = ldc; g.use_bias = false;
  if (accum) {
    g.resid = C;
    g.ldr = ldc;
  }
  return run_linear(c, g, 1, M, s);
}
inline Opnd F32(const float* p, long ld, int trans) { return Opnd{p, 1, ld, trans}; }
inline Opnd F16(const half_t* p, long ld, int trans) { return Opnd{p, 0, ld, trans}; }

// dense fp16 [rows][up8(C)] copy of an fp32 gradient (the A operand of a dgrad GEMM)
int grad16(mvd_ctx* c, const float* g, long ld, long rows, int C, half_t** out, hipStream_t s, half_t** outT = nullptr) {
  half_t* d = ws_alloc<half_t>(c, (size_t)rows * up8(C));
  WS_CHECK(d);
  *out = d;
  if (outT) {  // also the K-contiguous image [C][up64(rows)] for the layer's weight-gradient GEMM, from the same read
    const int Rp = up64((int)rows);
    half_t* t = ws_alloc<half_t>(c, (size_t)C * Rp);
    WS_CHECK(t);
    *outT = t;
    return bwd_tcast(g, 1, ld, (int)rows, C, t, Rp, s, 0, d, up8(C));
  }
  return bwd_cast_rows(g, ld, rows, C, up8(C), d, s);
}

// dX[rows][cin] (lddx) (+)= dY16[rows][Np] wT   (Linear / 1x1 conv adjoint on the forward GEMM kernels)
int dgrad_linear(Bwd& b, const ConvW& w, const half_t* dy16, int ld16, void* dx, int lddx, int out_f32, int rows, bool accum) {
  if (!w.wT) return mvd_fail("training backward: adjoint weights missing (context not finalized in training mode)");
  ConvW t;
  t.w = w.wT;
  t.N = w.cin_l > 0 ? w.cin_l : w.Cin;
  t.Cin = w.Np;
  t.taps = 1;
  GemmArgs g;
  g.a = dy16; g.lda = ld16; g.w = &t; g.out = dx; g.out_f32 = out_f32; g.ldc = lddx; g.use_bias = false;
  if (accum) {
    g.resid = dx;
    g.resid_f32 = out_f32;
    g.ldr = lddx;
  }
  return run_linear(b.c, g, b.B, rows, b.s);
}
// 3x3 / stride 1 conv adjoint: the forward conv kernels on the tap-flipped transposed weights
int dgrad_conv3(Bwd& b, const ConvW& w, const half_t* dy16, float* dx, int lddx, int H, int W, bool accum) {
  if (!w.wT || w.taps != 9) return mvd_fail("training backward: 3x3 adjoint weights missing");
  ConvW t;
  t.w = w.wT;
  t.N = w.cin_l > 0 ? w.cin_l : w.Cin;
  t.Cin = w.Np;
  t.taps = 9;
  GemmArgs g;
  g.a = dy16; g.lda = w.Np; g.w = &t; g.out = dx; g.ldc = lddx; g.use_bias = false;
  if (accum) {
    g.resid = dx;
    g.ldr = lddx;
  }
  return run_conv2d(b.c, g, b.B, H, W, 1, 0, b.s);
}
// G[N][K2] += dyT[N][Rp] colT[K2][Rp]^T
int wgrad_gemm(Bwd& b, const half_t* dyT, int N, const half_t* colT, int K2, int Rp, float* G, bool accum = true) {
  if (!G) return 0;  // not a parameter of this run
  ConvW t;
  t.w = const_cast<half_t*>(colT);
  t.N = K2;
  t.Cin = Rp;
  t.taps = 1;
  GemmArgs g;
  g.a = dyT; g.lda = Rp; g.w = &t; g.out = G; g.ldc = K2; g.use_bias = false;
  if (accum) {
    g.resid = G;
    g.ldr = K2;
  }
  return run_linear(b.c, g, 1, N, b.s);
}
// weight + bias gradient of a Linear / 1x1 conv: dy fp32 [rows][N] (ld), x [rows][K] (fp16 or fp32)
int wgrad_linear(Bwd& b, const ConvW& w, const float* dy, long ldy, const void* x, int x_f32, long ldx, int rows, int K,
                 const half_t* dyT_pre = nullptr) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int Rp = up64(rows), N = w.N;
  if (float* G = engine_grad(c, w.key)) {
    const half_t* dyT = dyT_pre;
    if (!dyT) {
      half_t* t = ws_alloc<half_t>(c, (size_t)N * Rp);
      WS_CHECK(t);
      RET_IF(bwd_tcast(dy, 1, ldy, rows, N, t, Rp, b.s));
      dyT = t;
    }
    half_t* xT = ws_alloc<half_t>(c, (size_t)K * Rp);
    WS_CHECK(xT);
    RET_IF(bwd_tcast(x, x_f32, ldx, rows, K, xT, Rp, b.s));
    RET_IF(wgrad_gemm(b, dyT, N, xT, K, Rp, G));
  }
  if (float* Gb = w.bkey.empty() ? nullptr : engine_grad(c, w.bkey)) {
    float* part = ws_alloc<float>(c, (size_t)b.B * N);
    WS_CHECK(part);
    RET_IF(bwd_colsum_samples(dy, 1, ldy, b.B, rows / b.B, N, part, N, b.s));
    RET_IF(bwd_sum_rows_add(part, b.B, N, N, Gb, 1, b.s));
  }
  return 0;
}
// weight + bias gradient of a 3x3 conv (stride / nearest-upsampled input as in the forward): x [B,H,W,K] physical
int wgrad_conv3(Bwd& b, const ConvW& w, const float* dy, long ldy, const void* x, int x_f32, long ldx, int H, int W, int K, int stride,
                int ups, const half_t* dyT_pre = nullptr) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int Ho = ((H << ups) - 1) / stride + 1, Wo = ((W << ups) - 1) / stride + 1;
  const int rows = b.B * Ho * Wo, Rp = up64(rows), N = w.N;
  if (float* G = engine_grad(c, w.key)) {
    const half_t* dyT = dyT_pre;
    if (!dyT) {
      half_t* t = ws_alloc<half_t>(c, (size_t)N * Rp);
      WS_CHECK(t);
      RET_IF(bwd_tcast(dy, 1, ldy, rows, N, t, Rp, b.s));
      dyT = t;
    }
    half_t* colT = ws_alloc<half_t>(c, (size_t)K * 9 * Rp);
    WS_CHECK(colT);
    RET_IF(bwd_im2colT(x, x_f32, ldx, b.B, H, W, K, stride, ups, colT, Rp, b.s));
    RET_IF(wgrad_gemm(b, dyT, N, colT, K * 9, Rp, G));
  }
  if (float* Gb = w.bkey.empty() ? nullptr : engine_grad(c, w.bkey)) {
    float* part = ws_alloc<float>(c, (size_t)b.B * N);
    WS_CHECK(part);
    RET_IF(bwd_colsum_samples(dy, 1, ldy, b.B, Ho * Wo, N, part, N, b.s));
    RET_IF(bwd_sum_rows_add(part, b.B, N, N, Gb, 1, b.s));
  }
  return 0;
}
// gain and bias gradients of a norm in one launch: Gw[c] += sum_r pw[r * ldp + c], Gb[c] += sum_r pb[r * ldp + c]
int sum_pair(Bwd& b, const float* pw, const float* pb, int R, int C, long ldp, float* Gw, float* Gb) {
  const float* part[2];
  float* out[2];
  int n = 0;
  if (Gw) part[n] = pw, out[n++] = Gw;
  if (Gb) part[n] = pb, out[n++] = Gb;
  if (!n) return 0;
  const int Rs[2] = {R, R}, Cs[2] = {C, C}, acc[2] = {1, 1};
  const long lds[2] = {ldp, ldp};
  return bwd_sum_rows_multi(n, part, Rs, Cs, lds, out, acc, b.s);
}
// GroupNorm backward incl. its gain / bias gradients (slabbed: B * G * S workgroups)
// pre / pld: the per-sample pre-add of the forward norm (FiLM); dpre [B][ldp] (may be null) receives its gradient
int gn_backward(Bwd& b, const NormW& n, int groups, float eps, int act, const float* x, long ld, const float* dy, long ldy, int rows_ps,
                float* dx, long lddx, bool accum, const float* pre = nullptr, int pld = 0, float* dpre = nullptr, long ldp = 0) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int S = bwd_gn_slabs(b.B, groups, rows_ps);
  float* dg = ws_alloc<float>(c, (size_t)b.B * S * n.C);
  float* db = ws_alloc<float>(c, (size_t)b.B * S * n.C);
  float* dp = dpre ? ws_alloc<float>(c, (size_t)b.B * S * n.C) : nullptr;
  float* part = ws_alloc<float>(c, (size_t)b.B * groups * S * 4);
  WS_CHECK(dg && db && part && (dp || !dpre));
  RET_IF(bwd_group_norm_slab(x, ld, pre, pld, dy, ldy, b.B, rows_ps, n.C, groups, n.g, n.b, eps, act, dx, lddx, accum ? 1 : 0, part,
                             part + (size_t)b.B * groups * S * 2, dg, db, dp, S, b.s));
  RET_IF(sum_pair(b, dg, db, b.B * S, n.C, n.C, engine_grad(c, n.key + ".weight"), engine_grad(c, n.key + ".bias")));
  if (dpre) RET_IF(bwd_colsum_samples(dp, 1, n.C, b.B, S, n.C, dpre, ldp, b.s));  // per sample: its S slab rows
  return 0;
}
// GroupNorm forward in fp32 for the DepthTransformer re-computation
int gn_forward32(Bwd& b, const float* x, int rows_ps, int C, int G, const float* gamma, const float* beta, int act, float* y) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int S = bwd_gn_slabs(b.B, G, rows_ps);
  float* part = ws_alloc<float>(c, (size_t)b.B * G * S * 2);
  WS_CHECK(part);
  return bwd_group_norm_fwd(x, C, b.B, rows_ps, C, G, gamma, beta, 1e-5f, act, y, C, part, S, b.s);
}
int ln_backward(Bwd& b, const NormW& n, const float* x, long ld, const float* dy, long ldy, int rows, float* dx, long lddx, bool accum) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  float* part = ws_alloc<float>(c, (size_t)bwd_ln_max_blocks() * 2 * n.C);
  WS_CHECK(part);
  int nblk = 0;
  RET_IF(bwd_layer_norm(x, ld, dy, ldy, rows, n.C, n.g, 1e-5f, dx, lddx, accum ? 1 : 0, part, &nblk, b.s));
  return sum_pair(b, part, part + n.C, nblk, n.C, 2 * n.C, engine_grad(c, n.key + ".weight"), engine_grad(c, n.key + ".bias"));
}

// ---- ResBlock (openaimodel.py:256-276) ------------------------------------------------------------------------------
int bwd_res(Bwd& b, const ResW& r, const ResSaved& sv, View in, View dout, View din, bool accum, int H, int W) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  const int HW = H * W, rows = b.B * HW, cin = r.cin, cout = r.cout;
  float* d_a2 = ws_alloc<float>(c, (size_t)rows * cout);
  float* d_h1 = ws_alloc<float>(c, (size_t)rows * cout);
  float* d_a1 = ws_alloc<float>(c, (size_t)rows * cin);
  WS_CHECK(d_a2 && d_h1 && d_a1);
  half_t *dy16, *dyT, *dhT;
  RET_IF(grad16(c, dout.p, dout.ld, rows, cout, &dy16, b.s, &dyT));
  // out = conv2(a2) + b2 + skip(x)
  RET_IF(dgrad_conv3(b, r.c2, dy16, d_a2, cout, H, W, false));
  RET_IF(wgrad_conv3(b, r.c2, dout.p, dout.ld, sv.a2, 0, sv.ld2, H, W, cout, 1, 0, dyT));
  // a2 = silu(GN2(h1))
  RET_IF(gn_backward(b, r.n2, 32, 1e-5f, ACT_SILU, sv.h1, cout, d_a2, cout, HW, d_h1, cout, false));
  // h1 = conv1(a1) + b1 + emb[b]: the per-sample sums of d_h1 are the gradient of this block's slice of the stacked emb
  // projection, their sum over the samples the conv bias gradient
  RET_IF(bwd_colsum_samples(d_h1, 1, cout, b.B, HW, cout, b.demb + r.emb_off, c->emb_total, b.s));
  if (float* Gb = engine_grad(c, r.c1.bkey)) RET_IF(bwd_sum_rows_add(b.demb + r.emb_off, b.B, cout, c->emb_total, Gb, 1, b.s));
  half_t* dh16;
  RET_IF(grad16(c, d_h1, cout, rows, cout, &dh16, b.s, &dhT));
  RET_IF(dgrad_conv3(b, r.c1, dh16, d_a1, cin, H, W, false));
  {
    ConvW w1 = r.c1;
    w1.bkey.clear();  // the bias gradient was taken from the emb sums above
    RET_IF(wgrad_conv3(b, w1, d_h1, cout, sv.a1, 0, sv.ld1, H, W, cin, 1, 0, dhT));
  }
  // a1 = silu(GN1(x))
  RET_IF(gn_backward(b, r.n1, 32, 1e-5f, ACT_SILU, in.p, in.ld, d_a1, cin, HW, din.p, din.ld, accum));
  if (r.has_skip) {
    RET_IF(dgrad_linear(b, r.skip, dy16, up8(cout), din.p, din.ld, 1, rows, true));
    RET_IF(wgrad_linear(b, r.skip, dout.p, dout.ld, in.p, 1, in.ld, rows, cin, dyT));
  } else {
    RET_IF(bwd_add_views(din.p, din.ld, dout.p, dout.ld, nullptr, 0, rows, cout, 1, b.s));
  }
  return 0;
}

// ---- SpatialTransformer (modules/attention.py:325-336, BasicTransformerBlock._forward :265-269) ----------------------
int bwd_st(Bwd& b, const STW& t, const STSaved& sv, View in, View dout, View din, bool accum, int H, int W) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  const int C = t.C, T = H * W, rows = b.B * T;
  float* d_t = ws_alloc<float>(c, (size_t)rows * C);       // dL/d t3, then + LN3 path = dL/d t2, then + LN1 path = dL/d t0
  float* d_gg = ws_alloc<float>(c, (size_t)rows * 4 * C);
  half_t* pre16 = ws_alloc<half_t>(c, (size_t)rows * 8 * C);
  half_t* dpre16 = ws_alloc<half_t>(c, (size_t)rows * 8 * C);
  float* d_l = ws_alloc<float>(c, (size_t)rows * C);       // dL/d l3, later dL/d l1, later dL/d n0
  half_t* d_ao16 = ws_alloc<half_t>(c, (size_t)rows * C);
  half_t* dqkv16 = ws_alloc<half_t>(c, (size_t)rows * 3 * C);
  float* lse = ws_alloc<float>(c, (size_t)b.B * t.heads * T);
  float* delta = ws_alloc<float>(c, (size_t)b.B * t.heads * T);
  float* tmpW = ws_alloc<float>(c, (size_t)8 * C * C);
  float* part = ws_alloc<float>(c, (size_t)b.B * 8 * C);
  WS_CHECK(d_t && d_gg && pre16 && dpre16 && d_l && d_ao16 && dqkv16 && lse && delta && tmpW && part);
  const std::string tb = t.key + ".transformer_blocks.0";
  half_t *g16, *gT;
  // out = proj_out(t3) + x
  RET_IF(grad16(c, dout.p, dout.ld, rows, C, &g16, b.s, &gT));
  RET_IF(dgrad_linear(b, t.proj_out, g16, C, d_t, C, 1, rows, false));
  RET_IF(wgrad_linear(b, t.proj_out, dout.p, dout.ld, sv.t3, 0, sv.ldt3, rows, C, gT));
  // t3 = t2 + ff2(gg)
  RET_IF(grad16(c, d_t, C, rows, C, &g16, b.s, &gT));
  RET_IF(dgrad_linear(b, t.ff2, g16, C, d_gg, 4 * C, 1, rows, false));
  RET_IF(wgrad_linear(b, t.ff2, d_t, C, sv.gg, 0, 4 * C, rows, 4 * C, gT));
  // gg = GEGLU(ff1(l3)): the pre-activations are re-computed (packed column order), never stored by the forward pass
  {
    GemmArgs g;
    g.a = sv.l3; g.lda = C; g.w = &t.ff1; g.out = pre16; g.out_f32 = 0; g.ldc = 8 * C;
    RET_IF(run_linear(c, g, b.B, rows, b.s));
  }
  RET_IF(bwd_geglu(pre16, d_gg, 4 * C, rows, 8 * C, dpre16, b.s));
  RET_IF(dgrad_linear(b, t.ff1, dpre16, 8 * C, d_l, C, 1, rows, false));
  {  // FF1 weight / bias gradient: computed in the packed row order, un-permuted into the reference's [value | gate] rows
    WsScope sc2(c, WS_TEMP);
    const int Rp = up64(rows);
    half_t* dyT = ws_alloc<half_t>(c, (size_t)8 * C * Rp);
    half_t* xT = ws_alloc<half_t>(c, (size_t)C * Rp);
    WS_CHECK(dyT && xT);
    if (float* G = engine_grad(c, t.ff1.key)) {
      RET_IF(bwd_tcast(dpre16, 0, 8 * C, rows, 8 * C, dyT, Rp, b.s));
      RET_IF(bwd_tcast(sv.l3, 0, C, rows, C, xT, Rp, b.s));
      RET_IF(wgrad_gemm(b, dyT, 8 * C, xT, C, Rp, tmpW, false));
      RET_IF(bwd_geglu_unpermute_add(tmpW, 8 * C, C, G, b.s));
    }
    if (float* Gb = engine_grad(c, t.ff1.bkey)) {
      RET_IF(bwd_colsum_samples(dpre16, 0, 8 * C, b.B, T, 8 * C, part, 8 * C, b.s));
      RET_IF(bwd_sum_rows_add(part, b.B, 8 * C, 8 * C, tmpW, 0, b.s));
      RET_IF(bwd_geglu_unpermute_add(tmpW, 8 * C, 1, Gb, b.s));
    }
  }
  // l3 = LN3(t2)
  RET_IF(ln_backward(b, t.ln3, sv.t2, C, d_l, C, rows, d_t, C, true));
  // t2 = t0 + to_out(ao) + b_o + attn2[b]
  RET_IF(grad16(c, d_t, C, rows, C, &g16, b.s, &gT));
  RET_IF(dgrad_linear(b, t.attn_out, g16, C, d_ao16, C, 0, rows, false));
  {
    ConvW wo = t.attn_out;
    wo.bkey.clear();
    RET_IF(wgrad_linear(b, wo, d_t, C, sv.ao, 0, C, rows, C, gT));
    // per-sample token sums of d_t2: gradient of the attn2 output (a per-sample constant) and, summed, of attn1's output bias
    RET_IF(bwd_colsum_samples(d_t, 1, C, b.B, T, C, b.da2 + t.a2_off, c->a2_total, b.s));
    if (float* Gb = engine_grad(c, t.attn_out.bkey)) RET_IF(bwd_sum_rows_add(b.da2 + t.a2_off, b.B, C, c->a2_total, Gb, 1, b.s));
  }
  // self-attention
  RET_IF(bwd_attention(sv.qkv, 3 * C, sv.ao, d_ao16, C, dqkv16, 3 * C, lse, delta, b.B, T, t.heads, C / t.heads, b.s));
  RET_IF(dgrad_linear(b, t.qkv, dqkv16, 3 * C, d_l, C, 1, rows, false));
  {
    WsScope sc2(c, WS_TEMP);
    const int Rp = up64(rows);
    half_t* dyT = ws_alloc<half_t>(c, (size_t)3 * C * Rp);
    half_t* xT = ws_alloc<half_t>(c, (size_t)C * Rp);
    WS_CHECK(dyT && xT);
    RET_IF(bwd_tcast(dqkv16, 0, 3 * C, rows, 3 * C, dyT, Rp, b.s));
    RET_IF(bwd_tcast(sv.l1, 0, C, rows, C, xT, Rp, b.s));
    RET_IF(wgrad_gemm(b, dyT, 3 * C, xT, C, Rp, tmpW, false));
    const char* names[3] = {".attn1.to_q.weight", ".attn1.to_k.weight", ".attn1.to_v.weight"};
    for (int i = 0; i < 3; ++i)
      if (float* G = engine_grad(c, tb + names[i]))
        RET_IF(bwd_add_views(G, C, tmpW + (size_t)i * C * C, C, nullptr, 0, C, C, 1, b.s));
  }
  // l1 = LN1(t0)
  RET_IF(ln_backward(b, t.ln1, sv.t0, C, d_l, C, rows, d_t, C, true));
  // t0 = proj_in(n0) + b
  RET_IF(grad16(c, d_t, C, rows, C, &g16, b.s, &gT));
  RET_IF(dgrad_linear(b, t.proj_in, g16, C, d_l, C, 1, rows, false));
  RET_IF(wgrad_linear(b, t.proj_in, d_t, C, sv.n0, 0, sv.ldn0, rows, C, gT));
  // n0 = GN(x) (eps 1e-6, no activation); out also carries x itself
  RET_IF(gn_backward(b, t.norm, 32, 1e-6f, ACT_NONE, in.p, in.ld, d_l, C, T, din.p, din.ld, accum));
  RET_IF(bwd_add_views(din.p, din.ld, dout.p, dout.ld, nullptr, 0, rows, C, 1, b.s));
  return 0;
}

// ---- conv_in / Downsample / Upsample (openaimodel.py:110-160) --------------------------------------------------------
int bwd_conv(Bwd& b, int kind, const ConvW& w, View in, View dout, View din, bool accum, bool need_din, int H, int W) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  const int stride = kind == OP_DOWN ? 2 : 1, ups = kind == OP_UP ? 1 : 0;
  const int Ho = ((H << ups) - 1) / stride + 1, Wo = ((W << ups) - 1) / stride + 1, rows_o = b.B * Ho * Wo;
  const int cin = w.cin_l > 0 ? w.cin_l : w.Cin, cout = w.N;
  RET_IF(wgrad_conv3(b, w, dout.p, dout.ld, in.p, 1, in.ld, H, W, cin, stride, ups));
  if (!need_din) return 0;
  half_t* dy16;
  RET_IF(grad16(c, dout.p, dout.ld, rows_o, cout, &dy16, b.s));
  if (kind == OP_UP) {
    float* d_up = ws_alloc<float>(c, (size_t)rows_o * cin);
    WS_CHECK(d_up);
    RET_IF(dgrad_conv3(b, w, dy16, d_up, cin, Ho, Wo, false));
    return bwd_upsample2(d_up, b.B, H, W, cin, din.p, din.ld, accum ? 1 : 0, b.s);
  }
  if (kind == OP_DOWN) {
    // dcol[r][t2 * cin + ci] = sum_co dy[r][co] wT[t2][ci][co]: one plain GEMM against the adjoint pack read as [9 cin][Np]
    float* dcol = ws_alloc<float>(c, (size_t)rows_o * 9 * cin);
    WS_CHECK(dcol);
    ConvW t;
    t.w = w.wT;
    t.N = 9 * cin;
    t.Cin = w.Np;
    t.taps = 1;
    GemmArgs g;
    g.a = dy16; g.lda = w.Np; g.w = &t; g.out = dcol; g.ldc = 9 * cin; g.use_bias = false;
    RET_IF(run_linear(c, g, b.B, rows_o, b.s));
    return bwd_col2im3(dcol, b.B, H, W, cin, 2, din.p, din.ld, accum ? 1 : 0, b.s);
  }
  return dgrad_conv3(b, w, dy16, din.p, din.ld, H, W, accum);
}

// ---- DepthTransformer (attention.py:49-84), unfolded, from the block input -----------------------------------------
int bwd_cond(Bwd& b, const CondW& cd, View in, View dout, View din, bool accum, int H, int W, int level) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  hipStream_t s = b.s;
  const int B = b.B, HW = H * W, R = B * HW, dim = cd.dim, I = cd.I, Cc = cd.Cc, hn = 4, hd = Cc / 2;
  const int D = b.tape->depth0 >> level;
  const long RC = (long)B * D * HW;
  const float scale = 1.0f / sqrtf((float)hd);
  const std::string P = cd.key + ".";
  auto Wm = [&](const char* n) { return engine_master(c, P + n); };
  auto G = [&](const char* n) { return engine_grad(c, P + n); };
  const float *w_pi = Wm("proj_in.0.weight"), *b_pi = Wm("proj_in.0.bias"), *g_pi = Wm("proj_in.1.weight"), *e_pi = Wm("proj_in.1.bias");
  const float *w_pc = Wm("proj_context.0.weight"), *g_pc = Wm("proj_context.1.weight"), *e_pc = Wm("proj_context.1.bias");
  const float *w_q = Wm("depth_attn.to_q.weight"), *w_k = Wm("depth_attn.to_k.weight"), *w_v = Wm("depth_attn.to_v.weight"),
              *w_o = Wm("depth_attn.to_out.weight");
  const float *g_o0 = Wm("proj_out.0.weight"), *e_o0 = Wm("proj_out.0.bias"), *w_c1 = Wm("proj_out.2.weight");
  const float *g_o3 = Wm("proj_out.3.weight"), *e_o3 = Wm("proj_out.3.bias"), *w_c2 = Wm("proj_out.5.weight");
  for (const float* q_ : {w_pi, b_pi, g_pi, e_pi, w_pc, g_pc, e_pc, w_q, w_k, w_v, w_o, g_o0, e_o0, w_c1, g_o3, e_o3, w_c2})
    if (!q_) return mvd_fail("training backward: DepthTransformer master weights missing");
  if (!b.tape->src || !b.tape->src[level].p || !b.tape->src[level].f32) return mvd_fail("training backward: fp32 context volume missing");
  const float* C0 = (const float*)b.tape->src[level].p;  // channels-last [B][D][HW][Cc]
  auto F = [&](size_t n) { return ws_alloc<float>(c, n); };
  // ---------------- forward recompute, every intermediate kept (fp32 in HBM, fp16 MFMA operands) ----------------
  float* X = F((size_t)R * dim);
  float *p = F((size_t)R * I), *pn = F((size_t)R * I);
  float *pc = F((size_t)RC * Cc), *cn = F((size_t)RC * Cc);
  float *q = F((size_t)R * I), *k = F((size_t)RC * I), *v = F((size_t)RC * I);
  float *attn = F((size_t)R * hn * D), *z = F((size_t)R * I), *o = F((size_t)R * I);
  float *a1 = F((size_t)R * I), *col1 = F((size_t)R * 9 * I), *o2 = F((size_t)R * I);
  float *a2 = F((size_t)R * I), *col2 = F((size_t)R * 9 * I);
  float *m_c1 = F((size_t)I * 9 * I), *m_c2 = F((size_t)dim * 9 * I);
  const size_t wmax = (size_t)std::max(dim, I) * 9 * I;  // the larger of the two 3x3 conv weights
  float *dcol = F((size_t)R * 9 * I), *dI1 = F((size_t)R * I), *dI2 = F((size_t)R * I), *mg = F(wmax);
  float *dk = F((size_t)RC * I), *dv = F((size_t)RC * I), *dcn = F((size_t)RC * Cc), *dpc = F((size_t)RC * Cc);
  float* dh = F((size_t)R * dim);
  float* mgp = F(wmax);
  WS_CHECK(mgp);
  WS_CHECK(X && p && pn && pc && cn && q && k && v && attn && z && o && a1 && col1 && o2 && a2 && col2 && m_c1 && m_c2 && dcol && dI1 &&
           dI2 && mg && dk && dv && dcn && dpc && dh);
  RET_IF(train_copy_rows(in.p, in.ld, R, dim, X, s));
  RET_IF(train_copy_rows(dout.p, dout.ld, R, dim, dh, s));
  RET_IF(train_perm_w3(w_c1, I, I, 1, m_c1, s));
  RET_IF(train_perm_w3(w_c2, dim, I, 1, m_c2, s));
  // proj_in: conv1x1 + bias, GN8, SiLU        (attention.py:52-56)
  RET_IF(tgemm(c, F32(X, dim, 0), F32(w_pi, dim, 1), p, I, R, I, dim, false, s, true));
  RET_IF(train_add_bias_rows(p, R, I, b_pi, s));
  RET_IF(gn_forward32(b, p, HW, I, 8, g_pi, e_pi, ACT_SILU, pn));
  // proj_context: conv1x1x1 (no bias), GN8, ReLU   (:57-61)
  RET_IF(tgemm(c, F32(C0, Cc, 0), F32(w_pc, Cc, 1), pc, Cc, (int)RC, Cc, Cc, false, s, true));
  RET_IF(gn_forward32(b, pc, D * HW, Cc, 8, g_pc, e_pc, ACT_RELU, cn));
  // depth attention   (:26-47)
  RET_IF(tgemm(c, F32(pn, I, 0), F32(w_q, I, 1), q, I, R, I, I, false, s, true));
  RET_IF(tgemm(c, F32(cn, Cc, 0), F32(w_k, Cc, 1), k, I, (int)RC, I, Cc, false, s, true));
  RET_IF(tgemm(c, F32(cn, Cc, 0), F32(w_v, Cc, 1), v, I, (int)RC, I, Cc, false, s, true));
  RET_IF(train_depth_fwd(q, k, v, R, HW, D, hn, hd, scale, attn, z, s));
  RET_IF(tgemm(c, F32(z, I, 0), F32(w_o, I, 1), o, I, R, I, I, false, s, true));
  // proj_out: GN8, ReLU, conv3x3, GN8, ReLU, conv3x3   (:63-70)
  RET_IF(gn_forward32(b, o, HW, I, 8, g_o0, e_o0, ACT_RELU, a1));
  RET_IF(train_im2col3(a1, B, H, W, I, col1, s));
  RET_IF(tgemm(c, F32(col1, 9 * I, 0), F32(m_c1, 9 * I, 1), o2, I, R, I, 9 * I, false, s, true));
  RET_IF(gn_forward32(b, o2, HW, I, 8, g_o3, e_o3, ACT_RELU, a2));
  RET_IF(train_im2col3(a2, B, H, W, I, col2, s));
  // ---------------- backward: dh = dL/d(x + proj_out(.)) ----------------
  // second conv3x3 of proj_out
  RET_IF(tgemm(c, F32(dh, dim, 1), F32(col2, 9 * I, 0), mg, 9 * I, dim, 9 * I, R, false, s, true));      // wgrad [dim][9][I]
  if (float* g_ = G("proj_out.5.weight")) {
    RET_IF(train_perm_w3(mg, dim, I, 0, mgp, s));  // [dim][9][I] -> the reference's [dim][I][3][3]
    RET_IF(train_add_inplace(g_, mgp, (size_t)dim * 9 * I, s));
  }
  RET_IF(tgemm(c, F32(dh, dim, 0), F32(m_c2, 9 * I, 0), dcol, 9 * I, R, 9 * I, dim, false, s, true));
  RET_IF(train_col2im3(dcol, B, H, W, I, dI1, s));                                                   // d a2
  {
    NormW n;
    n.g = const_cast<float*>(g_o3); n.b = const_cast<float*>(e_o3); n.C = I; n.key = P + "proj_out.3";
    RET_IF(gn_backward(b, n, 8, 1e-5f, ACT_RELU, o2, I, dI1, I, HW, dI2, I, false));
  }
  RET_IF(tgemm(c, F32(dI2, I, 1), F32(col1, 9 * I, 0), mg, 9 * I, I, 9 * I, R, false, s, true));         // first conv3x3
  if (float* g_ = G("proj_out.2.weight")) {
    RET_IF(train_perm_w3(mg, I, I, 0, mgp, s));
    RET_IF(train_add_inplace(g_, mgp, (size_t)I * 9 * I, s));
  }
  RET_IF(tgemm(c, F32(dI2, I, 0), F32(m_c1, 9 * I, 0), dcol, 9 * I, R, 9 * I, I, false, s, true));
  RET_IF(train_col2im3(dcol, B, H, W, I, dI1, s));                                                   // d a1
  {
    NormW n;
    n.g = const_cast<float*>(g_o0); n.b = const_cast<float*>(e_o0); n.C = I; n.key = P + "proj_out.0";
    RET_IF(gn_backward(b, n, 8, 1e-5f, ACT_RELU, o, I, dI1, I, HW, dI2, I, false));
  }
  // to_out (1x1, no bias): dI2 = d o
  if (float* g_ = G("depth_attn.to_out.weight")) RET_IF(tgemm(c, F32(dI2, I, 1), F32(z, I, 0), g_, I, I, I, R, true, s, true));
  RET_IF(tgemm(c, F32(dI2, I, 0), F32(w_o, I, 0), dI1, I, R, I, I, false, s, true));                      // d z
  float* dq = dI2;
  RET_IF(train_depth_bwd(q, k, v, attn, dI1, R, HW, D, hn, hd, scale, dq, dk, dv, s));
  if (float* g_ = G("depth_attn.to_q.weight")) RET_IF(tgemm(c, F32(dq, I, 1), F32(pn, I, 0), g_, I, I, I, R, true, s, true));
  if (float* g_ = G("depth_attn.to_k.weight")) RET_IF(tgemm(c, F32(dk, I, 1), F32(cn, Cc, 0), g_, Cc, I, Cc, (int)RC, true, s, true));
  if (float* g_ = G("depth_attn.to_v.weight")) RET_IF(tgemm(c, F32(dv, I, 1), F32(cn, Cc, 0), g_, Cc, I, Cc, (int)RC, true, s, true));
  // d cn = dk W_k + dv W_v ; proj_context backward
  RET_IF(tgemm(c, F32(dk, I, 0), F32(w_k, Cc, 0), dcn, Cc, (int)RC, Cc, I, false, s, true));
  RET_IF(tgemm(c, F32(dv, I, 0), F32(w_v, Cc, 0), dcn, Cc, (int)RC, Cc, I, true, s, true));
  {
    NormW n;
    n.g = const_cast<float*>(g_pc); n.b = const_cast<float*>(e_pc); n.C = Cc; n.key = P + "proj_context.1";
    RET_IF(gn_backward(b, n, 8, 1e-5f, ACT_RELU, pc, Cc, dcn, Cc, D * HW, dpc, Cc, false));
  }
  if (float* g_ = G("proj_context.0.weight")) RET_IF(tgemm(c, F32(dpc, Cc, 1), F32(C0, Cc, 0), g_, Cc, Cc, Cc, (int)RC, true, s, true));
  if (b.dsrc[level])  // gradient w.r.t. the context volume (several DepthTransformers share a level: accumulated)
    RET_IF(tgemm(c, F32(dpc, Cc, 0), F32(w_pc, Cc, 0), b.dsrc[level], Cc, (int)RC, Cc, Cc, true, s, true));
  // d pn = dq W_q ; proj_in backward
  RET_IF(tgemm(c, F32(dq, I, 0), F32(w_q, I, 0), dI1, I, R, I, I, false, s, true));
  float* dp = dcol;  // [R][I] fits
  {
    NormW n;
    n.g = const_cast<float*>(g_pi); n.b = const_cast<float*>(e_pi); n.C = I; n.key = P + "proj_in.1";
    RET_IF(gn_backward(b, n, 8, 1e-5f, ACT_SILU, p, I, dI1, I, HW, dp, I, false));
  }
  if (float* g_ = G("proj_in.0.weight")) RET_IF(tgemm(c, F32(dp, I, 1), F32(X, dim, 0), g_, dim, I, dim, R, true, s, true));
  if (float* g_ = G("proj_in.0.bias")) {
    float* part = F((size_t)B * I);
    WS_CHECK(part);
    RET_IF(bwd_colsum_samples(dp, 1, I, B, HW, I, part, I, s));
    RET_IF(bwd_sum_rows_add(part, B, I, I, g_, 1, s));
  }
  // d x = dh (residual) + dp W_pi
  RET_IF(tgemm(c, F32(dp, I, 0), F32(w_pi, dim, 0), din.p, din.ld, R, dim, I, accum, s, true));
  RET_IF(bwd_add_views(din.p, din.ld, dh, dim, nullptr, 0, R, dim, 1, s));
  return 0;
}

// ---- output head (openaimodel.py:717-721) ---------------------------------------------------------------------------
int bwd_head(Bwd& b, const float* dpred /* [rows][oc] channels-last */, float* d_final) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  const int S = c->u.image_size, mc = c->u.model_channels, oc = c->u.out_channels, rows = b.B * S * S;
  float* d_a = ws_alloc<float>(c, (size_t)rows * mc);
  WS_CHECK(d_a);
  half_t* dy16;
  RET_IF(grad16(c, dpred, oc, rows, oc, &dy16, b.s));
  RET_IF(dgrad_conv3(b, c->out_conv, dy16, d_a, mc, S, S, false));
  RET_IF(wgrad_conv3(b, c->out_conv, dpred, oc, b.tape->head_a, 0, b.tape->head_ld, S, S, mc, 1, 0));
  return gn_backward(b, c->out_norm, 32, 1e-5f, ACT_SILU, b.tape->final_h, mc, d_a, mc, S * S, d_final, mc, false);
}

// ---- time embedding MLP + the stacked ResBlock emb projections (openaimodel.py:527-532, :219-225) --------------------
int bwd_emb(Bwd& b) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  hipStream_t s = b.s;
  const int B = b.B, mc = c->u.model_channels, temb = 4 * mc, ET = c->emb_total;
  const std::string U = "model.diffusion_model.";
  float* u1 = ws_alloc<float>(c, (size_t)B * temb);
  float* u2 = ws_alloc<float>(c, (size_t)B * temb);
  float* d_e2 = ws_alloc<float>(c, (size_t)B * temb);
  float* d_e1 = ws_alloc<float>(c, (size_t)B * temb);
  WS_CHECK(u1 && u2 && d_e2 && d_e1);
  // every ResBlock's emb_layers.1: weight [cout][temb] += demb_slice^T silu(emb), bias += column sums (taken in bwd_res);
  // d silu(emb) = demb W_all
  for (auto& r : c->res) {
    if (float* G = engine_grad(c, r.key + ".emb_layers.1.weight"))
      RET_IF(bwd_outer_add(b.demb + r.emb_off, ET, b.tape->e2, temb, G, r.cout, temb, B, s));
    if (float* G = engine_grad(c, r.key + ".emb_layers.1.bias")) RET_IF(bwd_sum_rows_add(b.demb + r.emb_off, B, r.cout, ET, G, 1, s));
  }
  RET_IF(tgemm(c, F32(b.demb, ET, 0), F16(c->emb_all.w, temb, 0), d_e2, temb, B, temb, ET, false, s));
  // pre-activations of the two SiLUs (time_embed.0 / .2), re-computed with the forward kernels
  {
    ConvW w0, w2;
    w0.w = c->te0.w; w0.bias = c->te0.bias; w0.N = temb; w0.Cin = mc;
    w2.w = c->te2.w; w2.bias = c->te2.bias; w2.N = temb; w2.Cin = temb;
    GemmArgs g;
    g.a = b.tape->e0; g.a_f32 = 1; g.lda = mc; g.w = &w0; g.out = u1; g.ldc = temb;
    RET_IF(run_linear(c, g, B, B, s));
    g = GemmArgs();
    g.a = b.tape->e1; g.a_f32 = 1; g.lda = temb; g.w = &w2; g.out = u2; g.ldc = temb;
    RET_IF(run_linear(c, g, B, B, s));
  }
  RET_IF(bwd_silu_inplace(d_e2, u2, (size_t)B * temb, s));  // d u2
  if (float* G = engine_grad(c, U + "time_embed.2.weight")) RET_IF(bwd_outer_add(d_e2, temb, b.tape->e1, temb, G, temb, temb, B, s));
  if (float* G = engine_grad(c, U + "time_embed.2.bias")) RET_IF(bwd_sum_rows_add(d_e2, B, temb, temb, G, 1, s));
  RET_IF(tgemm(c, F32(d_e2, temb, 0), F16(c->te2.w, temb, 0), d_e1, temb, B, temb, temb, false, s));
  RET_IF(bwd_silu_inplace(d_e1, u1, (size_t)B * temb, s));  // d u1
  if (float* G = engine_grad(c, U + "time_embed.0.weight")) RET_IF(bwd_outer_add(d_e1, temb, b.tape->e0, mc, G, temb, mc, B, s));
  if (float* G = engine_grad(c, U + "time_embed.0.bias")) RET_IF(bwd_sum_rows_add(d_e1, B, temb, temb, G, 1, s));
  return 0;
}

// ---- attn2 of every SpatialTransformer: a single context token, i.e. out = to_out(to_v(ctx)) (modules/attention.py:187-203
// with one key: softmax == 1, so to_q / to_k / norm2 receive exactly zero gradient) ------------------------------------
int bwd_attn2(Bwd& b) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_BLOCK);
  hipStream_t s = b.s;
  const int B = b.B, cd = c->u.context_dim, AT = c->a2_total;
  for (auto& t : c->st) {
    WsScope sc2(c, WS_TEMP);
    const int C = t.C;
    const std::string a = t.key + ".transformer_blocks.0.attn2.";
    const float *wv = engine_master(c, a + "to_v.weight"), *wo = engine_master(c, a + "to_out.0.weight");
    if (!wv || !wo) return mvd_fail("training backward: attn2 master weights missing");
    float* u = ws_alloc<float>(c, (size_t)B * C);
    float* du = ws_alloc<float>(c, (size_t)B * C);
    WS_CHECK(u && du);
    const float* d = b.da2 + t.a2_off;
    RET_IF(tgemm(c, F32(b.tape->context, cd, 0), F32(wv, cd, 1), u, C, B, C, cd, false, s));
    if (float* G = engine_grad(c, a + "to_out.0.bias")) RET_IF(bwd_sum_rows_add(d, B, C, AT, G, 1, s));
    if (float* G = engine_grad(c, a + "to_out.0.weight")) RET_IF(bwd_outer_add(d, AT, u, C, G, C, C, B, s));
    RET_IF(tgemm(c, F32(d, AT, 0), F32(wo, C, 0), du, C, B, C, C, false, s));
    if (float* G = engine_grad(c, a + "to_v.weight")) RET_IF(bwd_outer_add(du, C, b.tape->context, cd, G, C, cd, B, s));
  }
  return 0;
}

// ---- mesh conditioner (SpatialVolumeNet) -------------------------------------------------------------------------------
// G[n] += sum over ALL rows of dy[row][n]: the rows are cut into up to 64 equal slabs ("pseudo samples") so that a one-sample
// volume of 49152 rows is summed by 64 x N/64 workgroups instead of N/64
int colsum_all(Bwd& b, const float* dy, long ldy, long rows, int N, float* G) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  int S = 64;
  while (S > 1 && rows % S) S >>= 1;
  float* part = ws_alloc<float>(c, (size_t)S * N);
  WS_CHECK(part);
  RET_IF(bwd_colsum_samples(dy, 1, ldy, S, (int)(rows / S), N, part, N, b.s));
  return bwd_sum_rows_add(part, S, N, N, G, 1, b.s);
}

// weight + bias gradient of a 3x3x3 conv (stride as in the forward): dy fp32 [rows_out][N] channels-last, x [B,D,H,W,K]
int wgrad_conv3d(Bwd& b, const ConvW& w, const float* dy, long ldy, const void* x, int x_f32, long ldx, int D, int H, int W, int K,
                 int stride) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const int rows = b.B * Do * Ho * Wo, Rp = up64(rows), N = w.N;
  if (float* G = engine_grad(c, w.key)) {
    half_t* dyT = ws_alloc<half_t>(c, (size_t)N * Rp);
    half_t* colT = ws_alloc<half_t>(c, (size_t)K * 27 * Rp);
    WS_CHECK(dyT && colT);
    RET_IF(bwd_tcast(dy, 1, ldy, rows, N, dyT, Rp, b.s));
    RET_IF(cbwd_im2colT3d(x, x_f32, ldx, b.B, D, H, W, K, stride, colT, Rp, b.s));
    RET_IF(wgrad_gemm(b, dyT, N, colT, K * 27, Rp, G));
  }
  if (float* Gb = w.bkey.empty() ? nullptr : engine_grad(c, w.bkey)) RET_IF(colsum_all(b, dy, ldy, (long)rows, N, Gb));
  return 0;
}
// ConvTranspose3d(k3, s2, p1, op1), weight [Cin][Cout][27]:  dW[ci][co*27 + k] = sum_i x[i][ci] dy[2 i - 1 + k][co] -- the strided
// conv's weight gradient with the roles swapped: dy (the fine volume [B,2D,2H,2W,Cout]) is the image, x [B,D,H,W,Cin] the "output
// gradient"
int wgrad_convT3d(Bwd& b, const ConvW& w, const float* dy, long ldy, const void* x, int x_f32, long ldx, int D, int H, int W) {
  mvd_ctx* c = b.c;
  WsScope scope(c, WS_TEMP);
  const int Cin = w.cin_l > 0 ? w.cin_l : w.Cin, Cout = w.N;
  const int rows = b.B * D * H * W, Rp = up64(rows);
  if (float* G = engine_grad(c, w.key)) {
    half_t* xT = ws_alloc<half_t>(c, (size_t)Cin * Rp);
    half_t* colT = ws_alloc<half_t>(c, (size_t)Cout * 27 * Rp);
    WS_CHECK(xT && colT);
    RET_IF(bwd_tcast(x, x_f32, ldx, rows, Cin, xT, Rp, b.s));
    RET_IF(cbwd_im2colT3d(dy, 1, ldy, b.B, 2 * D, 2 * H, 2 * W, Cout, 2, colT, Rp, b.s));
    RET_IF(wgrad_gemm(b, xT, Cin, colT, Cout * 27, Rp, G));
  }
  if (float* Gb = w.bkey.empty() ? nullptr : engine_grad(c, w.bkey)) RET_IF(colsum_all(b, dy, ldy, (long)rows * 8, Cout, Gb));
  return 0;
}
// adjoint of a 3x3x3 conv w.r.t. its input.  kind 0: stride-1 conv (the forward kernel on the flipped pack); 1: stride-2 conv
// (launched as a transposed conv from dy's coarse grid D,H,W); 2: transposed conv (launched as a stride-2 conv from dy's fine grid)
int dgrad_conv3d(Bwd& b, const ConvW& w, int kind, const half_t* dy16, float* dx, int lddx, int D, int H, int W, bool accum) {
  if (!w.wT || w.taps != 27) return mvd_fail("training backward: 3x3x3 adjoint weights missing");
  ConvW t;
  t.w = w.wT;
  t.N = w.cin_l > 0 ? w.cin_l : w.Cin;
  t.Cin = w.Np;
  t.taps = 27;
  GemmArgs g;
  g.a = dy16; g.lda = w.Np; g.w = &t; g.out = dx; g.ldc = lddx; g.use_bias = false;
  if (accum) {
    g.resid = dx;
    g.ldr = lddx;
  }
  if (kind == 1) return run_convT3d(b.c, g, b.B, D, H, W, b.s);
  return run_conv3d(b.c, g, b.B, D, H, W, kind == 2 ? 2 : 1, b.s);
}
// weight / bias gradient of a per-sample Linear (FiLM projections, step MLP): dW[N][K] += sum_r g[r][n] x[r][k], db += sum_r g
int lin_wgrad(mvd_ctx* c, const std::string& key, const float* g, long ldg, const float* x, long ldx, int rows, int N, int K, hipStream_t s) {
  if (float* G = engine_grad(c, key + ".weight")) RET_IF(bwd_outer_add(g, ldg, x, ldx, G, N, K, rows, s));
  if (float* G = engine_grad(c, key + ".bias")) RET_IF(bwd_sum_rows_add(g, rows, N, ldg, G, 1, s));
  return 0;
}

__global__ void iota_kernel(int* p, int n, int start) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = start + i;
}
__global__ void set_i64_kernel(int64_t* p, int64_t v) { *p = v; }

__global__ void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target, float k, size_t n, float* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = k * (pred[i] - target[i]);
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Closes gradient bucket number c->n_buckets behind what is enqueued on `s` so far.  prefixes: state_dict prefixes of the blocks
// whose backward just finished; their parameters are final EXCEPT the ones the stacked end-of-step kernels write (emb_layers.*
// through bwd_emb, attn2.* through bwd_attn2).  final: everything of model.diffusion_model.* not in a bucket yet.
static int bucket_close(mvd_ctx* c, const std::vector<std::string>& prefixes, bool final, hipStream_t s) {
  const int k = c->n_buckets;
  if ((int)c->buckets.size() <= k) c->buckets.resize(k + 1);
  mvd_ctx::GradBucket& b = c->buckets[k];
  if (!b.ev) HIP_CHECK_RET(hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
  if (!c->buckets_cached) {
    b.off.clear();
    b.len.clear();
    if (c->bucket_done.size() != c->params.size()) c->bucket_done.assign(c->params.size(), 0);
    static const std::string U = "model.diffusion_model.";
    for (size_t i = 0; i < c->params.size(); ++i) {
      if (c->bucket_done[i]) continue;
      const std::string& key = c->params[i].key;
      if (key.rfind(U, 0) != 0) continue;
      bool take = final;
      if (!take && key.find(".emb_layers.") == std::string::npos && key.find(".attn2.") == std::string::npos)
        for (const std::string& p : prefixes)
          if (!p.empty() && key.rfind(p, 0) == 0) {
            take = true;
            break;
          }
      if (!take) continue;
      c->bucket_done[i] = 1;
      const size_t off = c->params[i].off, len = (c->params[i].numel + 63) & ~(size_t)63;
      if (!b.off.empty() && b.off.back() + b.len.back() == off) b.len.back() += len;
      else {
        b.off.push_back(off);
        b.len.push_back(len);
      }
    }
  }
  if (c->bucket_snapshot)
    for (size_t r = 0; r < b.off.size(); ++r)
      HIP_CHECK_RET(hipMemcpyAsync(c->bucket_snapshot + b.off[r], c->arena_g + b.off[r], b.len[r] * sizeof(float), hipMemcpyDeviceToDevice, s));
  HIP_CHECK_RET(hipEventRecord(b.ev, s));
  ++c->n_buckets;
  return 0;
}
static std::string dot(const std::string& k) { return k.empty() || k.back() == '.' ? k : k + "."; }
static std::string strip_leaf(const std::string& k) {  // "a.b.weight" -> "a.b."
  const size_t p = k.rfind('.');
  return p == std::string::npos ? std::string() : k.substr(0, p + 1);
}

int engine_train_step(mvd_ctx* c, const float* x_nhwc, int x_ld, const int64_t* t, const float* context, int B, int depth0,
                      const Ctx5 src[4], const float* target_nhwc, float loss_scale, int recompute, float* pred_nhwc, float* loss_out,
                      float* const dsrc[4], hipStream_t s) {
  if (!c->finalized || !c->has_unet) return mvd_fail("UNet weights not uploaded / finalized");
  if (!c->train_mode) return mvd_fail("training step: enable training mode (mvd_train_enable) before mvd_finalize_weights");
  const mvd_unet_config& u = c->u;
  const int S = u.image_size, oc = u.out_channels, mc = u.model_channels;
  const size_t npred = (size_t)B * S * S * oc;
  WsScope scope(c);  // the whole step: tape + backward scratch
  TrainTape tape;
  tape.recompute = recompute != 0;
  // ---------------- forward ----------------
  c->ws.hold = recompute ? 3 : 2;
  int r = engine_unet(c, x_nhwc, x_ld, t, context, B, B, depth0, src, pred_nhwc, s, nullptr, &tape);
  c->ws.hold = 0;
  RET_IF(r);
  const size_t fwd_top = c->ws.off;
  (void)fwd_top;
  float* dpred = ws_alloc<float>(c, npred);
  WS_CHECK(dpred);
  if (loss_out) RET_IF(launch_mse(target_nhwc, pred_nhwc, npred, loss_out, s));
  {  // d mean((pred - target)^2) / d pred, times the loss scale
    const float k = 2.0f * loss_scale / (float)npred;
    const int blocks = (int)std::min<size_t>((npred + 255) / 256, 4096);
    hipLaunchKernelGGL(mse_grad_kernel, dim3(blocks), dim3(256), 0, s, pred_nhwc, target_nhwc, k, npred, dpred);
    HIP_CHECK_RET(hipGetLastError());
  }
  // ---------------- backward ----------------
  c->n_buckets = 0;
  Fwd f{c, s, B, B, depth0, tape.ea, context, tape.a2, src, {nullptr, nullptr, nullptr, nullptr}};
  f.train = true;
  for (int l = 0; l < 4; ++l) f.src16[l] = nullptr;
  Bwd b{c, s, B, &f, &tape};
  for (int l = 0; l < 4; ++l) b.dsrc[l] = dsrc ? dsrc[l] : nullptr;
  b.demb = ws_alloc<float>(c, (size_t)B * c->emb_total);
  b.da2 = ws_alloc<float>(c, (size_t)B * c->a2_total);
  WS_CHECK(b.demb && b.da2);
  HIP_CHECK_RET(hipMemsetAsync(b.demb, 0, (size_t)B * c->emb_total * sizeof(float), s));
  HIP_CHECK_RET(hipMemsetAsync(b.da2, 0, (size_t)B * c->a2_total * sizeof(float), s));
  const int nb = (int)c->in_blocks.size();
  // gradient buffers shaped like the concat buffers (see engine_unet: block j's output lives in the skip slice of cat[nb-1-j])
  std::vector<float*> dcat(nb);
  for (int i = 0; i < nb; ++i) {
    const int res = tape.in_res[nb - 1 - i];
    dcat[i] = ws_alloc<float>(c, (size_t)B * res * res * tape.cat_C[i]);
    WS_CHECK(dcat[i]);
  }
  float* d_final = ws_alloc<float>(c, (size_t)B * S * S * mc);
  WS_CHECK(d_final);
  RET_IF(bwd_head(b, dpred, d_final));
  // stages grouped by chain, in forward order
  std::vector<std::vector<int>> chains(2 * nb + 1);
  for (int i = 0; i < (int)tape.stages.size(); ++i) chains[tape.stages[i].chain].push_back(i);
  // a fp16 view of the context volumes is only needed when blocks are re-run (recompute): engine_unet made one per forward
  auto run_chain_bwd = [&](int chain, View d_out, View d_in, bool accum_in, bool need_din) -> int {
    WsScope ch_scope(c, WS_CHAIN);
    const std::vector<int>& ids = chains[chain];
    View g_out = d_out;
    for (int k = (int)ids.size() - 1; k >= 0; --k) {
      StageRec& st = tape.stages[ids[k]];
      const bool first = k == 0;
      View g_in;
      bool acc = false;
      if (first) {
        g_in = d_in;
        acc = accum_in;
      } else {
        g_in.p = ws_alloc<float>(c, (size_t)B * st.H * st.W * st.in.C);
        WS_CHECK(g_in.p);
        g_in.ld = st.in.C;
        g_in.C = st.in.C;
      }
      const size_t mark = c->ws.off;
      if (tape.recompute && (st.kind == OP_RES || st.kind == OP_ST)) {  // re-run the block, keeping its intermediates
        c->ws.hold = 2;
        int H = st.H, W = st.W;
        // the re-run writes the block's output again (same values): into scratch, the taped output may feed other readers
        View tmp_out;
        tmp_out.p = ws_alloc<float>(c, (size_t)B * st.H * st.W * st.out.C);
        tmp_out.ld = st.out.C;
        tmp_out.C = st.out.C;
        int rr = tmp_out.p ? 0 : mvd_fail("workspace exhausted: create the context with a larger workspace_bytes");
        if (!rr) {
          UOp op{st.kind, st.idx, st.in.C, st.out.C};
          rr = unet_do_op(f, op, st.in, tmp_out, H, W, &st);
        }
        c->ws.hold = 0;
        if (rr) {
          c->ws.off = mark;
          return rr;
        }
      }
      int rr = 0;
      switch (st.kind) {
        case OP_RES: rr = bwd_res(b, c->res[st.idx], st.rs, st.in, g_out, g_in, acc, st.H, st.W); break;
        case OP_ST: rr = bwd_st(b, c->st[st.idx], st.ss, st.in, g_out, g_in, acc, st.H, st.W); break;
        case OP_COND: rr = bwd_cond(b, c->conds[st.idx], st.in, g_out, g_in, acc, st.H, st.W, st.level); break;
        default: rr = bwd_conv(b, st.kind, c->convs[st.idx], st.in, g_out, g_in, acc, !(first && !need_din), st.H, st.W); break;
      }
      c->ws.off = mark;  // the re-run's intermediates (and nothing else: the block functions release their own scratch)
      RET_IF(rr);
      g_out = g_in;
    }
    {  // the chain's parameter gradients are final: one bucket
      std::vector<std::string> pre;
      if (!c->buckets_cached)
        for (int id : ids) {
          const StageRec& st = tape.stages[id];
          switch (st.kind) {
            case OP_RES: pre.push_back(dot(c->res[st.idx].key)); break;
            case OP_ST: pre.push_back(dot(c->st[st.idx].key)); break;
            case OP_COND: pre.push_back(dot(c->conds[st.idx].key)); break;
            default: pre.push_back(strip_leaf(c->convs[st.idx].key)); break;
          }
        }
      RET_IF(bucket_close(c, pre, false, s));
    }
    return 0;
  };
  auto skip_view = [&](int i) {  // gradient of input block (nb-1-i)'s output
    View v;
    v.p = dcat[i] + tape.h_ch[i];
    v.ld = tape.cat_C[i];
    v.C = tape.in_ch[nb - 1 - i];
    return v;
  };
  for (int i = nb - 1; i >= 0; --i) {  // output blocks
    View d_out;
    if (i + 1 < nb) {
      d_out.p = dcat[i + 1];
      d_out.ld = tape.cat_C[i + 1];
      d_out.C = tape.h_ch[i + 1];
    } else {
      d_out.p = d_final;
      d_out.ld = mc;
      d_out.C = mc;
    }
    View d_in;
    d_in.p = dcat[i];
    d_in.ld = tape.cat_C[i];
    d_in.C = tape.cat_C[i];
    RET_IF(run_chain_bwd(nb + 1 + i, d_out, d_in, false, true));
  }
  {  // middle block: reads input block nb-1's output (skip slice of cat[0]), whose gradient already holds output block 0's share
    View d_out;
    d_out.p = dcat[0];
    d_out.ld = tape.cat_C[0];
    d_out.C = tape.h_ch[0];
    RET_IF(run_chain_bwd(nb, d_out, skip_view(0), true, true));
  }
  for (int j = nb - 1; j >= 0; --j) {  // input blocks
    View d_out = skip_view(nb - 1 - j);
    View d_in;
    if (j > 0) d_in = skip_view(nb - j);
    RET_IF(run_chain_bwd(j, d_out, d_in, true, j > 0));
  }
  RET_IF(bwd_emb(b));
  RET_IF(bwd_attn2(b));
  RET_IF(bucket_close(c, {}, true, s));  // the stacked projections, the head, whatever no chain owns
  c->buckets_cached = true;
  return 0;
}

// One DepthTransformer's backward on its own (parity hook): x / dout / dx channels-last [B*H*W][dim], ctx_vol / dctx
// channels-last [B*D*H*W][Cc] with D = depth0 >> level.  Parameter gradients are accumulated into the arena.
int engine_train_cond_backward(mvd_ctx* c, int cond_idx, const float* x, const float* ctx_vol, const float* dout, int B, int H, int W,
                               int level, int depth0, float* dx, float* dctx, hipStream_t s) {
  if (!c->finalized || !c->train_mode) return mvd_fail("cond backward: context not finalized in training mode");
  if (cond_idx < 0 || cond_idx >= (int)c->conds.size() || level < 0 || level > 3) return mvd_fail("cond backward: bad index");
  const CondW& cd = c->conds[cond_idx];
  TrainTape tape;
  tape.depth0 = depth0;
  Ctx5 src[4];
  src[level].p = ctx_vol;
  src[level].f32 = 1;
  tape.src = src;
  Fwd f{c, s, B, B, depth0, nullptr, nullptr, nullptr, src, {nullptr, nullptr, nullptr, nullptr}};
  Bwd b{c, s, B, &f, &tape};
  b.dsrc[level] = dctx;
  View in, g_out, g_in;
  in.p = const_cast<float*>(x); in.ld = cd.dim; in.C = cd.dim;
  g_out.p = const_cast<float*>(dout); g_out.ld = cd.dim; g_out.C = cd.dim;
  g_in.p = dx; g_in.ld = cd.dim; g_in.C = cd.dim;
  return bwd_cond(b, cd, in, g_out, g_in, false, H, W, level);
}

// ---------------------------------------------------------------------------------------------------------------------
namespace {
// what the per-sample stages of the conditioner's backward hand to each other (all pointers into the call's workspace scope)
struct CondSample {
  int* vidx;
  float* vf;
  const float* sp_in[9];
  float *sp_raw[9], *sp_post[9], *sp_stats[9];
  const int* sp_nbr[9];
  int sp_nout[9], sp_nin[9];
};
}  // namespace

int engine_select_sample(mvd_ctx* c, int slot);

// Backward of the conditioner for B samples whose tables sit in slots[0..B): the per-sample stages (2-D encoder, gathers, sparse
// CNN: different meshes, cameras, BatchNorm statistics) run sample by sample, the frustum network between them -- the same
// weights on same-shaped volumes -- runs ONCE with the samples as its batch (its 8^3 / 4^3 levels are 96- and 768-row GEMMs per
// sample).  x_noisy [B][N,4,s,s], v_embed [B][N,vd], dsrc[l] [B][vox_l][C_l] channels-last (accumulated in place).
int engine_train_conditioner_backward_batch(mvd_ctx* c, int B, const int* slots, const float* x_noisy_all, const int64_t* timesteps,
                                            const float* v_embed_all, int n_views, const int* target_idx, float* const dsrc[4],
                                            float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, hipStream_t s) {
  if (!c->finalized || !c->train_mode) return mvd_fail("conditioner backward: context not finalized in training mode");
  if (!c->has_cond || !c->has_step) return mvd_fail("conditioner backward: spatial_volume / time_embed weights not uploaded");
  if (B < 1 || (B > 1 && (dbg_dvolume || dbg_dfused || dbg_dfeats || dbg_dtembed))) return mvd_fail("conditioner backward: bad batch arguments");
  for (int bi = 0; bi < B; ++bi)
    if (n_views != c->v.num_views || target_idx[bi] < 0 || target_idx[bi] >= n_views) return mvd_fail("conditioner backward: bad view arguments");
  for (int l = 0; l < 4; ++l)
    if (!dsrc[l]) return mvd_fail("conditioner backward: dL/d(frustum volume) of every level is required");
  WsScope scope(c);
  // MVD_COND_BWD_TIMING=1: host-side enqueue time of each phase on stderr (development aid)
  static const bool host_timing = getenv("MVD_COND_BWD_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_last = now();
  auto mark = [&](const char* what) {
    if (!host_timing) return;
    const double t = now();
    fprintf(stderr, "[cond-bwd host] %-28s %7.3f ms\n", what, t - t_last);
    t_last = t;
  };
  const int N = n_views, S = c->u.image_size, HW = S * S, rows = N * HW, td = c->v.time_dim, vd = c->v.view_dim;
  const int V = c->v.spatial_volume_size, persp = c->v.projection == 0;
  const std::string SV = "spatial_volume.", FV = SV + "frustum_volume_feats.";
  auto F = [&](size_t n) { return ws_alloc<float>(c, n); };
  auto H16 = [&](size_t n) { return ws_alloc<half_t>(c, n); };
  const int* fd = c->v.frustum_dims;
  int Dl[4], Sl[4];
  size_t vox[4];
  for (int l = 0; l < 4; ++l) {
    Dl[l] = l ? (Dl[l - 1] - 1) / 2 + 1 : c->v.frustum_volume_depth;
    Sl[l] = l ? (Sl[l - 1] - 1) / 2 + 1 : c->v.input_image_size / 8;
    vox[l] = (size_t)Dl[l] * Sl[l] * Sl[l];
  }
  const int FT = c->film_total;
  struct SlotGuard {  // the active slot is the caller's again on every exit path
    mvd_ctx* c;
    int slot;
    ~SlotGuard() { engine_select_sample(c, slot); }
  } slot_guard{c, c->cur_slot};
  // extended-precision packs of the 2-D encoder's convs, once per call
  auto xp_pack = [&](const ConvW& w, int cin_src, ConvW* o) -> int {
    const float* mw = engine_master(c, w.key);
    if (!mw) return mvd_fail("conditioner backward: encoder master weights missing");
    *o = w;
    const int Cl = w.cin_l > 0 ? w.cin_l : w.Cin;
    o->xp = 1;
    o->cin_l = Cl;
    o->Cin = 3 * Cl;
    o->wT = nullptr;
    o->w = H16((size_t)w.taps * w.N * 3 * Cl);
    WS_CHECK(o->w);
    return launch_pack_weight(mw, w.N, 3 * Cl, w.taps, 0, 0, o->w, s, cin_src, 1);
  };
  ConvW x_init, x_c1[3], x_c2[3], x_final;
  RET_IF(xp_pack(c->enc_init, 4, &x_init));
  for (int i = 0; i < 3; ++i) {
    RET_IF(xp_pack(c->enc_blocks[i].c1, 16, &x_c1[i]));
    RET_IF(xp_pack(c->enc_blocks[i].c2, 16, &x_c2[i]));
  }
  RET_IF(xp_pack(c->enc_final, 16, &x_final));
  // what crosses the stages, for all samples
  float *t_emb_all = F((size_t)B * td), *d_temb_all = F((size_t)B * td), *vt_all = F((size_t)B * vd), *pre_f_all = F((size_t)B * FT);
  half_t* gath_all = H16((size_t)B * vox[0] * 64);
  float* d_gath_all = F((size_t)B * vox[0] * 64);
  float *feats_all = F((size_t)B * rows * 16), *d_feats_all = F((size_t)B * rows * 16);
  WS_CHECK(t_emb_all && d_temb_all && vt_all && pre_f_all && gath_all && d_gath_all && feats_all && d_feats_all);
  HIP_CHECK_RET(hipMemsetAsync(d_feats_all, 0, (size_t)B * rows * 16 * sizeof(float), s));
  HIP_CHECK_RET(hipMemsetAsync(d_temb_all, 0, (size_t)B * td * sizeof(float), s));
  std::vector<CondSample> st(B);

  // ---------------- stage 0, all samples at once: step embedding MLP and the 2-D encoder (B * N views as its batch) ----------------
  // step embedding (morphable_diffusion.py:491-494): t_emb = W2 silu(W0 temb(t) + b0) + b2
  const int BN = B * N;
  const size_t rows_all = (size_t)BN * HW;
  int64_t* t_dev = (int64_t*)c->ws.alloc(sizeof(int64_t) * B);
  float *e0 = F((size_t)B * td), *u1 = F((size_t)B * td), *e1 = F((size_t)B * td);
  WS_CHECK(t_dev && e0 && u1 && e1);
  for (int bi = 0; bi < B; ++bi) hipLaunchKernelGGL(set_i64_kernel, dim3(1), dim3(1), 0, s, t_dev + bi, timesteps[bi]);
  HIP_CHECK_RET(hipGetLastError());
  RET_IF(launch_timestep_embedding(t_dev, B, td, e0, s));
  RET_IF(launch_small_linear(e0, td, B, td, c->step_te0.w, c->step_te0.bias, td, ACT_NONE, u1, td, 0, s));
  RET_IF(bwd_silu_fwd(u1, e1, (size_t)B * td, s));
  RET_IF(launch_small_linear(e1, td, B, td, c->step_te2.w, c->step_te2.bias, td, ACT_NONE, t_emb_all, td, 0, s));
  // 2-D encoder (NoisyTargetViewEncoder, network.py:181-207), layer by layer and in EXTENDED precision (hi/lo operand split on
  // packs made here from the master weights; 16 channels: the cost is nothing): the sparse CNN behind it has nine BatchNorm +
  // ReLU layers whose masks are re-derived from these features -- an fp16-rounded encoder moves the gradients upstream of them
  // by 4-10e-2 (measured), the extended-precision one by 1e-3.  No split-K: a view's features do not depend on the batch.
  float *x8 = F(rows_all * 8), *pre_e = F((size_t)BN * 48);
  float* cur_e[4];
  float* r1_e[3];
  half_t *a1_e[3], *a2_e[3], *af = H16(rows_all * 48), *x8s = H16(rows_all * 24);
  for (int i = 0; i < 4; ++i) cur_e[i] = F(rows_all * 16);
  for (int i = 0; i < 3; ++i) {
    r1_e[i] = F(rows_all * 16);
    a1_e[i] = H16(rows_all * 48);
    a2_e[i] = H16(rows_all * 48);
    WS_CHECK(r1_e[i] && a1_e[i] && a2_e[i]);
  }
  WS_CHECK(x8 && pre_e && af && x8s && cur_e[0] && cur_e[1] && cur_e[2] && cur_e[3]);
  for (int bi = 0; bi < B; ++bi)  // pre[b, v] = time_embed_i(t_emb[b]) + view_embed_i(v_embed[b, v]), the three blocks side by side
    RET_IF(launch_small_linear(t_emb_all + (size_t)bi * td, td, -N, td, c->enc_t.w, c->enc_t.bias, 48, ACT_NONE, pre_e + (size_t)bi * N * 48, 48, 0, s));
  RET_IF(launch_small_linear(v_embed_all, vd, BN, vd, c->enc_v.w, c->enc_v.bias, 48, ACT_NONE, pre_e, 48, 1, s));
  RET_IF(launch_nchw_to_nhwc(x_noisy_all, BN, 4, HW, x8, 8, 8, s));
  RET_IF(launch_rows_f32_to_f16_split(x8, 8, (long)rows_all, 8, x8s, s));
  {
    GemmArgs g;
    g.a = x8s; g.lda = 24; g.w = &x_init; g.out = cur_e[0]; g.ldc = 16; g.force_splitk = 1;
    RET_IF(run_conv2d(c, g, BN, S, S, 1, 0, s));
    for (int i = 0; i < 3; ++i) {
      const EncBlockW& e = c->enc_blocks[i];
      RET_IF(run_group_norm(c, cur_e[i], 16, BN, HW, e.n1, 8, 1e-5f, ACT_SILU, pre_e + 16 * i, a1_e[i], 48, s, 48, 1));
      g = GemmArgs();
      g.a = a1_e[i]; g.lda = 48; g.w = &x_c1[i]; g.out = r1_e[i]; g.ldc = 16; g.force_splitk = 1;
      RET_IF(run_conv2d(c, g, BN, S, S, 1, 0, s));
      RET_IF(run_group_norm(c, r1_e[i], 16, BN, HW, e.n2, 8, 1e-5f, ACT_SILU, nullptr, a2_e[i], 48, s, 0, 1));
      g = GemmArgs();
      g.a = a2_e[i]; g.lda = 48; g.w = &x_c2[i]; g.out = cur_e[i + 1]; g.ldc = 16; g.resid = cur_e[i]; g.ldr = 16; g.force_splitk = 1;
      RET_IF(run_conv2d(c, g, BN, S, S, 1, 0, s));
    }
    RET_IF(run_group_norm(c, cur_e[3], 16, BN, HW, c->enc_final_norm, 8, 1e-5f, ACT_SILU, nullptr, af, 48, s, 0, 1));
    g = GemmArgs();
    g.a = af; g.lda = 48; g.w = &x_final; g.out = feats_all; g.ldc = 16; g.force_splitk = 1;
    RET_IF(run_conv2d(c, g, BN, S, S, 1, 0, s));
  }
  mark("fwd: step mlp + 2-D encoder");

  // ---------------- stage 1, per sample: forward up to the gathered frustum features ----------------
  auto stage1 = [&](int bi) -> int {
  if (!c->mesh.Nv || !c->cams) return mvd_fail("mvd_set_mesh / mvd_set_cameras must be called first");
  MeshTables& m = c->mesh;
  const int Nv = m.Nv;
  const float* v_embed = v_embed_all + (size_t)bi * N * vd;
  int* vidx = (int*)c->ws.alloc(sizeof(int) * (N + 1));
  WS_CHECK(vidx);
  hipLaunchKernelGGL(iota_kernel, dim3(1), dim3(64), 0, s, vidx, N, 0);
  hipLaunchKernelGGL(iota_kernel, dim3(1), dim3(64), 0, s, vidx + N, 1, target_idx[bi]);
  HIP_CHECK_RET(hipGetLastError());
  const float *t_emb = t_emb_all + (size_t)bi * td, *feats = feats_all + (size_t)bi * rows * 16;
  // vertex features, view fusion
  float *vf = F((size_t)N * Nv * 16), *fused = F((size_t)Nv * 16);
  WS_CHECK(vf && fused);
  RET_IF(launch_vertex_gather(feats, c->cams, vidx, N, m.verts, Nv, V, c->v.spatial_volume_length, S, persp, vf, s));
  RET_IF(launch_fuse_views(vf, N, Nv, N, c->fuse_w, c->fuse_b, fused, 0, s));
  // sparse voxel CNN, train mode: raw conv output and post-activation rows of every layer
  const float* sp_in[9];
  float *sp_raw[9], *sp_post[9], *sp_stats[9];  // raw conv output, post-activation rows, BatchNorm [mean | rstd]
  const int* sp_nbr[9];
  int sp_nout[9], sp_nin[9];
  {
    const float* in = fused;
    int lvl = 0, n_in = Nv;
    for (int i = 0; i < 9; ++i) {
      const SparseLayerW& L = c->sparse[i];
      if (L.strided) {
        sp_nbr[i] = m.nbr_down[lvl];
        ++lvl;
      } else {
        sp_nbr[i] = m.nbr_subm[lvl];
      }
      sp_nout[i] = m.n_sites[lvl];
      sp_nin[i] = n_in;
      sp_in[i] = in;
      sp_raw[i] = F((size_t)sp_nout[i] * L.cout);
      sp_post[i] = F((size_t)sp_nout[i] * L.cout);
      WS_CHECK(sp_raw[i] && sp_post[i]);
      sp_stats[i] = F((size_t)2 * L.cout);
      WS_CHECK(sp_stats[i]);
      RET_IF(launch_sparse_conv(in, sp_nbr[i], sp_nout[i], L.cin, L.cout, L.w, L.wp, nullptr, nullptr, sp_raw[i], s));
      RET_IF(launch_bn_rows_relu(sp_raw[i], sp_post[i], sp_nout[i], L.cout, L.gamma, L.beta, 1e-3f, sp_stats[i], s));
      in = sp_post[i];
      n_in = sp_nout[i];
    }
  }
  mark("fwd: gather, fuse, sparse net");
  float* volume = F((size_t)V * V * V * 64);
  WS_CHECK(volume);
  RET_IF(launch_latent_gather(sp_post[8], m.grid2, m.shape[2][0], m.shape[2][1], m.shape[2][2], m.min_xyz, m.out_sh, c->v.voxel_size, V,
                              c->v.spatial_volume_length, volume, s));
  RET_IF(launch_frustum_gather(volume, c->cams, vidx + N, 1, Dl[0], Sl[0], V, c->v.spatial_volume_length, persp,
                               gath_all + (size_t)bi * vox[0] * 64, s));
  float* pre_row = pre_f_all + (size_t)bi * FT;
  RET_IF(launch_small_linear(t_emb, td, 1, td, c->film_t.w, c->film_t.bias, FT, ACT_NONE, pre_row, FT, 0, s));
  RET_IF(launch_small_linear(v_embed + (size_t)target_idx[bi] * vd, vd, 1, vd, c->film_v.w, c->film_v.bias, FT, ACT_NONE, pre_row, FT, 1, s));
  HIP_CHECK_RET(hipMemcpyAsync(vt_all + (size_t)bi * vd, v_embed + (size_t)target_idx[bi] * vd, vd * sizeof(float), hipMemcpyDeviceToDevice, s));
  CondSample& P = st[bi];
  P.vidx = vidx, P.vf = vf;
  for (int i = 0; i < 9; ++i)
    P.sp_in[i] = sp_in[i], P.sp_raw[i] = sp_raw[i], P.sp_post[i] = sp_post[i], P.sp_stats[i] = sp_stats[i], P.sp_nbr[i] = sp_nbr[i],
    P.sp_nout[i] = sp_nout[i], P.sp_nin[i] = sp_nin[i];
  return 0;
  };
  for (int bi = 0; bi < B; ++bi) {
    RET_IF(engine_select_sample(c, slots[bi]));
    RET_IF(stage1(bi));
  }

  // ---------------- stage 2: FrustumTV3DNet forward + backward, the samples as its batch (chunks of at most 16: 32-bit operand
  // offsets of the level-0 weight-gradient GEMM) ----------------
  for (int b0 = 0; b0 < B; b0 += 16) {
  const size_t Bc = (size_t)std::min(16, B - b0);
  WsScope chunk_scope(c, WS_BLOCK);
  half_t* gath = gath_all + (size_t)b0 * vox[0] * 64;
  float* pre_f = pre_f_all + (size_t)b0 * FT;
  GemmArgs g;
  float *xd[4], *xf[4], *tmp_f[3];
  half_t *a1_f[3], *a2_f[3], *au_f[3];
  for (int l = 0; l < 4; ++l) {
    xd[l] = F(Bc * vox[l] * fd[l]);
    xf[l] = F(Bc * vox[l] * fd[l]);
    WS_CHECK(xd[l] && xf[l]);
  }
  for (int l = 0; l < 3; ++l) {
    tmp_f[l] = F(Bc * vox[l + 1] * fd[l + 1]);
    a1_f[l] = H16(Bc * vox[l] * fd[l]);
    a2_f[l] = H16(Bc * vox[l + 1] * fd[l + 1]);
    au_f[l] = H16(Bc * vox[l + 1] * fd[l + 1]);
    WS_CHECK(tmp_f[l] && a1_f[l] && a2_f[l] && au_f[l]);
  }
  g = GemmArgs();
  g.a = gath; g.lda = 64; g.w = &c->fr_conv0; g.out = xd[0]; g.ldc = fd[0];
  RET_IF(run_conv3d(c, g, (int)Bc, Dl[0], Sl[0], Sl[0], 1, s));
  for (int l = 0; l < 3; ++l) {
    const FrustumBlockW& b1 = c->fr_blocks[2 * l];
    const FrustumBlockW& b2 = c->fr_blocks[2 * l + 1];
    RET_IF(run_group_norm(c, xd[l], fd[l], (int)Bc, (int)vox[l], b1.gn, 8, 1e-5f, ACT_SILU, pre_f + c->film_off[2 * l], a1_f[l], fd[l], s, FT));
    g = GemmArgs();
    g.a = a1_f[l]; g.lda = fd[l]; g.w = &b1.conv; g.out = tmp_f[l]; g.ldc = fd[l + 1];
    RET_IF(run_conv3d(c, g, (int)Bc, Dl[l], Sl[l], Sl[l], 2, s));
    RET_IF(run_group_norm(c, tmp_f[l], fd[l + 1], (int)Bc, (int)vox[l + 1], b2.gn, 8, 1e-5f, ACT_SILU, pre_f + c->film_off[2 * l + 1], a2_f[l],
                          fd[l + 1], s, FT));
    g = GemmArgs();
    g.a = a2_f[l]; g.lda = fd[l + 1]; g.w = &b2.conv; g.out = xd[l + 1]; g.ldc = fd[l + 1];
    RET_IF(run_conv3d(c, g, (int)Bc, Dl[l + 1], Sl[l + 1], Sl[l + 1], 1, s));
  }
  HIP_CHECK_RET(hipMemcpyAsync(xf[3], xd[3], Bc * vox[3] * fd[3] * sizeof(float), hipMemcpyDeviceToDevice, s));
  for (int l = 2; l >= 0; --l) {
    const FrustumBlockW& u = c->fr_up[2 - l];
    RET_IF(run_group_norm(c, xf[l + 1], fd[l + 1], (int)Bc, (int)vox[l + 1], u.gn, 8, 1e-5f, ACT_SILU, pre_f + c->film_off[6 + (2 - l)], au_f[l],
                          fd[l + 1], s, FT));
    g = GemmArgs();
    g.a = au_f[l]; g.lda = fd[l + 1]; g.w = &u.conv; g.out = xf[l]; g.ldc = fd[l]; g.resid = xd[l]; g.ldr = fd[l];
    RET_IF(run_convT3d(c, g, (int)Bc, Dl[l + 1], Sl[l + 1], Sl[l + 1], s));
  }
  mark("fwd: frustum net");
  Fwd f{c, s, (int)Bc, (int)Bc, 0, nullptr, nullptr, nullptr, nullptr, {nullptr, nullptr, nullptr, nullptr}};
  TrainTape tape;
  Bwd b{c, s, (int)Bc, &f, &tape};
  float* d_pre_f = F(Bc * FT);
  WS_CHECK(d_pre_f);
  HIP_CHECK_RET(hipMemsetAsync(d_pre_f, 0, Bc * FT * sizeof(float), s));
  half_t* dy16;
  float* gl[4];
  for (int l = 0; l < 4; ++l) gl[l] = dsrc[l] + (size_t)b0 * vox[l] * fd[l];  // dL/d x_l, accumulated in place
  // up path (forward order l = 2, 1, 0): x_l = xd_l + convT(silu(GN(xf_{l+1} + film)))
  for (int l = 0; l <= 2; ++l) {
    WsScope sc(c, WS_BLOCK);
    const FrustumBlockW& u = c->fr_up[2 - l];
    float* d_au = F(Bc * vox[l + 1] * fd[l + 1]);
    WS_CHECK(d_au);
    RET_IF(grad16(c, gl[l], fd[l], (long)(Bc * vox[l]), fd[l], &dy16, s));
    RET_IF(dgrad_conv3d(b, u.conv, 2, dy16, d_au, fd[l + 1], Dl[l], Sl[l], Sl[l], false));
    RET_IF(wgrad_convT3d(b, u.conv, gl[l], fd[l], au_f[l], 0, fd[l + 1], Dl[l + 1], Sl[l + 1], Sl[l + 1]));
    RET_IF(gn_backward(b, u.gn, 8, 1e-5f, ACT_SILU, xf[l + 1], fd[l + 1], d_au, fd[l + 1], (int)vox[l + 1], gl[l + 1], fd[l + 1], true,
                       pre_f + c->film_off[6 + (2 - l)], FT, d_pre_f + c->film_off[6 + (2 - l)], FT));
  }
  mark("bwd: frustum up path");
  // down path (forward order l = 0, 1, 2)
  for (int l = 2; l >= 0; --l) {
    WsScope sc(c, WS_BLOCK);
    const FrustumBlockW& b1 = c->fr_blocks[2 * l];
    const FrustumBlockW& b2 = c->fr_blocks[2 * l + 1];
    float *d_a2 = F(Bc * vox[l + 1] * fd[l + 1]), *d_tmp = F(Bc * vox[l + 1] * fd[l + 1]), *d_a1 = F(Bc * vox[l] * fd[l]);
    WS_CHECK(d_a2 && d_tmp && d_a1);
    RET_IF(grad16(c, gl[l + 1], fd[l + 1], (long)(Bc * vox[l + 1]), fd[l + 1], &dy16, s));
    RET_IF(dgrad_conv3d(b, b2.conv, 0, dy16, d_a2, fd[l + 1], Dl[l + 1], Sl[l + 1], Sl[l + 1], false));
    RET_IF(wgrad_conv3d(b, b2.conv, gl[l + 1], fd[l + 1], a2_f[l], 0, fd[l + 1], Dl[l + 1], Sl[l + 1], Sl[l + 1], fd[l + 1], 1));
    RET_IF(gn_backward(b, b2.gn, 8, 1e-5f, ACT_SILU, tmp_f[l], fd[l + 1], d_a2, fd[l + 1], (int)vox[l + 1], d_tmp, fd[l + 1], false,
                       pre_f + c->film_off[2 * l + 1], FT, d_pre_f + c->film_off[2 * l + 1], FT));
    RET_IF(grad16(c, d_tmp, fd[l + 1], (long)(Bc * vox[l + 1]), fd[l + 1], &dy16, s));
    RET_IF(dgrad_conv3d(b, b1.conv, 1, dy16, d_a1, fd[l], Dl[l + 1], Sl[l + 1], Sl[l + 1], false));
    RET_IF(wgrad_conv3d(b, b1.conv, d_tmp, fd[l + 1], a1_f[l], 0, fd[l], Dl[l], Sl[l], Sl[l], fd[l], 2));
    RET_IF(gn_backward(b, b1.gn, 8, 1e-5f, ACT_SILU, xd[l], fd[l], d_a1, fd[l], (int)vox[l], gl[l], fd[l], true,
                       pre_f + c->film_off[2 * l], FT, d_pre_f + c->film_off[2 * l], FT));
  }
  mark("bwd: frustum down path");
  // conv0 on the gathered frustum features
  float* d_gath = d_gath_all + (size_t)b0 * vox[0] * 64;
  RET_IF(grad16(c, gl[0], fd[0], (long)(Bc * vox[0]), fd[0], &dy16, s));
  RET_IF(dgrad_conv3d(b, c->fr_conv0, 0, dy16, d_gath, 64, Dl[0], Sl[0], Sl[0], false));
  RET_IF(wgrad_conv3d(b, c->fr_conv0, gl[0], fd[0], gath, 0, 64, Dl[0], Sl[0], Sl[0], 64, 1));
  // FiLM projections of the nine frustum blocks: film = t_conv(t_emb) + v_conv(v_embed[target])
  for (int i = 0; i < 9; ++i) {
    const FrustumBlockW& fb = i < 6 ? c->fr_blocks[i] : c->fr_up[i - 6];
    const float* dp = d_pre_f + c->film_off[i];
    RET_IF(lin_wgrad(c, fb.t_conv.key, dp, FT, t_emb_all + (size_t)b0 * td, td, (int)Bc, fb.cin, td, s));
    RET_IF(lin_wgrad(c, fb.v_conv.key, dp, FT, vt_all + (size_t)b0 * vd, vd, (int)Bc, fb.cin, vd, s));
  }
  RET_IF(cbwd_small_linear_bwd(d_pre_f, FT, (int)Bc, FT, c->film_t.w, td, d_temb_all + (size_t)b0 * td, td, 1, s));
  mark("bwd: frustum net, conv0 + FiLM");
  }

  // ---------------- stage 3, per sample: scatters, sparse CNN, view fusion, 2-D encoder, step MLP ----------------
  auto stage3 = [&](int bi) -> int {
  WsScope sample_scope(c, WS_BLOCK);
  MeshTables& m = c->mesh;
  const int Nv = m.Nv;
  CondSample& P = st[bi];
  int* vidx = P.vidx;
  float* vf = P.vf;
  const float** sp_in = P.sp_in;
  float **sp_raw = P.sp_raw, **sp_stats = P.sp_stats;
  const int** sp_nbr = P.sp_nbr;
  int *sp_nout = P.sp_nout, *sp_nin = P.sp_nin;
  // frustum gather, latent-code gather: scatter adjoints
  float* d_vol = F((size_t)V * V * V * 64);
  float* d_cur = F((size_t)sp_nout[8] * 64);
  WS_CHECK(d_vol && d_cur);
  HIP_CHECK_RET(hipMemsetAsync(d_vol, 0, (size_t)V * V * V * 64 * sizeof(float), s));
  HIP_CHECK_RET(hipMemsetAsync(d_cur, 0, (size_t)sp_nout[8] * 64 * sizeof(float), s));
  RET_IF(cbwd_frustum_scatter(d_gath_all + (size_t)bi * vox[0] * 64, c->cams, vidx + N, 1, Dl[0], Sl[0], V, c->v.spatial_volume_length, persp, d_vol, s));
  if (dbg_dvolume) RET_IF(launch_nhwc_to_nchw(d_vol, 64, 1, 64, V * V * V, dbg_dvolume, s));
  RET_IF(cbwd_latent_scatter(d_vol, m.grid2, m.shape[2][0], m.shape[2][1], m.shape[2][2], m.min_xyz, m.out_sh, c->v.voxel_size, V,
                             c->v.spatial_volume_length, d_cur, s));
  mark("bwd: scatters");
  // sparse voxel CNN
  for (int i = 8; i >= 0; --i) {
    const SparseLayerW& L = c->sparse[i];
    float* Gg = engine_grad(c, L.bnkey + ".weight");
    float* Gb = engine_grad(c, L.bnkey + ".bias");
    float* Gw = engine_grad(c, L.wkey);
    if (!Gg || !Gb || !Gw) return mvd_fail("conditioner backward: sparse layer parameters missing from the arena");
    {
      WsScope sc(c, WS_TEMP);
      float* bn_scr = F((size_t)cbwd_bn_scratch_floats(sp_nout[i], L.cout));
      WS_CHECK(bn_scr);
      RET_IF(cbwd_bn_rows_relu(sp_raw[i], d_cur, sp_nout[i], L.cout, L.gamma, L.beta, sp_stats[i], bn_scr, Gg, Gb, s));
    }
    float* dwp = F((size_t)27 * L.cin * L.cout);
    float* d_in = F((size_t)sp_nin[i] * L.cin);
    WS_CHECK(dwp && d_in);
    if (L.wp && L.wd) {
      // matrix-core form (k_cond.hip / k_cond_bwd.hip): gather-form data gradient through the layer's own table with the tap
      // flipped (submanifold) or through the inverse table (strided) -- no atomics, no zero fill.  At level 0 the gradients of
      // duplicate vertices' rows are folded into their representatives' first (the table is symmetric over those only).
      const bool lvl0_subm = !L.strided && sp_nbr[i] == m.nbr_subm[0];
      if (lvl0_subm) RET_IF(cbwd_sparse_fold_dups(d_cur, sp_nbr[i], sp_nout[i], L.cout, s));
      {
        WsScope sc(c, WS_TEMP);
        float* dw_part = F((size_t)cbwd_sparse_wgrad_chunks(sp_nout[i]) * 27 * L.cin * L.cout);
        WS_CHECK(dw_part);
        RET_IF(cbwd_sparse_wgrad_mfma(sp_in[i], sp_nbr[i], d_cur, sp_nout[i], L.cin, L.cout, dwp, dw_part, s));
      }
      const int* table = sp_nbr[i];
      WsScope sc(c, WS_TEMP);
      if (L.strided) {
        int* inv = (int*)F((size_t)sp_nin[i] * 27);
        WS_CHECK(inv);
        RET_IF(cbwd_sparse_inverse_table(sp_nbr[i], sp_nout[i], sp_nin[i], inv, s));
        table = inv;
      }
      RET_IF(launch_sparse_conv(d_cur, table, sp_nin[i], L.cout, L.cin, nullptr, L.wd, nullptr, nullptr, d_in, s, lvl0_subm ? 1 : 0));
    } else {
      HIP_CHECK_RET(hipMemsetAsync(d_in, 0, (size_t)sp_nin[i] * L.cin * sizeof(float), s));
      WsScope sc(c, WS_TEMP);
      float* dw_part = F((size_t)cbwd_sparse_wgrad_chunks(sp_nout[i]) * 27 * L.cin * L.cout);
      WS_CHECK(dw_part);
      RET_IF(cbwd_sparse_conv(sp_in[i], sp_nbr[i], d_cur, sp_nout[i], L.cin, L.cout, L.w, d_in, dwp, dw_part, s));
    }
    RET_IF(cbwd_sparse_w_unpack_add(dwp, L.cin, L.cout, L.layout, Gw, s));
    d_cur = d_in;
  }
  mark("bwd: sparse net");
  float* d_fused = d_cur;  // [Nv][16]
  if (dbg_dfused) HIP_CHECK_RET(hipMemcpyAsync(dbg_dfused, d_fused, (size_t)Nv * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
  // view fusion, vertex gather
  float *d_vf = F((size_t)N * Nv * 16), *d_feats = d_feats_all + (size_t)bi * rows * 16;
  float* fuse_part = F((size_t)cbwd_fuse_scratch_floats(Nv));
  WS_CHECK(d_vf && fuse_part);
  RET_IF(cbwd_fuse(d_fused, vf, c->fuse_w, N, Nv, N, d_vf, engine_grad(c, SV + "smpl_feature_extractor.conv0.weight"),
                   engine_grad(c, SV + "smpl_feature_extractor.conv0.bias"), fuse_part, s));
  RET_IF(cbwd_vertex_scatter(d_vf, c->cams, vidx, N, m.verts, Nv, V, c->v.spatial_volume_length, S, persp, d_feats, s));
  if (dbg_dfeats) RET_IF(launch_nhwc_to_nchw(d_feats, 16, N, 16, HW, dbg_dfeats, s));
  mark("bwd: fuse + vertex scatter");
  return 0;
  };
  for (int bi = 0; bi < B; ++bi) {
    RET_IF(engine_select_sample(c, slots[bi]));
    RET_IF(stage3(bi));
  }

  // ---------------- stage 4, all samples at once: 2-D encoder backward (B * N views as its batch), FiLM projections, step MLP ----------------
  {
  Fwd f{c, s, BN, BN, 0, nullptr, nullptr, nullptr, nullptr, {nullptr, nullptr, nullptr, nullptr}};
  TrainTape tape;
  Bwd b{c, s, BN, &f, &tape};
  half_t* dy16;
  float *d_a = F(rows_all * 16), *d_r = F(rows_all * 16), *d_x = F(rows_all * 16), *d_nxt = F(rows_all * 16);
  float* d_pre_e = F((size_t)BN * 48);
  WS_CHECK(d_a && d_r && d_x && d_nxt && d_pre_e);
  RET_IF(grad16(c, d_feats_all, 16, (long)rows_all, 16, &dy16, s));
  RET_IF(dgrad_conv3(b, c->enc_final, dy16, d_a, 16, S, S, false));
  RET_IF(wgrad_conv3(b, c->enc_final, d_feats_all, 16, af, 0, 48, S, S, 16, 1, 0));
  RET_IF(gn_backward(b, c->enc_final_norm, 8, 1e-5f, ACT_SILU, cur_e[3], 16, d_a, 16, HW, d_nxt, 16, false));
  for (int i = 2; i >= 0; --i) {
    const EncBlockW& e = c->enc_blocks[i];
    RET_IF(grad16(c, d_nxt, 16, (long)rows_all, 16, &dy16, s));
    RET_IF(dgrad_conv3(b, e.c2, dy16, d_a, 16, S, S, false));
    RET_IF(wgrad_conv3(b, e.c2, d_nxt, 16, a2_e[i], 0, 48, S, S, 16, 1, 0));
    RET_IF(gn_backward(b, e.n2, 8, 1e-5f, ACT_SILU, r1_e[i], 16, d_a, 16, HW, d_r, 16, false));
    RET_IF(grad16(c, d_r, 16, (long)rows_all, 16, &dy16, s));
    RET_IF(dgrad_conv3(b, e.c1, dy16, d_a, 16, S, S, false));
    RET_IF(wgrad_conv3(b, e.c1, d_r, 16, a1_e[i], 0, 48, S, S, 16, 1, 0));
    RET_IF(gn_backward(b, e.n1, 8, 1e-5f, ACT_SILU, cur_e[i], 16, d_a, 16, HW, d_x, 16, false, pre_e + 16 * i, 48, d_pre_e + 16 * i, 48));
    RET_IF(bwd_add_views(d_x, 16, d_nxt, 16, nullptr, 0, (long)rows_all, 16, 1, s));  // + the residual branch
    std::swap(d_x, d_nxt);
  }
  RET_IF(wgrad_conv3(b, c->enc_init, d_nxt, 16, x8, 1, 8, S, S, 4, 1, 0));  // 4 latent channels (the pack pads them to 8)
  {  // FiLM of the three encoder blocks: pre[b, v] = time_embed_i(t_emb[b]) + view_embed_i(v_embed[b, v])
    float* dsum = F((size_t)B * 48);
    WS_CHECK(dsum);
    RET_IF(bwd_colsum_samples(d_pre_e, 1, 48, B, N, 48, dsum, 48, s));  // per sample: its N views
    for (int i = 0; i < 3; ++i) {
      const EncBlockW& e = c->enc_blocks[i];
      RET_IF(lin_wgrad(c, e.t.key, dsum + 16 * i, 48, t_emb_all, td, B, 16, td, s));
      RET_IF(lin_wgrad(c, e.v.key, d_pre_e + 16 * i, 48, v_embed_all, vd, BN, 16, vd, s));
    }
    RET_IF(cbwd_small_linear_bwd(dsum, 48, B, 48, c->enc_t.w, td, d_temb_all, td, 1, s));
  }
  mark("bwd: 2-D encoder + FiLM");
  if (dbg_dtembed) HIP_CHECK_RET(hipMemcpyAsync(dbg_dtembed, d_temb_all, td * sizeof(float), hipMemcpyDeviceToDevice, s));
  // step MLP
  RET_IF(lin_wgrad(c, c->step_te2.key, d_temb_all, td, e1, td, B, td, td, s));
  float* d_e1 = F((size_t)B * td);
  WS_CHECK(d_e1);
  RET_IF(cbwd_small_linear_bwd(d_temb_all, td, B, td, c->step_te2.w, td, d_e1, td, 0, s));
  RET_IF(bwd_silu_inplace(d_e1, u1, (size_t)B * td, s));
  RET_IF(lin_wgrad(c, c->step_te0.key, d_e1, td, e0, td, B, td, td, s));
  }
  return 0;
}

// one sample: the active slot (parity hook with the debug outputs; dsrc as above with B = 1)
int engine_train_conditioner_backward(mvd_ctx* c, const float* x_noisy_nchw, int64_t timestep, const float* v_embed, int n_views,
                                      int target_idx, float* const dsrc[4], float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats,
                                      float* dbg_dtembed, hipStream_t s) {
  const int slot = c->cur_slot;
  return engine_train_conditioner_backward_batch(c, 1, &slot, x_noisy_nchw, &timestep, v_embed, n_views, &target_idx, dsrc, dbg_dvolume,
                                                 dbg_dfused, dbg_dfeats, dbg_dtembed, s);
}
