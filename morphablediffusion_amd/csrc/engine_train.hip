// Training-step backward, first slice: gradients of every parameter of the LAST DepthTransformer
// (output_conditions.<last>, attention.py:49-84) of the loss of training_step (morphable_diffusion.py:520-549).
// The UNet forward runs in the inference engine with the tape on: the input of that DepthTransformer and the UNet's final
// hidden state are kept (fp32).  From those, the block's forward is re-computed in fp32 from the master weights (kept as
// uploaded) with every intermediate saved, and the loss gradient is propagated
//     dL/dpred -> out conv (dgrad) -> SiLU / GroupNorm32 -> [x + proj_out(.)] -> conv3x3, ReLU, GN8, conv3x3, ReLU, GN8
//              -> to_out -> depth attention -> to_q / to_k / to_v -> proj_in (conv1x1, GN8, SiLU) / proj_context (conv1x1x1,
//                 GN8, ReLU)
// with the weight / bias / gain gradients collected on the way.  Everything upstream (the other DepthTransformers, the UNet's
// own blocks, the conditioner) needs the backward of the whole UNet and is not built yet.
#include <string.h>

#include "engine.h"

int train_sgemm(const float* A, int lda, int ta, const float* B, int ldb, int tb, float* C, int M, int N, int K, float* scratch,
                size_t scratch_floats, hipStream_t s);
int train_im2col3(const float* X, int B, int H, int W, int C, float* col, hipStream_t s);
int train_col2im3(const float* dcol, int B, int H, int W, int C, float* dX, hipStream_t s);
int train_perm_w3(const float* src, int N, int C, int to_mat, float* dst, hipStream_t s);
int train_gn_fwd(const float* x, int B, int rows, int C, int G, const float* gamma, const float* beta, float eps, int act, float* y,
                 float* stats, hipStream_t s);
int train_gn_bwd(const float* x, const float* dy, int B, int rows, int C, int G, const float* gamma, const float* beta,
                 const float* stats, int act, float* dx, float* dgamma, float* dbeta, float* tmp1, float* tmp2, hipStream_t s);
int train_colsum(const float* v, long R, int C, float* out, hipStream_t s);
int train_depth_fwd(const float* q, const float* k, const float* v, int R, int HW, int D, int hn, int hd, float scale, float* attn,
                    float* z, hipStream_t s);
int train_depth_bwd(const float* q, const float* k, const float* v, const float* attn, const float* dz, int R, int HW, int D, int hn,
                    int hd, float scale, float* dq, float* dk, float* dv, hipStream_t s);
int train_add_inplace(float* a, const float* b, size_t n, hipStream_t s);
int train_copy_rows(const float* src, int ld, long rows, int C, float* dst, hipStream_t s);
int train_add_bias_rows(float* x, long rows, int C, const float* bias, hipStream_t s);

namespace {
const float* TW(mvd_ctx* c, const std::string& k) {
  auto it = c->train_w.find(k);
  return it == c->train_w.end() ? nullptr : it->second.d;
}
float* grad_buf(mvd_ctx* c, const std::string& k) {
  auto it = c->train_w.find(k);
  if (it == c->train_w.end()) return nullptr;
  RawTensor& g = c->train_g[k];
  if (!g.d) {
    if (hipMalloc((void**)&g.d, it->second.numel * sizeof(float)) != hipSuccess) return nullptr;
    g.numel = it->second.numel;
    g.shape = it->second.shape;
  }
  return g.d;
}
}  // namespace

// keeps fp32 copies of the parameters this slice differentiates (called from engine_finalize while the raw tensors exist)
int engine_train_keep(mvd_ctx* c) {
  if (!c->has_unet || c->conds.empty()) return 0;
  const std::string U = "model.diffusion_model.";
  c->train_prefix = c->conds.back().key + ".";
  for (auto& kv : c->raw) {
    const std::string& k = kv.first;
    if (k.rfind(c->train_prefix, 0) != 0 && k.rfind(U + "out.", 0) != 0) continue;
    RawTensor t = kv.second;
    t.d = nullptr;
    HIP_CHECK_RET(hipMalloc((void**)&t.d, std::max<size_t>(t.numel, 1) * sizeof(float)));
    HIP_CHECK_RET(hipMemcpy(t.d, kv.second.d, t.numel * sizeof(float), hipMemcpyDeviceToDevice));
    c->train_w[k] = t;
  }
  return 0;
}

int engine_tape_enable(mvd_ctx* c, int max_batch) {
  for (float** p : {&c->tape_x, &c->tape_h}) {
    if (*p) hipFree(*p);
    *p = nullptr;
  }
  c->tape_B = 0;
  c->tape_valid = 0;
  if (max_batch <= 0) return 0;
  const size_t n = (size_t)max_batch * c->u.image_size * c->u.image_size * c->u.model_channels * c->u.channel_mult[0];
  HIP_CHECK_RET(hipMalloc((void**)&c->tape_x, n * sizeof(float)));
  HIP_CHECK_RET(hipMalloc((void**)&c->tape_h, n * sizeof(float)));
  c->tape_B = max_batch;
  return 0;
}

// called by engine_unet around the last DepthTransformer when the tape is on
int engine_tape_record(mvd_ctx* c, const float* x, int ldx, const float* h, int ldh, int Bv, hipStream_t s) {
  if (!c->tape_B) return 0;
  if (Bv > c->tape_B) return mvd_fail("training tape: batch larger than mvd_train_tape(max_batch)");
  const int S = c->u.image_size, dim = c->u.model_channels * c->u.channel_mult[0];
  RET_IF(train_copy_rows(x, ldx, (long)Bv * S * S, dim, c->tape_x, s));
  RET_IF(train_copy_rows(h, ldh, (long)Bv * S * S, dim, c->tape_h, s));
  c->tape_valid = Bv;
  return 0;
}

// dpred [B,oc,S,S] (NCHW) = dL/d(UNet output); ctx0 [B,Cc,D,S,S] (NCDHW) = the finest source_dict volume the forward saw
int engine_train_backward_last_condition(mvd_ctx* c, const float* dpred_nchw, const float* ctx0_ncdhw, int B, int D, hipStream_t s) {
  if (!c->finalized || !c->has_unet) return mvd_fail("UNet weights not uploaded / finalized");
  if (c->tape_valid != B) return mvd_fail("training backward: run the UNet forward with the tape on (same batch) first");
  const std::string U = "model.diffusion_model.", P = c->train_prefix;
  const CondW& cd = c->conds.back();
  const int S = c->u.image_size, HW = S * S, R = B * HW, dim = cd.dim, I = cd.I, Cc = cd.Cc, hn = 4, hd = Cc / 2, oc = c->u.out_channels;
  if (D != (48 * S) / 32 && D <= 0) return mvd_fail("training backward: bad depth");
  const long RC = (long)B * D * HW;  // context rows
  const float scale = 1.0f / sqrtf((float)hd);
  auto W = [&](const char* n) { return TW(c, P + n); };
  const float *w_pi = W("proj_in.0.weight"), *b_pi = W("proj_in.0.bias"), *g_pi = W("proj_in.1.weight"), *e_pi = W("proj_in.1.bias");
  const float *w_pc = W("proj_context.0.weight"), *g_pc = W("proj_context.1.weight"), *e_pc = W("proj_context.1.bias");
  const float *w_q = W("depth_attn.to_q.weight"), *w_k = W("depth_attn.to_k.weight"), *w_v = W("depth_attn.to_v.weight"),
              *w_o = W("depth_attn.to_out.weight");
  const float *g_o0 = W("proj_out.0.weight"), *e_o0 = W("proj_out.0.bias"), *w_c1 = W("proj_out.2.weight");
  const float *g_o3 = W("proj_out.3.weight"), *e_o3 = W("proj_out.3.bias"), *w_c2 = W("proj_out.5.weight");
  const float *g_out = TW(c, U + "out.0.weight"), *e_out = TW(c, U + "out.0.bias"), *w_out = TW(c, U + "out.2.weight");
  for (const float* q_ : {w_pi, b_pi, g_pi, e_pi, w_pc, g_pc, e_pc, w_q, w_k, w_v, w_o, g_o0, e_o0, w_c1, g_o3, e_o3, w_c2, g_out, e_out, w_out})
    if (!q_) return mvd_fail("training backward: master weights of the last DepthTransformer / output head were not kept");
  const char* gnames[] = {"proj_in.0.weight", "proj_in.0.bias", "proj_in.1.weight", "proj_in.1.bias", "proj_context.0.weight",
                          "proj_context.1.weight", "proj_context.1.bias", "depth_attn.to_q.weight", "depth_attn.to_k.weight",
                          "depth_attn.to_v.weight", "depth_attn.to_out.weight", "proj_out.0.weight", "proj_out.0.bias",
                          "proj_out.2.weight", "proj_out.3.weight", "proj_out.3.bias", "proj_out.5.weight"};
  for (const char* n : gnames)
    if (!grad_buf(c, P + n)) return mvd_fail("training backward: gradient buffer allocation failed");
  for (const char* n : {"out.0.weight", "out.0.bias", "out.2.weight", "out.2.bias"})  // the output head (finetune_unet=True)
    if (!grad_buf(c, U + n)) return mvd_fail("training backward: gradient buffer allocation failed");
  auto G = [&](const char* n) { return c->train_g[P + n].d; };
  WsScope ws_scope(c);
  auto F = [&](size_t n) { return ws_alloc<float>(c, n); };
  // ---------------- forward recompute, fp32, every intermediate kept ----------------
  const float *X = c->tape_x, *Hh = c->tape_h;
  float* C0 = F((size_t)RC * Cc);
  float *p = F((size_t)R * I), *pn = F((size_t)R * I), *st_pi = F(B * 8 * 2);
  float *pc = F((size_t)RC * Cc), *cn = F((size_t)RC * Cc), *st_pc = F(B * 8 * 2);
  float *q = F((size_t)R * I), *k = F((size_t)RC * I), *v = F((size_t)RC * I);
  float *attn = F((size_t)R * hn * D), *z = F((size_t)R * I), *o = F((size_t)R * I);
  float *a1 = F((size_t)R * I), *st_o0 = F(B * 8 * 2), *col1 = F((size_t)R * 9 * I), *o2 = F((size_t)R * I);
  float *a2 = F((size_t)R * I), *st_o3 = F(B * 8 * 2), *col2 = F((size_t)R * 9 * I);
  float *aout = F((size_t)R * dim), *st_out = F(B * 32 * 2);
  float *m_c1 = F((size_t)I * 9 * I), *m_c2 = F((size_t)dim * 9 * I), *m_out = F((size_t)oc * 9 * dim);
  const size_t scr_n = (size_t)64 << 20;  // floats: split-reduction scratch of the GEMMs
  float* scr = F(scr_n);
  // backward buffers
  float *dpred = F((size_t)R * oc), *dcolo = F((size_t)R * 9 * dim), *da = F((size_t)R * dim), *dh = F((size_t)R * dim);
  float *t1 = F((size_t)RC * Cc > (size_t)R * dim ? (size_t)RC * Cc : (size_t)R * dim), *t2 = F((size_t)RC * Cc > (size_t)R * dim ? (size_t)RC * Cc : (size_t)R * dim);
  float *dcol = F((size_t)R * 9 * I), *dI1 = F((size_t)R * I), *dI2 = F((size_t)R * I), *mg = F((size_t)dim * 9 * I);
  float *dk = F((size_t)RC * I), *dv = F((size_t)RC * I), *dcn = F((size_t)RC * Cc), *dpc = F((size_t)RC * Cc);
  WS_CHECK(C0 && p && pn && st_pi && pc && cn && st_pc && q && k && v && attn && z && o && a1 && st_o0 && col1 && o2 && a2 && st_o3 &&
           col2 && aout && st_out && m_c1 && m_c2 && m_out && scr && dpred && dcolo && da && dh && t1 && t2 && dcol && dI1 && dI2 &&
           mg && dk && dv && dcn && dpc);
  auto gemm = [&](const float* A, int lda, int ta, const float* Bm, int ldb, int tb, float* Cm, int M, int N, int K) {
    return train_sgemm(A, lda, ta, Bm, ldb, tb, Cm, M, N, K, scr, scr_n, s);
  };
  RET_IF(launch_nchw_to_nhwc(ctx0_ncdhw, B, Cc, D * HW, C0, Cc, Cc, s));
  RET_IF(train_perm_w3(w_c1, I, I, 1, m_c1, s));
  RET_IF(train_perm_w3(w_c2, dim, I, 1, m_c2, s));
  RET_IF(train_perm_w3(w_out, oc, dim, 1, m_out, s));
  // proj_in: conv1x1 + bias, GN8, SiLU        (attention.py:52-56)
  RET_IF(gemm(X, dim, 0, w_pi, dim, 1, p, R, I, dim));
  RET_IF(train_add_bias_rows(p, R, I, b_pi, s));
  RET_IF(train_gn_fwd(p, B, HW, I, 8, g_pi, e_pi, 1e-5f, ACT_SILU, pn, st_pi, s));
  // proj_context: conv1x1x1 (no bias), GN8, ReLU   (:57-61)
  RET_IF(gemm(C0, Cc, 0, w_pc, Cc, 1, pc, (int)RC, Cc, Cc));
  RET_IF(train_gn_fwd(pc, B, D * HW, Cc, 8, g_pc, e_pc, 1e-5f, ACT_RELU, cn, st_pc, s));
  // depth attention   (:26-47)
  RET_IF(gemm(pn, I, 0, w_q, I, 1, q, R, I, I));
  RET_IF(gemm(cn, Cc, 0, w_k, Cc, 1, k, (int)RC, I, Cc));
  RET_IF(gemm(cn, Cc, 0, w_v, Cc, 1, v, (int)RC, I, Cc));
  RET_IF(train_depth_fwd(q, k, v, R, HW, D, hn, hd, scale, attn, z, s));
  RET_IF(gemm(z, I, 0, w_o, I, 1, o, R, I, I));
  // proj_out: GN8, ReLU, conv3x3, GN8, ReLU, conv3x3   (:63-70)
  RET_IF(train_gn_fwd(o, B, HW, I, 8, g_o0, e_o0, 1e-5f, ACT_RELU, a1, st_o0, s));
  RET_IF(train_im2col3(a1, B, S, S, I, col1, s));
  RET_IF(gemm(col1, 9 * I, 0, m_c1, 9 * I, 1, o2, R, I, 9 * I));
  RET_IF(train_gn_fwd(o2, B, HW, I, 8, g_o3, e_o3, 1e-5f, ACT_RELU, a2, st_o3, s));
  RET_IF(train_im2col3(a2, B, S, S, I, col2, s));
  // (the block's output x + conv(a2) is the taped final hidden state Hh)
  // output head: GN32, SiLU (openaimodel.py:717-719); its conv's input is only needed for the (frozen) conv's own wgrad
  RET_IF(train_gn_fwd(Hh, B, HW, dim, 32, g_out, e_out, 1e-5f, ACT_SILU, aout, st_out, s));
  // ---------------- backward ----------------
  RET_IF(launch_nchw_to_nhwc(dpred_nchw, B, oc, HW, dpred, oc, oc, s));
  RET_IF(gemm(dpred, oc, 0, m_out, 9 * dim, 0, dcolo, R, 9 * dim, oc));            // dgrad of the output conv
  RET_IF(train_col2im3(dcolo, B, S, S, dim, da, s));
  RET_IF(train_gn_bwd(Hh, da, B, HW, dim, 32, g_out, e_out, st_out, ACT_SILU, dh, c->train_g[U + "out.0.weight"].d,
                      c->train_g[U + "out.0.bias"].d, t1, t2, s));
  {  // the output conv's own weight / bias gradient: dW = dpred^T im2col(a), db = column sums of dpred
    float* colo = dcolo;  // [R][9*dim]: the dgrad columns are consumed, the buffer is free again
    float* mgo = F((size_t)oc * 9 * dim);
    WS_CHECK(mgo);
    RET_IF(train_im2col3(aout, B, S, S, dim, colo, s));
    RET_IF(gemm(dpred, oc, 1, colo, 9 * dim, 0, mgo, oc, 9 * dim, R));
    RET_IF(train_perm_w3(mgo, oc, dim, 0, c->train_g[U + "out.2.weight"].d, s));
    RET_IF(train_colsum(dpred, R, oc, c->train_g[U + "out.2.bias"].d, s));
  }
  // dh = dL/d(x + proj_out(.)): second conv3x3 of proj_out
  RET_IF(gemm(dh, dim, 1, col2, 9 * I, 0, mg, dim, 9 * I, R));                      // wgrad [dim][9][I]
  RET_IF(train_perm_w3(mg, dim, I, 0, G("proj_out.5.weight"), s));
  RET_IF(gemm(dh, dim, 0, m_c2, 9 * I, 0, dcol, R, 9 * I, dim));
  RET_IF(train_col2im3(dcol, B, S, S, I, dI1, s));                                  // d a2
  RET_IF(train_gn_bwd(o2, dI1, B, HW, I, 8, g_o3, e_o3, st_o3, ACT_RELU, dI2, G("proj_out.3.weight"), G("proj_out.3.bias"), t1, t2, s));
  RET_IF(gemm(dI2, I, 1, col1, 9 * I, 0, mg, I, 9 * I, R));                         // first conv3x3
  RET_IF(train_perm_w3(mg, I, I, 0, G("proj_out.2.weight"), s));
  RET_IF(gemm(dI2, I, 0, m_c1, 9 * I, 0, dcol, R, 9 * I, I));
  RET_IF(train_col2im3(dcol, B, S, S, I, dI1, s));                                  // d a1
  RET_IF(train_gn_bwd(o, dI1, B, HW, I, 8, g_o0, e_o0, st_o0, ACT_RELU, dI2, G("proj_out.0.weight"), G("proj_out.0.bias"), t1, t2, s));
  // to_out (1x1, no bias): dI2 = d o
  RET_IF(gemm(dI2, I, 1, z, I, 0, G("depth_attn.to_out.weight"), I, I, R));
  RET_IF(gemm(dI2, I, 0, w_o, I, 0, dI1, R, I, I));                                 // d z
  float* dq = dI2;
  RET_IF(train_depth_bwd(q, k, v, attn, dI1, R, HW, D, hn, hd, scale, dq, dk, dv, s));
  RET_IF(gemm(dq, I, 1, pn, I, 0, G("depth_attn.to_q.weight"), I, I, R));
  RET_IF(gemm(dk, I, 1, cn, Cc, 0, G("depth_attn.to_k.weight"), I, Cc, (int)RC));
  RET_IF(gemm(dv, I, 1, cn, Cc, 0, G("depth_attn.to_v.weight"), I, Cc, (int)RC));
  // d cn = dk W_k + dv W_v ; proj_context backward
  RET_IF(gemm(dk, I, 0, w_k, Cc, 0, dcn, (int)RC, Cc, I));
  RET_IF(gemm(dv, I, 0, w_v, Cc, 0, dpc, (int)RC, Cc, I));
  RET_IF(train_add_inplace(dcn, dpc, (size_t)RC * Cc, s));
  RET_IF(train_gn_bwd(pc, dcn, B, D * HW, Cc, 8, g_pc, e_pc, st_pc, ACT_RELU, dpc, G("proj_context.1.weight"), G("proj_context.1.bias"), t1, t2, s));
  RET_IF(gemm(dpc, Cc, 1, C0, Cc, 0, G("proj_context.0.weight"), Cc, Cc, (int)RC));
  // d pn = dq W_q ; proj_in backward
  RET_IF(gemm(dq, I, 0, w_q, I, 0, dI1, R, I, I));
  float* dp = dcol;  // [R][I] fits
  RET_IF(train_gn_bwd(p, dI1, B, HW, I, 8, g_pi, e_pi, st_pi, ACT_SILU, dp, G("proj_in.1.weight"), G("proj_in.1.bias"), t1, t2, s));
  RET_IF(gemm(dp, I, 1, X, dim, 0, G("proj_in.0.weight"), I, dim, R));
  RET_IF(train_colsum(dp, R, I, G("proj_in.0.bias"), s));
  return 0;
}
