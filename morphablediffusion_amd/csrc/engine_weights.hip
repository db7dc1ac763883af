// Weight store: turns the reference's state_dict (uploaded key by key, fp32, reference layouts) into the
// packed / folded device layouts the kernels consume, and builds the UNet block plan from the config
// exactly as the reference ctor does (openaimodel.py:535-720, attention.py:87-115).
#include <algorithm>
#include <math.h>
#include <string.h>

#include "engine.h"

// Owned device allocation.  While re-packing (engine_repack) the build functions run again in the same order and get the
// allocations of the first run back, so every pointer the executors hold stays valid and no memory is allocated.
int engine_dmalloc(mvd_ctx* c, void** p, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (c->repacking) {
    if (c->repack_cursor >= c->sec_end || c->owned_bytes[c->repack_cursor] != bytes)
      return mvd_fail("engine_repack: allocation sequence differs from the first build");
    *p = c->owned[c->repack_cursor++];
    return 0;
  }
  HIP_CHECK_RET(hipMalloc(p, bytes));
  c->owned.push_back(*p);
  c->owned_bytes.push_back(bytes);
  return 0;
}

// Build streams of a multi-stream re-pack (mvd_ctx::bs): the launches of one packed tensor stay on one stream, consecutive
// tensors go round-robin; a join makes every stream wait for all of them (before packs that read earlier packs: the adjoint
// weights of engine_build_dgrad*).  Outside engine_repack both are no-ops and bs stays the null stream.
void engine_build_rotate(mvd_ctx* c) {
  if (c->bs_multi) c->bs = c->bstreams[(c->bs_rr++) & 3];
}
int engine_build_join(mvd_ctx* c) {
  if (!c->bs_multi) return 0;
  for (int i = 0; i < 4; ++i) HIP_CHECK_RET(hipEventRecord(c->bevents[i], c->bstreams[i]));
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (i != j) HIP_CHECK_RET(hipStreamWaitEvent(c->bstreams[i], c->bevents[j], 0));
  return 0;
}

namespace {

int dmalloc(mvd_ctx* c, void** p, size_t bytes) { return engine_dmalloc(c, p, bytes); }

int get_raw(mvd_ctx* c, const std::string& k, RawTensor** out) {
  auto it = c->raw.find(k);
  if (it == c->raw.end()) {
    static thread_local std::string msg;
    msg = "missing weight: " + k;
    return mvd_fail(msg.c_str());
  }
  *out = &it->second;
  return 0;
}

int copy_f32(mvd_ctx* c, const std::string& k, float** out, int* n = nullptr) {
  RawTensor* r;
  RET_IF(get_raw(c, k, &r));
  if (c->train_mode && c->param_index.count(k)) {  // biases / gains are read straight from the master arena: always current,
    *out = r->d;                                    // no copy to refresh at re-pack time
    if (n) *n = (int)r->numel;
    return 0;
  }
  RET_IF(dmalloc(c, (void**)out, r->numel * sizeof(float)));
  HIP_CHECK_RET(hipMemcpyAsync(*out, r->d, r->numel * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
  if (n) *n = (int)r->numel;
  return 0;
}

int load_norm(mvd_ctx* c, const std::string& p, NormW* n) {
  n->key = p;
  RET_IF(copy_f32(c, p + ".weight", &n->g, &n->C));
  RET_IF(copy_f32(c, p + ".bias", &n->b));
  return 0;
}

// conv / linear weight -> fp16 [taps][N][Cin(+pad)]
int pack_conv(mvd_ctx* c, const std::string& wkey, const std::string& bkey, bool transposed, bool geglu, ConvW* o,
              int cin_pad = 0, bool xp = false) {
  engine_build_rotate(c);
  RawTensor* r;
  RET_IF(get_raw(c, wkey, &r));
  if (r->shape.size() < 2) return mvd_fail("pack_conv: weight rank < 2");
  int d0 = (int)r->shape[0], d1 = (int)r->shape[1];
  int taps = 1;
  for (size_t i = 2; i < r->shape.size(); ++i) taps *= (int)r->shape[i];
  const int N = transposed ? d1 : d0, cin_src = transposed ? d0 : d1;
  const int Cl = cin_pad > cin_src ? cin_pad : cin_src;
  const int Cin = xp ? 3 * Cl : Cl;
  o->N = N;
  o->Cin = Cin;
  o->xp = xp ? 1 : 0;
  o->cin_l = Cl;
  o->taps = taps;
  o->key = wkey;
  o->bkey = bkey;
  RET_IF(dmalloc(c, (void**)&o->w, (size_t)taps * N * Cin * sizeof(half_t)));
  RET_IF(launch_pack_weight(r->d, N, Cin, taps, transposed ? 1 : 0, geglu ? 1 : 0, o->w, c->bs, cin_src, xp ? 1 : 0));
  if (!bkey.empty()) {
    if (geglu) {
      RawTensor* b;
      RET_IF(get_raw(c, bkey, &b));
      RET_IF(dmalloc(c, (void**)&o->bias, (size_t)N * sizeof(float)));
      RET_IF(launch_permute_geglu_bias(b->d, N, o->bias, c->bs));
    } else {
      RET_IF(copy_f32(c, bkey, &o->bias));
    }
  }
  return 0;
}

int pack_lin(mvd_ctx* c, const std::string& wkey, const std::string& bkey, LinW* o) {
  engine_build_rotate(c);
  RawTensor* r;
  RET_IF(get_raw(c, wkey, &r));
  o->N = (int)r->shape[0];
  o->K = (int)(r->numel / r->shape[0]);
  o->key = wkey.size() > 7 ? wkey.substr(0, wkey.size() - 7) : wkey;  // strip ".weight"
  RET_IF(dmalloc(c, (void**)&o->w, r->numel * sizeof(half_t)));
  RET_IF(launch_f32_to_f16(r->d, o->w, r->numel, c->bs));
  if (!bkey.empty()) RET_IF(copy_f32(c, bkey, &o->bias));
  return 0;
}

int build_res(mvd_ctx* c, const std::string& p, int cin, int cout, ResW* r) {
  r->key = p;
  r->cin = cin;
  r->cout = cout;
  RET_IF(load_norm(c, p + ".in_layers.0", &r->n1));
  RET_IF(pack_conv(c, p + ".in_layers.2.weight", p + ".in_layers.2.bias", false, false, &r->c1));
  RET_IF(load_norm(c, p + ".out_layers.0", &r->n2));
  RET_IF(pack_conv(c, p + ".out_layers.3.weight", p + ".out_layers.3.bias", false, false, &r->c2));
  r->has_skip = c->raw.count(p + ".skip_connection.weight") > 0;
  if (r->has_skip) RET_IF(pack_conv(c, p + ".skip_connection.weight", p + ".skip_connection.bias", false, false, &r->skip));
  else if (cin != cout) return mvd_fail("ResBlock changes width but has no skip_connection weight");
  return 0;
}

int build_st(mvd_ctx* c, const std::string& p, int C, STW* s) {
  s->key = p;
  s->C = C;
  s->heads = c->u.num_heads;
  const std::string t = p + ".transformer_blocks.0";
  RET_IF(load_norm(c, p + ".norm", &s->norm));
  RET_IF(pack_conv(c, p + ".proj_in.weight", p + ".proj_in.bias", false, false, &s->proj_in));
  RET_IF(pack_conv(c, p + ".proj_out.weight", p + ".proj_out.bias", false, false, &s->proj_out));
  RET_IF(load_norm(c, t + ".norm1", &s->ln1));
  RET_IF(load_norm(c, t + ".norm3", &s->ln3));
  // to_q | to_k | to_v stacked: one projection GEMM, the attention kernel reads V row-major out of its third block
  RawTensor *q, *k, *v;
  RET_IF(get_raw(c, t + ".attn1.to_q.weight", &q));
  RET_IF(get_raw(c, t + ".attn1.to_k.weight", &k));
  RET_IF(get_raw(c, t + ".attn1.to_v.weight", &v));
  s->qkv.N = 3 * C;
  s->qkv.Cin = C;
  s->qkv.cin_l = C;
  s->qkv.taps = 1;
  RET_IF(dmalloc(c, (void**)&s->qkv.w, (size_t)3 * C * C * sizeof(half_t)));
  RET_IF(launch_f32_to_f16(q->d, s->qkv.w, (size_t)C * C, c->bs));
  RET_IF(launch_f32_to_f16(k->d, s->qkv.w + (size_t)C * C, (size_t)C * C, c->bs));
  RET_IF(launch_f32_to_f16(v->d, s->qkv.w + (size_t)2 * C * C, (size_t)C * C, c->bs));
  RET_IF(pack_conv(c, t + ".attn1.to_out.0.weight", t + ".attn1.to_out.0.bias", false, false, &s->attn_out));
  RET_IF(pack_conv(c, t + ".ff.net.0.proj.weight", t + ".ff.net.0.proj.bias", false, true, &s->ff1));
  RET_IF(pack_conv(c, t + ".ff.net.2.weight", t + ".ff.net.2.bias", false, false, &s->ff2));
  return 0;
}

int build_cond(mvd_ctx* c, const std::string& p, int dim, int Cc, CondW* d) {
  const int heads = 4, hd = Cc / 2, I = 2 * Cc;
  d->key = p;
  d->dim = dim;
  d->Cc = Cc;
  d->I = I;
  RET_IF(pack_conv(c, p + ".proj_in.0.weight", p + ".proj_in.0.bias", false, false, &d->proj_in));
  RET_IF(load_norm(c, p + ".proj_in.1", &d->gn_in));
  RET_IF(pack_conv(c, p + ".proj_context.0.weight", "", false, false, &d->proj_ctx));
  RET_IF(load_norm(c, p + ".proj_context.1", &d->gn_ctx));
  RawTensor *wq, *wk, *wv, *wo;
  RET_IF(get_raw(c, p + ".depth_attn.to_q.weight", &wq));
  RET_IF(get_raw(c, p + ".depth_attn.to_k.weight", &wk));
  RET_IF(get_raw(c, p + ".depth_attn.to_v.weight", &wv));
  RET_IF(get_raw(c, p + ".depth_attn.to_out.weight", &wo));
  d->wqk.N = heads * Cc;
  d->wqk.Cin = I;
  d->wqk.taps = 1;
  RET_IF(dmalloc(c, (void**)&d->wqk.w, (size_t)heads * Cc * I * sizeof(half_t)));
  RET_IF(launch_fold_qk(wq->d, wk->d, heads, hd, Cc, I, 1.0f / sqrtf((float)hd), d->wqk.w, c->bs));
  d->wov.N = I;
  d->wov.Cin = heads * Cc;
  d->wov.taps = 1;
  RET_IF(dmalloc(c, (void**)&d->wov.w, (size_t)heads * Cc * I * sizeof(half_t)));
  RET_IF(launch_fold_ov(wo->d, wv->d, heads, hd, Cc, I, d->wov.w, c->bs));
  RET_IF(load_norm(c, p + ".proj_out.0", &d->gn_o1));
  RET_IF(pack_conv(c, p + ".proj_out.2.weight", "", false, false, &d->conv1));
  RET_IF(load_norm(c, p + ".proj_out.3", &d->gn_o2));
  RET_IF(pack_conv(c, p + ".proj_out.5.weight", "", false, false, &d->conv2));
  RET_IF(dmalloc(c, (void**)&d->relu_beta, (size_t)heads * Cc * sizeof(half_t)));
  RET_IF(launch_relu_beta_tile(d->gn_ctx.b, Cc, heads, d->relu_beta, c->bs));
  return 0;
}

int build_sparse_layer(mvd_ctx* c, const std::string& p, const std::string& blk, int slot, bool strided, SparseLayerW* L) {
  RawTensor *w, *g, *b, *rm, *rv;
  const std::string wk = p + blk + "." + std::to_string(slot) + ".weight";
  const std::string bn = p + blk + "." + std::to_string(slot + 1);
  RET_IF(get_raw(c, wk, &w));
  RET_IF(get_raw(c, bn + ".weight", &g));
  RET_IF(get_raw(c, bn + ".bias", &b));
  RET_IF(get_raw(c, bn + ".running_mean", &rm));
  RET_IF(get_raw(c, bn + ".running_var", &rv));
  // three layouts of the same 3x3x3 kernel (offset index k = (kd*3 + kh)*3 + kw in all of them):
  //   [cout][cin][3][3][3]   nn.Conv3d order (the dense emulation the goldens were generated with)
  //   [cout][3][3][3][cin]   spconv >= 2.2 (KRSC, what an un-pinned `spconv-cu113` installs, requirements.txt:18)
  //   [3][3][3][cin][cout]   spconv 1.x / 2.1
  // told apart by where the three 3s sit (channel counts are 16 / 32 / 64, never 3)
  if (w->shape.size() != 5) return mvd_fail("sparse conv weight: rank 5 expected");
  const int64_t* sh5 = w->shape.data();
  int layout = -1, cout = 0, cin = 0;
  if (sh5[2] == 3 && sh5[3] == 3 && sh5[4] == 3 && sh5[0] != 3) { layout = 0; cout = (int)sh5[0]; cin = (int)sh5[1]; }
  else if (sh5[1] == 3 && sh5[2] == 3 && sh5[3] == 3 && sh5[0] != 3) { layout = 1; cout = (int)sh5[0]; cin = (int)sh5[4]; }
  else if (sh5[0] == 3 && sh5[1] == 3 && sh5[2] == 3) { layout = 2; cin = (int)sh5[3]; cout = (int)sh5[4]; }
  else return mvd_fail("sparse conv weight: not a 3x3x3 kernel in a known layout");
  if ((int)g->numel != cout) return mvd_fail("sparse conv weight: channel count does not match its BatchNorm");
  L->cin = cin;
  L->cout = cout;
  L->strided = strided;
  L->wkey = wk;
  L->bnkey = bn;
  L->layout = layout;
  // packed on the device (no host round trip: the re-pack of a training step runs this for nine layers)
  RET_IF(dmalloc(c, (void**)&L->w, w->numel * 4));
  RET_IF(dmalloc(c, (void**)&L->scale, cout * 4));
  RET_IF(dmalloc(c, (void**)&L->shift, cout * 4));
  RET_IF(launch_sparse_w_pack(w->d, cin, cout, layout, L->w, c->bs));                       // -> [27][cin][cout]
  // MVD_SPARSE_VALU=1 (A/B switch, read when the weights are built): the one-site-per-workgroup kernels and the scatter-form
  // data gradient instead -- tests/test_gpu_train.py holds the two forms against each other
  const bool valu_only = getenv("MVD_SPARSE_VALU") != nullptr && getenv("MVD_SPARSE_VALU")[0] == '1';
  if (sparse_mfma_takes(cin, cout) && !valu_only) {  // B fragments of the matrix-core kernel; the data-gradient's in a training context
    RET_IF(dmalloc(c, (void**)&L->wp, w->numel * 4));
    RET_IF(launch_sparse_w_frag(L->w, cin, cout, 0, 0, L->wp, c->bs));
    if (c->train_mode) {
      RET_IF(dmalloc(c, (void**)&L->wd, w->numel * 4));
      RET_IF(launch_sparse_w_frag(L->w, cout, cin, 1, strided ? 0 : 1, L->wd, c->bs));
    }
  }
  RET_IF(launch_bn_fold(g->d, b->d, rm->d, rv->d, 1e-3f, cout, L->scale, L->shift, c->bs));  // eval BatchNorm1d(eps 1e-3), network.py:105
  RET_IF(copy_f32(c, bn + ".weight", &L->gamma));
  RET_IF(copy_f32(c, bn + ".bias", &L->beta));
  if (c->train_mode) {  // the buffers stay resident in a training context (engine_finalize): train-mode forwards update them
    L->rmean = rm->d;
    L->rvar = rv->d;
  }
  return 0;
}

int build_frustum_block(mvd_ctx* c, const std::string& p, const char* norm_name, bool transposed, int stride,
                        FrustumBlockW* f) {
  RET_IF(pack_lin(c, p + "t_conv.weight", p + "t_conv.bias", &f->t_conv));
  RET_IF(pack_lin(c, p + "v_conv.weight", p + "v_conv.bias", &f->v_conv));
  RET_IF(load_norm(c, p + norm_name, &f->gn));
  RET_IF(pack_conv(c, p + "conv.weight", p + "conv.bias", transposed, false, &f->conv));
  f->cin = f->conv.Cin;
  f->cout = f->conv.N;
  f->stride = stride;
  return 0;
}

}  // namespace

// ---------------- first-stage decoder: shapes are read off the uploaded tensors (no extra config struct) ----------
int build_vae_res(mvd_ctx* c, const std::string& p, VaeResW* r) {
  RawTensor* w1;
  RET_IF(get_raw(c, p + ".conv1.weight", &w1));
  r->cout = (int)w1->shape[0];
  r->cin = (int)w1->shape[1];
  RET_IF(load_norm(c, p + ".norm1", &r->n1));
  RET_IF(pack_conv(c, p + ".conv1.weight", p + ".conv1.bias", false, false, &r->c1, 0, c->vae_exact));
  RET_IF(load_norm(c, p + ".norm2", &r->n2));
  RET_IF(pack_conv(c, p + ".conv2.weight", p + ".conv2.bias", false, false, &r->c2, 0, c->vae_exact));
  r->has_skip = c->raw.count(p + ".nin_shortcut.weight") > 0;
  if (r->has_skip) RET_IF(pack_conv(c, p + ".nin_shortcut.weight", p + ".nin_shortcut.bias", false, false, &r->skip, 0, c->vae_exact));
  else if (r->cin != r->cout) return mvd_fail("VAE ResnetBlock changes width but has no nin_shortcut (conv_shortcut is not used by the reference config)");
  return 0;
}

__global__ void vae_fold_v_bias_kernel(const float* __restrict__ wp, const float* __restrict__ bv, const float* __restrict__ bp,
                                       int C, float* __restrict__ out) {
  // out = W_p b_v + b_p : softmax rows sum to 1, so attention(v + b_v) = attention(v) + b_v
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C) return;
  double acc = bp[i];
  for (int j = 0; j < C; ++j) acc += (double)wp[(long)i * C + j] * (double)bv[j];
  out[i] = (float)acc;
}

int build_vae_attn(mvd_ctx* c, const std::string& p, VaeAttnW* a) {
  RET_IF(load_norm(c, p + ".norm", &a->norm));
  RET_IF(pack_conv(c, p + ".q.weight", p + ".q.bias", false, false, &a->q, 0, c->vae_exact));
  RET_IF(pack_conv(c, p + ".k.weight", p + ".k.bias", false, false, &a->k, 0, c->vae_exact));
  RET_IF(pack_conv(c, p + ".v.weight", "", false, false, &a->v, 0, c->vae_exact));
  RET_IF(pack_conv(c, p + ".proj_out.weight", "", false, false, &a->proj, 0, c->vae_exact));
  RawTensor *wp, *bv, *bp;
  RET_IF(get_raw(c, p + ".proj_out.weight", &wp));
  RET_IF(get_raw(c, p + ".v.bias", &bv));
  RET_IF(get_raw(c, p + ".proj_out.bias", &bp));
  const int C = (int)bp->numel;
  RET_IF(dmalloc(c, (void**)&a->proj.bias, C * sizeof(float)));
  hipLaunchKernelGGL(vae_fold_v_bias_kernel, dim3(cdiv(C, 128)), dim3(128), 0, 0, wp->d, bv->d, bp->d, C, a->proj.bias);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// CLIP vision tower (openai/CLIP VisionTransformer key names under clip_image_encoder.model.visual.)
int pack_rows(mvd_ctx* c, const float* src, int N, int Cin, int cin_src, ConvW* o) {
  engine_build_rotate(c);
  o->N = N;
  o->Cin = Cin;
  o->taps = 1;
  RET_IF(dmalloc(c, (void**)&o->w, (size_t)N * Cin * sizeof(half_t)));
  return launch_pack_weight(src, N, Cin, 1, 0, 0, o->w, c->bs, cin_src);
}

int build_clip(mvd_ctx* c) {
  const std::string P = "clip_image_encoder.model.visual.";
  ClipW& k = c->clip;
  RawTensor *c1, *pos, *pj;
  RET_IF(get_raw(c, P + "conv1.weight", &c1));
  RET_IF(get_raw(c, P + "positional_embedding", &pos));
  RET_IF(get_raw(c, P + "proj", &pj));
  if (c1->shape.size() != 4 || c1->shape[1] != 3 || c1->shape[2] != c1->shape[3] || pos->shape.size() != 2 ||
      pj->shape.size() != 2)
    return mvd_fail("CLIP: unexpected conv1 / positional_embedding / proj shapes");
  k.width = (int)c1->shape[0];
  k.patch = (int)c1->shape[2];
  k.T = (int)pos->shape[0];
  k.embed = (int)pj->shape[1];
  int grid = 1;
  while (grid * grid < k.T - 1) ++grid;
  if (grid * grid != k.T - 1 || (int)pos->shape[1] != k.width || (int)pj->shape[0] != k.width)
    return mvd_fail("CLIP: positional_embedding must hold 1 + grid^2 tokens of `width` channels");
  k.image = grid * k.patch;
  k.heads = k.width / 64;  // clip/model.py build_model: vision_heads = vision_width // 64
  if (k.width % 64 || k.width > 1536) return mvd_fail("CLIP: width must be a multiple of 64, at most 1536");
  k.Tp = (k.T + 7) & ~7;
  const int K = 3 * k.patch * k.patch;
  k.Kp = (K + 7) & ~7;
  RET_IF(pack_rows(c, c1->d, k.width, k.Kp, K, &k.conv1));
  RET_IF(copy_f32(c, P + "class_embedding", &k.cls));
  RET_IF(copy_f32(c, P + "positional_embedding", &k.pos));
  RET_IF(load_norm(c, P + "ln_pre", &k.ln_pre));
  RET_IF(load_norm(c, P + "ln_post", &k.ln_post));
  {  // proj [width][embed] -> [embed][width]
    ConvW t;
    RET_IF(pack_conv(c, P + "proj", "", true, false, &t));
    k.proj.w = t.w;
    k.proj.N = k.embed;
    k.proj.K = k.width;
  }
  k.layers = 0;
  while (c->raw.count(P + "transformer.resblocks." + std::to_string(k.layers) + ".attn.in_proj_weight")) ++k.layers;
  if (k.layers < 1) return mvd_fail("CLIP: no transformer.resblocks uploaded");
  const int w = k.width;
  k.blk.assign(k.layers, ClipLayerW());
  for (int i = 0; i < k.layers; ++i) {
    const std::string L = P + "transformer.resblocks." + std::to_string(i);
    ClipLayerW& b = k.blk[i];
    RawTensor *iw, *ib, *ow, *ob, *fb;
    RET_IF(get_raw(c, L + ".attn.in_proj_weight", &iw));
    RET_IF(get_raw(c, L + ".attn.in_proj_bias", &ib));
    RET_IF(get_raw(c, L + ".attn.out_proj.weight", &ow));
    RET_IF(get_raw(c, L + ".attn.out_proj.bias", &ob));
    RET_IF(get_raw(c, L + ".mlp.c_fc.bias", &fb));
    if ((long)iw->numel != 3L * w * w || (long)ib->numel != 3L * w) return mvd_fail("CLIP: in_proj shape");
    RET_IF(load_norm(c, L + ".ln_1", &b.ln1));
    RET_IF(load_norm(c, L + ".ln_2", &b.ln2));
    RET_IF(pack_rows(c, iw->d, 3 * w, w, w, &b.qkv));  // in_proj_weight is q | k | v stacked already
    RET_IF(dmalloc(c, (void**)&b.qkv.bias, (size_t)3 * w * sizeof(float)));
    HIP_CHECK_RET(hipMemcpyAsync(b.qkv.bias, ib->d, (size_t)2 * w * sizeof(float), hipMemcpyDeviceToDevice, 0));
    HIP_CHECK_RET(hipMemset(b.qkv.bias + 2 * w, 0, (size_t)w * sizeof(float)));  // b_v lives in out.bias
    RET_IF(pack_conv(c, L + ".attn.out_proj.weight", "", false, false, &b.out));
    RET_IF(dmalloc(c, (void**)&b.out.bias, (size_t)w * sizeof(float)));
    hipLaunchKernelGGL(vae_fold_v_bias_kernel, dim3(cdiv(w, 128)), dim3(128), 0, 0, ow->d, ib->d + 2 * w, ob->d, w, b.out.bias);
    HIP_CHECK_RET(hipGetLastError());
    RET_IF(pack_conv(c, L + ".mlp.c_fc.weight", "", false, false, &b.fc));
    RET_IF(dmalloc(c, (void**)&b.fc.bias, fb->numel * sizeof(float)));
    RET_IF(launch_scale_copy(fb->d, fb->numel, 1.702f, b.fc.bias, 0));
    RET_IF(pack_conv(c, L + ".mlp.c_proj.weight", L + ".mlp.c_proj.bias", false, false, &b.proj));
  }
  k.present = true;
  return 0;
}

int build_vae_encoder(mvd_ctx* c) {
  const std::string V = "first_stage_model.", E = V + "encoder.";
  VaeEncW& v = c->vae_enc;
  RawTensor *ci, *q;
  RET_IF(get_raw(c, E + "conv_in.weight", &ci));
  RET_IF(get_raw(c, V + "quant_conv.weight", &q));
  v.in_ch = (int)ci->shape[1];
  v.mom = (int)q->shape[0];
  if (v.in_ch > 8 || (v.mom & 3) || (q->shape[1] & 7)) return mvd_fail("VAE encoder: in_channels <= 8, moments % 4 == 0 expected");
  RET_IF(pack_conv(c, E + "conv_in.weight", E + "conv_in.bias", false, false, &v.conv_in, 8, c->vae_exact));
  v.nlev = 0;
  while (c->raw.count(E + "down." + std::to_string(v.nlev) + ".block.0.conv1.weight")) ++v.nlev;
  if (v.nlev < 1) return mvd_fail("VAE encoder: no down blocks uploaded");
  v.down.assign(v.nlev, {});
  v.down_conv.assign(v.nlev, ConvW());
  for (int l = 0; l < v.nlev; ++l) {
    const std::string L = E + "down." + std::to_string(l);
    for (int i = 0; c->raw.count(L + ".block." + std::to_string(i) + ".conv1.weight"); ++i) {
      VaeResW r;
      RET_IF(build_vae_res(c, L + ".block." + std::to_string(i), &r));
      v.down[l].push_back(r);
    }
    if (l < v.nlev - 1) RET_IF(pack_conv(c, L + ".downsample.conv.weight", L + ".downsample.conv.bias", false, false, &v.down_conv[l], 0, c->vae_exact));
  }
  RET_IF(build_vae_res(c, E + "mid.block_1", &v.mid1));
  RET_IF(build_vae_attn(c, E + "mid.attn_1", &v.attn));
  RET_IF(build_vae_res(c, E + "mid.block_2", &v.mid2));
  RET_IF(load_norm(c, E + "norm_out", &v.norm_out));
  RET_IF(pack_conv(c, E + "conv_out.weight", E + "conv_out.bias", false, false, &v.conv_out, 0, c->vae_exact));
  RET_IF(pack_conv(c, V + "quant_conv.weight", V + "quant_conv.bias", false, false, &v.quant, 0, c->vae_exact));
  v.present = true;
  return 0;
}

int build_vae(mvd_ctx* c) {
  const std::string V = "first_stage_model.", D = V + "decoder.";
  VaeW& v = c->vae;
  RawTensor *co, *ci, *pq;
  RET_IF(get_raw(c, D + "conv_out.weight", &co));
  RET_IF(get_raw(c, D + "conv_in.weight", &ci));
  RET_IF(get_raw(c, V + "post_quant_conv.weight", &pq));
  v.out_ch = (int)co->shape[0];
  v.ch = (int)co->shape[1];
  v.block_in = (int)ci->shape[0];
  v.zc = (int)ci->shape[1];
  v.embed = (int)pq->shape[1];
  if (v.out_ch > 4 || v.zc > 8 || (v.zc & 3) || v.embed > 8)
    return mvd_fail("VAE decoder: out_ch <= 4, z_channels in {4, 8} and embed_dim <= 8 expected");
  RET_IF(pack_conv(c, V + "post_quant_conv.weight", V + "post_quant_conv.bias", false, false, &v.post_quant, 8, c->vae_exact));
  RET_IF(pack_conv(c, D + "conv_in.weight", D + "conv_in.bias", false, false, &v.conv_in, 8, c->vae_exact));
  RET_IF(build_vae_res(c, D + "mid.block_1", &v.mid1));
  RET_IF(build_vae_res(c, D + "mid.block_2", &v.mid2));
  RET_IF(build_vae_attn(c, D + "mid.attn_1", &v.attn));
  v.nlev = 0;
  while (c->raw.count(D + "up." + std::to_string(v.nlev) + ".block.0.conv1.weight")) ++v.nlev;
  if (v.nlev < 1) return mvd_fail("VAE decoder: no up blocks uploaded");
  v.up.assign(v.nlev, {});
  v.up_conv.assign(v.nlev, ConvW());
  for (int l = 0; l < v.nlev; ++l) {
    const std::string L = D + "up." + std::to_string(l);
    for (int i = 0; c->raw.count(L + ".block." + std::to_string(i) + ".conv1.weight"); ++i) {
      VaeResW r;
      RET_IF(build_vae_res(c, L + ".block." + std::to_string(i), &r));
      v.up[l].push_back(r);
    }
    if (l > 0) {
      ConvW& uc = v.up_conv[l];
      RET_IF(pack_conv(c, L + ".upsample.conv.weight", L + ".upsample.conv.bias", false, false, &uc, 0, c->vae_exact));
      RawTensor* r;
      RET_IF(get_raw(c, L + ".upsample.conv.weight", &r));
      if (!c->vae_exact) {  // the parity-folded form (pre-summed taps) exists for plain fp16 weights only
        RET_IF(dmalloc(c, (void**)&uc.w_up, (size_t)16 * uc.N * uc.Cin * sizeof(half_t)));
        RET_IF(launch_pack_upconv_weight(r->d, uc.N, uc.Cin, uc.w_up, c->bs));
      }
    }
  }
  RET_IF(load_norm(c, D + "norm_out", &v.norm_out));
  {  // conv_out with N padded to 4 (16-byte epilogue stores): a zero fourth filter and bias
    RawTensor pad, padb, *b;
    RET_IF(get_raw(c, D + "conv_out.bias", &b));
    const size_t per = co->numel / co->shape[0];
    pad.shape = {4, co->shape[1], co->shape[2], co->shape[3]};
    pad.numel = 4 * per;
    HIP_CHECK_RET(hipMalloc((void**)&pad.d, pad.numel * sizeof(float)));
    HIP_CHECK_RET(hipMemset(pad.d, 0, pad.numel * sizeof(float)));
    HIP_CHECK_RET(hipMemcpyAsync(pad.d, co->d, co->numel * sizeof(float), hipMemcpyDeviceToDevice, 0));
    padb.shape = {4};
    padb.numel = 4;
    HIP_CHECK_RET(hipMalloc((void**)&padb.d, 4 * sizeof(float)));
    HIP_CHECK_RET(hipMemset(padb.d, 0, 4 * sizeof(float)));
    HIP_CHECK_RET(hipMemcpyAsync(padb.d, b->d, b->numel * sizeof(float), hipMemcpyDeviceToDevice, 0));
    c->raw[D + "conv_out.weight.pad4"] = pad;
    c->raw[D + "conv_out.bias.pad4"] = padb;
    RET_IF(pack_conv(c, D + "conv_out.weight.pad4", D + "conv_out.bias.pad4", false, false, &v.conv_out, 0, c->vae_exact));
  }
  v.present = true;
  return 0;
}

// Extended-precision policy (ConvW::xp), level = mvd_set_precision_level (MVD_XP in the environment overrides it):
//   0 none;  1 the output conv and conv_in;
//   2 (default) + the cheap layers of the LAST output block: ResBlock skip conv, SpatialTransformer proj_in / proj_out, its
//     DepthTransformer's folded attention projections and two 3x3 convs;
//   3 + that block's ResBlock 3x3 convs;   4 + the level-2 set of the block before it;
//   5 every such layer (3x3 convs included) of ALL full-resolution output blocks;   6 + of the full-resolution input blocks.
// Share of the end-to-end error variance removed on the full-width UNet (tools/precision_probe4.py; measured on the GPU in
// DESIGN.md): 25 % / 52 % / 60 % / 70 % for levels 1-4, at +0.6 % / +2.2 % / +3.6 % / +5.1 % step time.
// The re-pack needs the fp32 source tensors, so it runs inside finalize before they are released.
int repack_xp(mvd_ctx* c, const std::string& wkey, ConvW* o, int cin_pad = 0) {
  if (o->xp) return 0;
  ConvW n;
  RET_IF(pack_conv(c, wkey, "", false, false, &n, cin_pad, true));
  n.bias = o->bias;
  n.bkey = o->bkey;
  n.w_up = nullptr;
  *o = n;
  return 0;
}

int xp_cond(mvd_ctx* c, CondW& d) {
  if (d.wqk.xp) return 0;
  const int heads = 4, hd = d.Cc / 2;
  RET_IF(repack_xp(c, d.key + ".proj_out.2.weight", &d.conv1));
  RET_IF(repack_xp(c, d.key + ".proj_out.5.weight", &d.conv2));
  RawTensor *wq, *wk, *wv, *wo;
  RET_IF(get_raw(c, d.key + ".depth_attn.to_q.weight", &wq));
  RET_IF(get_raw(c, d.key + ".depth_attn.to_k.weight", &wk));
  RET_IF(get_raw(c, d.key + ".depth_attn.to_v.weight", &wv));
  RET_IF(get_raw(c, d.key + ".depth_attn.to_out.weight", &wo));
  const size_t nf = (size_t)heads * d.Cc * d.I;
  // fold scratch: an allocation of the build like any other (a re-pack replays it: no hipMalloc / hipDeviceSynchronize / hipFree
  // per training step); both folds and their packs run in stream order
  float* tmp = nullptr;
  RET_IF(dmalloc(c, (void**)&tmp, nf * sizeof(float)));
  auto fold = [&](ConvW& w, int N, int Cl, bool qk) -> int {
    if (qk) RET_IF(launch_fold_qk(wq->d, wk->d, heads, hd, d.Cc, d.I, 1.0f / sqrtf((float)hd), nullptr, c->bs, tmp));
    else RET_IF(launch_fold_ov(wo->d, wv->d, heads, hd, d.Cc, d.I, nullptr, c->bs, tmp));
    w.N = N;
    w.cin_l = Cl;
    w.Cin = 3 * Cl;
    w.xp = 1;
    w.taps = 1;
    RET_IF(dmalloc(c, (void**)&w.w, (size_t)N * 3 * Cl * sizeof(half_t)));
    return launch_pack_weight(tmp, N, 3 * Cl, 1, 0, 0, w.w, c->bs, Cl, 1);
  };
  RET_IF(fold(d.wqk, heads * d.Cc, d.I, true));
  RET_IF(fold(d.wov, d.I, heads * d.Cc, false));
  RET_IF(dmalloc(c, (void**)&d.relu_beta, (size_t)3 * heads * d.Cc * sizeof(half_t)));
  return launch_relu_beta_tile(d.gn_ctx.b, d.Cc, heads, d.relu_beta, c->bs, 1);
}

int xp_ops(mvd_ctx* c, const std::vector<UOp>& ops, bool convs3) {
  for (const UOp& op : ops) {
    if (op.kind == OP_RES) {
      ResW& r = c->res[op.idx];
      if (r.has_skip) RET_IF(repack_xp(c, r.key + ".skip_connection.weight", &r.skip));
      if (convs3) {
        RET_IF(repack_xp(c, r.key + ".in_layers.2.weight", &r.c1));
        RET_IF(repack_xp(c, r.key + ".out_layers.3.weight", &r.c2));
      }
    } else if (op.kind == OP_ST) {
      STW& t = c->st[op.idx];
      RET_IF(repack_xp(c, t.key + ".proj_in.weight", &t.proj_in));
      RET_IF(repack_xp(c, t.key + ".proj_out.weight", &t.proj_out));
    }
  }
  return 0;
}

// The 3x3 ResBlock convolutions that run at resolutions divisible by 16 also get their weights as a conv3x fragment stream
// (k_conv3x.hip); the level of a convolution is not recorded in ResW, its width is: levels 0 .. l16 have at most
// model_channels * channel_mult[l16] output channels, where l16 is the last level whose resolution is a multiple of 16.
int build_conv3x_streams(mvd_ctx* c) {
  static const bool off = getenv("MVD_NO_CONV3X") != nullptr;
  if (off || !c->has_unet) return 0;
  // conv3x takes resolutions divisible by 16 and 8 x 8 images (conv3x_eligible): only convolutions that RUN at such a resolution
  // get a stream (the plan records it in ResW::res / CondW::res) -- with channel_mult (1, 2, 4, 4) the 4 x 4 level's 1280-wide
  // ResBlocks share their width with the 8 x 8 level's, and a stream for them would be ~29 MB each, packed (and re-packed every
  // training step) for a kernel that never takes them
  auto takes = [](int res) { return (res % 16 == 0 && res >= 16) || res == 8; };
  std::vector<ConvW*> cs;
  for (ResW& r : c->res) {
    if (!takes(r.res)) continue;
    cs.push_back(&r.c1);
    cs.push_back(&r.c2);
  }
  for (CondW& d : c->conds) {  // the DepthTransformers' 3x3 output convolutions (attention.py:66-73)
    if (!takes(d.res)) continue;
    cs.push_back(&d.conv1);
    cs.push_back(&d.conv2);
  }
  // the Upsample convolution that produces 8 x 8 images (openaimodel.py:112-117): few pixels, so instead of the parity-folded form
  // it runs as nearest-upsample + cast (one small launch) and conv3x over the 8 x 8 images (unet_do_op)
  for (ConvW& w : c->convs)
    if (w.res_out == 8 && !w.xp) cs.push_back(&w);
  for (ConvW* w : cs) {
    if (w->taps != 9 || w->Cin % 64) continue;
    const int bn = w->N % 160 == 0 ? 160 : (w->N % 128 == 0 ? 128 : 0);
    if (!bn) continue;
    engine_build_rotate(c);
    RET_IF(dmalloc(c, (void**)&w->wx, conv3x_stream_halfs(w->N, w->Cin, bn) * sizeof(half_t)));
    RET_IF(conv3x_pack(w->w, w->N, w->Cin, bn, w->wx, c->bs));
    w->wx_bn = bn;
  }
  return engine_build_join(c);
}

// The row-local halves of the transformer blocks as fragment streams (k_rowchain.hip): to_out | LayerNorm3-folded FF1 | FF2 |
// proj_out for the row-chain kernel, proj_in | LayerNorm1-folded q|k|v for the row head (C = 320).  Packed AFTER the
// extended-precision policy: a block whose proj_in / proj_out run in extended precision gets the (hi, lo) forms of those sections.
// Inference contexts only (a training context re-packs every step and never takes these paths).
int build_rowchain_streams(mvd_ctx* c) {
  if (!c->has_unet || c->train_mode) return 0;
  for (STW& st : c->st) {
    const int C = st.C;
    if (!rc_supported_c(C)) continue;
    const std::string& p = st.key;
    const std::string t = p + ".transformer_blocks.0";
    engine_build_rotate(c);
    RawTensor *wao, *g3, *b3, *w1, *b1, *w2, *b2, *wpo;
    RET_IF(get_raw(c, t + ".attn1.to_out.0.weight", &wao));
    RET_IF(get_raw(c, t + ".norm3.weight", &g3));
    RET_IF(get_raw(c, t + ".norm3.bias", &b3));
    RET_IF(get_raw(c, t + ".ff.net.0.proj.weight", &w1));
    RET_IF(get_raw(c, t + ".ff.net.0.proj.bias", &b1));
    RET_IF(get_raw(c, t + ".ff.net.2.weight", &w2));
    RET_IF(get_raw(c, t + ".ff.net.2.bias", &b2));
    RET_IF(get_raw(c, p + ".proj_out.weight", &wpo));
    if (wao->numel != (size_t)C * C || w1->numel != (size_t)8 * C * C || w2->numel != (size_t)4 * C * C || wpo->numel != (size_t)C * C)
      return mvd_fail("build_rowchain_streams: transformer block weights do not have the row-chain kernel's shapes");
    st.rc_po = st.proj_out.xp ? 2 : 1;
    // a width without the extended-precision form (C = 128 / 256): the stream carries no proj_out section, the block runs the
    // (1, 0) form + the separate extended-precision proj_out GEMM (unet_do_st: rc_po false)
    if (!rowchain_form_instantiated(C, 1, st.rc_po)) st.rc_po = 0;
    float* tmp = nullptr;
    RET_IF(dmalloc(c, (void**)&tmp, (size_t)8 * C * sizeof(float)));
    RET_IF(dmalloc(c, (void**)&st.rc_stream, rowchain_stream_halfs(C, 1, st.rc_po) * sizeof(half_t)));
    RcWeights rw;
    rw.w_ao = wao->d; rw.ln_g = g3->d; rw.ln_b = b3->d; rw.w1 = w1->d; rw.b1 = b1->d; rw.w2 = w2->d; rw.b2 = b2->d; rw.w_po = wpo->d;
    RET_IF(rowchain_pack(rw, C, 1, st.rc_po, tmp, st.rc_stream, c->bs));
    if (C == RH_C) {
      RawTensor *wpi, *g1, *b1n, *q, *k, *v;
      RET_IF(get_raw(c, p + ".proj_in.weight", &wpi));
      RET_IF(get_raw(c, t + ".norm1.weight", &g1));
      RET_IF(get_raw(c, t + ".norm1.bias", &b1n));
      RET_IF(get_raw(c, t + ".attn1.to_q.weight", &q));
      RET_IF(get_raw(c, t + ".attn1.to_k.weight", &k));
      RET_IF(get_raw(c, t + ".attn1.to_v.weight", &v));
      if (wpi->numel != (size_t)C * C || q->numel != (size_t)C * C) return mvd_fail("build_rowchain_streams: unexpected proj_in / to_q shape");
      st.rh_xp = st.proj_in.xp ? 1 : 0;
      RET_IF(dmalloc(c, (void**)&st.rh_stream, rowhead_stream_halfs(st.rh_xp) * sizeof(half_t)));
      RhWeights hw;
      hw.w_pi = wpi->d; hw.ln_g = g1->d; hw.ln_b = b1n->d; hw.w_q = q->d; hw.w_k = k->d; hw.w_v = v->d;
      RET_IF(rowhead_pack(hw, st.rh_xp, tmp, st.rh_stream, c->bs));
    }
  }
  return engine_build_join(c);
}

// FF2 + proj_out folded into one GEMM per transformer block (STW::ffp).  After the extended-precision policy: a block whose proj_out
// runs in extended precision keeps the two layers.  Inference contexts only (a training context re-packs every step and its
// backward pass needs both layers).
int build_ffp(mvd_ctx* c) {
  static const bool off = getenv("MVD_NO_FFP") != nullptr;
  if (off || !c->has_unet || c->train_mode) return 0;
  for (STW& st : c->st) {
    if (st.proj_out.xp || (st.C & 7)) continue;
    const int C = st.C;
    const std::string t = st.key + ".transformer_blocks.0";
    RawTensor *w2, *b2, *wpo, *bpo;
    RET_IF(get_raw(c, t + ".ff.net.2.weight", &w2));
    RET_IF(get_raw(c, t + ".ff.net.2.bias", &b2));
    RET_IF(get_raw(c, st.key + ".proj_out.weight", &wpo));
    RET_IF(get_raw(c, st.key + ".proj_out.bias", &bpo));
    if (w2->numel != (size_t)4 * C * C || wpo->numel != (size_t)C * C || b2->numel != (size_t)C || bpo->numel != (size_t)C) continue;
    engine_build_rotate(c);
    ConvW& f = st.ffp;
    f.N = C; f.Cin = 5 * C; f.cin_l = 5 * C; f.taps = 1;
    half_t* tmp16 = nullptr;
    float* tmp32 = nullptr;
    RET_IF(dmalloc(c, (void**)&f.w, (size_t)C * 5 * C * sizeof(half_t)));
    RET_IF(dmalloc(c, (void**)&f.bias, (size_t)C * sizeof(float)));
    RET_IF(dmalloc(c, (void**)&tmp16, (size_t)C * C * sizeof(half_t)));
    RET_IF(dmalloc(c, (void**)&tmp32, (size_t)C * sizeof(float)));
    // columns [0, 4C): W_po W_2  (M = C rows of W_po, K = C, N = 4C columns of W_2), fp64 accumulation, rows 5C halfs apart
    RET_IF(launch_fold_mm(wpo->d, C, w2->d, 4 * C, C, 4 * C, C, f.w, 5 * C, nullptr, c->bs));
    // columns [4C, 5C): W_po itself
    RET_IF(launch_f32_to_f16(wpo->d, tmp16, (size_t)C * C, c->bs));
    HIP_CHECK_RET(hipMemcpy2DAsync(f.w + 4 * C, (size_t)5 * C * sizeof(half_t), tmp16, (size_t)C * sizeof(half_t), (size_t)C * sizeof(half_t), C,
                                   hipMemcpyDeviceToDevice, c->bs));
    // bias: W_po b2 + b_po
    RET_IF(launch_fold_mm(wpo->d, C, b2->d, 1, C, 1, C, nullptr, 1, tmp32, c->bs));
    RET_IF(launch_add_rows(f.bias, tmp32, bpo->d, (size_t)C, c->bs));
  }
  return engine_build_join(c);
}

int apply_xp_policy(mvd_ctx* c) {
  const int lvl = getenv("MVD_XP") ? atoi(getenv("MVD_XP")) : c->precision_level;
  if (lvl <= 0 || !c->has_unet) return 0;
  const std::string U = "model.diffusion_model.";
  RET_IF(repack_xp(c, U + "out.2.weight", &c->out_conv));
  RET_IF(repack_xp(c, U + "input_blocks.0.0.weight", &c->convs[c->in_blocks[0][0].idx], 8));
  if (lvl < 2) return 0;
  // Ladder (round 6: ordered by error removed per unit of time, profiles/r06_k_xp_levels.txt): 2 = the last output block's cheap
  // layers (skip conv, transformer proj_in / proj_out) and its DepthTransformer; 3 = the same for the block before it (guided eps
  // -12 % for +1.2 % of the step: the default); 4 = + the last block's 3x3 ResBlock convolutions (-7 % more for +2.4 %);
  // 5 = three blocks with their 3x3 convolutions; 6 = + the full-resolution input blocks.
  // (Until round 5 level 3 was "level 2 + the last block's 3x3 convolutions": the costlier step came first.)
  const int nb = (int)c->out_blocks.size();
  const int nblk = lvl >= 5 ? 3 : (lvl >= 3 ? 2 : 1);
  for (int k = 0; k < nblk && k < nb; ++k) {
    const int bi = nb - 1 - k;
    RET_IF(xp_ops(c, c->out_blocks[bi], lvl >= 5 || (lvl >= 4 && k == 0)));
    if (bi >= 3 && 1 + (bi - 3) < (int)c->conds.size()) RET_IF(xp_cond(c, c->conds[1 + (bi - 3)]));  // attention.py:100
  }
  if (lvl >= 6)
    for (int j = 1; j <= 2 && j < (int)c->in_blocks.size(); ++j) RET_IF(xp_ops(c, c->in_blocks[j], true));
  return 0;
}

namespace {

// ---------------- UNet plan (openaimodel.py:535-720) ----------------
int build_unet_section(mvd_ctx* c) {
  const std::string U = "model.diffusion_model.";
  const mvd_unet_config& u = c->u;
  const int mc = u.model_channels;
  const int temb = 4 * mc;
  c->res.clear();
  c->st.clear();
  c->convs.clear();
  c->conds.clear();
  c->in_blocks.clear();
  c->out_blocks.clear();
  c->mid_block.clear();
  c->emb_total = 0;
  c->a2_total = 0;
  RET_IF(pack_lin(c, U + "time_embed.0.weight", U + "time_embed.0.bias", &c->te0));
  RET_IF(pack_lin(c, U + "time_embed.2.weight", U + "time_embed.2.bias", &c->te2));
  struct EmbPiece {
    std::string key;
    int cout;
  };
  std::vector<EmbPiece> emb_pieces;
  auto add_res = [&](const std::string& name, int cin, int cout, std::vector<UOp>& ops) -> int {
    ResW r;
    RET_IF(build_res(c, U + name, cin, cout, &r));
    r.emb_off = c->emb_total;
    c->emb_total += cout;
    emb_pieces.push_back({U + name + ".emb_layers.1", cout});
    c->res.push_back(r);
    ops.push_back({OP_RES, (int)c->res.size() - 1, cin, cout});
    return 0;
  };
  std::vector<std::string> a2v_keys;
  auto add_st = [&](const std::string& name, int C, std::vector<UOp>& ops) -> int {
    STW s;
    RET_IF(build_st(c, U + name, C, &s));
    s.a2_off = c->a2_total;
    c->a2_total += C;
    a2v_keys.push_back(U + name + ".transformer_blocks.0.attn2");
    c->st.push_back(s);
    ops.push_back({OP_ST, (int)c->st.size() - 1, C, C});
    return 0;
  };
  auto add_conv = [&](const std::string& wkey, int kind, int cin, int cout, std::vector<UOp>& ops) -> int {
    ConvW w;
    RET_IF(pack_conv(c, U + wkey + ".weight", U + wkey + ".bias", false, false, &w));
    if (kind == OP_UP && w.taps == 9) {
      RawTensor* r;
      RET_IF(get_raw(c, U + wkey + ".weight", &r));
      RET_IF(dmalloc(c, (void**)&w.w_up, (size_t)16 * w.N * w.Cin * sizeof(half_t)));
      RET_IF(launch_pack_upconv_weight(r->d, w.N, w.Cin, w.w_up, c->bs));
    }
    if (kind == OP_CONV_IN && w.taps == 9 && (cin == 4 || cin == 8) && cout <= 512) {
      RET_IF(copy_f32(c, U + wkey + ".weight", &w.w32));  // training mode: a view of the master arena, always current
      w.cin_src = cin;
    }
    c->convs.push_back(w);
    ops.push_back({kind, (int)c->convs.size() - 1, cin, cout});
    return 0;
  };
  {
    std::vector<UOp> ops;
    RET_IF(add_conv("input_blocks.0.0", OP_CONV_IN, u.in_channels, mc, ops));
    c->in_blocks.push_back(ops);
  }
  std::vector<int> chans{mc};
  int ch = mc, ds = 1, bi = 1;
  for (int level = 0; level < 4; ++level) {
    const int mult = u.channel_mult[level];
    for (int nr = 0; nr < u.num_res_blocks; ++nr) {
      std::vector<UOp> ops;
      const std::string b = "input_blocks." + std::to_string(bi);
      RET_IF(add_res(b + ".0", ch, mult * mc, ops));
      c->res.back().res = u.image_size / ds;
      ch = mult * mc;
      if (u.attention_levels & ds) RET_IF(add_st(b + ".1", ch, ops));
      c->in_blocks.push_back(ops);
      chans.push_back(ch);
      ++bi;
    }
    if (level != 3) {
      std::vector<UOp> ops;
      RET_IF(add_conv("input_blocks." + std::to_string(bi) + ".0.op", OP_DOWN, ch, ch, ops));
      c->in_blocks.push_back(ops);
      chans.push_back(ch);
      ++bi;
      ds *= 2;
    }
  }
  RET_IF(add_res("middle_block.0", ch, ch, c->mid_block));
  c->res.back().res = u.image_size / ds;
  RET_IF(add_st("middle_block.1", ch, c->mid_block));
  RET_IF(add_res("middle_block.2", ch, ch, c->mid_block));
  c->res.back().res = u.image_size / ds;
  bi = 0;
  for (int level = 3; level >= 0; --level) {
    const int mult = u.channel_mult[level];
    for (int i = 0; i <= u.num_res_blocks; ++i) {
      const int ich = chans.back();
      chans.pop_back();
      std::vector<UOp> ops;
      const std::string b = "output_blocks." + std::to_string(bi);
      RET_IF(add_res(b + ".0", ch + ich, mc * mult, ops));
      c->res.back().res = u.image_size / ds;
      ch = mc * mult;
      int j = 1;
      if (u.attention_levels & ds) {
        RET_IF(add_st(b + "." + std::to_string(j), ch, ops));
        ++j;
      }
      if (level && i == u.num_res_blocks) {
        RET_IF(add_conv(b + "." + std::to_string(j) + ".conv", OP_UP, ch, ch, ops));
        ds /= 2;
        c->convs.back().res_out = u.image_size / ds;
      }
      c->out_blocks.push_back(ops);
      ++bi;
    }
  }
  RET_IF(load_norm(c, U + "out.0", &c->out_norm));
  RET_IF(pack_conv(c, U + "out.2.weight", U + "out.2.bias", false, false, &c->out_conv));
  if (c->out_conv.taps == 9 && c->out_conv.N <= 4 && mc % 32 == 0 && mc <= 512) {  // inference: the exact fp32 head (launch_out_conv_f32)
    RET_IF(copy_f32(c, U + "out.2.weight", &c->out_conv.w32));
    c->out_conv.cin_src = mc;
  }
  // all ResBlock emb projections as ONE [emb_total][temb] linear (openaimodel.py:219-225)
  c->emb_all.N = c->emb_total;
  c->emb_all.K = temb;
  RET_IF(dmalloc(c, (void**)&c->emb_all.w, (size_t)c->emb_total * temb * sizeof(half_t)));
  RET_IF(dmalloc(c, (void**)&c->emb_all.bias, (size_t)c->emb_total * sizeof(float)));
  {
    int off = 0;
    for (auto& pz : emb_pieces) {
      RawTensor *w, *b;
      RET_IF(get_raw(c, pz.key + ".weight", &w));
      RET_IF(get_raw(c, pz.key + ".bias", &b));
      RET_IF(launch_f32_to_f16(w->d, c->emb_all.w + (size_t)off * temb, w->numel, c->bs));
      HIP_CHECK_RET(hipMemcpyAsync(c->emb_all.bias + off, b->d, pz.cout * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
      off += pz.cout;
    }
  }
  // attn2 sees one CLIP token, so softmax == 1 and the block reduces to to_out(to_v(ctx)): fold W_o W_v per
  // SpatialTransformer and stack them into one [a2_total][context_dim] matrix (one GEMM per UNet forward)
  c->a2_all.N = c->a2_total;
  c->a2_all.Cin = u.context_dim;
  c->a2_all.taps = 1;
  RET_IF(dmalloc(c, (void**)&c->a2_all.w, (size_t)c->a2_total * u.context_dim * sizeof(half_t)));
  RET_IF(dmalloc(c, (void**)&c->a2_all.bias, (size_t)c->a2_total * sizeof(float)));
  {
    size_t off = 0;
    for (auto& k : a2v_keys) {
      RawTensor *wv, *wo, *bo;
      RET_IF(get_raw(c, k + ".to_v.weight", &wv));
      RET_IF(get_raw(c, k + ".to_out.0.weight", &wo));
      RET_IF(get_raw(c, k + ".to_out.0.bias", &bo));
      const int C = (int)bo->numel;
      RET_IF(launch_fold_ov(wo->d, wv->d, 1, C, u.context_dim, C, c->a2_all.w + off * u.context_dim, c->bs));
      HIP_CHECK_RET(hipMemcpyAsync(c->a2_all.bias + off, bo->d, C * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
      off += C;
    }
  }
  // conditioning blocks (attention.py:97-115)
  {
    const int c2 = mc * u.channel_mult[2], c1 = mc * u.channel_mult[1], c0 = mc * u.channel_mult[0];
    const int* d = u.volume_dims;
    const int dims[10] = {c2, c2, c2, c2, c1, c1, c1, c0, c0, c0};
    const int ccs[10] = {d[3], d[2], d[2], d[1], d[1], d[1], d[0], d[0], d[0], d[0]};
    const int lvl[10] = {3, 2, 2, 1, 1, 1, 0, 0, 0, 0};  // the volume level each block attends to = the resolution it runs at
    c->conds.resize(10);
    RET_IF(build_cond(c, U + "middle_conditions", dims[0], ccs[0], &c->conds[0]));
    for (int k = 0; k < 9; ++k)
      RET_IF(build_cond(c, U + "output_conditions." + std::to_string(k), dims[k + 1], ccs[k + 1], &c->conds[k + 1]));
    for (int k = 0; k < 10; ++k) c->conds[k].res = u.image_size >> lvl[k];
    // stacked context projections per level (mvd_ctx::CtxGroup): [nblk * Cc][Cc] fp16 rows copied from the blocks' own packs,
    // gain / bias concatenated.  Plain (not extended-precision) projections only -- proj_context never is.
    c->ctx_groups.clear();
    for (int L = 0; L < 4; ++L) {
      mvd_ctx::CtxGroup gp;
      gp.level = L;
      for (int k = 0; k < 10; ++k)
        if (lvl[k] == L && gp.nblk < 4 && !c->conds[k].proj_ctx.xp && (gp.nblk == 0 || c->conds[k].Cc == gp.Cc)) {
          gp.Cc = c->conds[k].Cc;
          gp.cond[gp.nblk++] = k;
        }
      if (gp.nblk < 2) continue;
      const int Cc = gp.Cc, N = gp.nblk * Cc;
      gp.w.N = N; gp.w.Cin = Cc; gp.w.taps = 1;
      gp.gn.C = N;
      RET_IF(dmalloc(c, (void**)&gp.w.w, (size_t)N * Cc * sizeof(half_t)));
      RET_IF(dmalloc(c, (void**)&gp.gn.g, (size_t)N * sizeof(float)));
      RET_IF(dmalloc(c, (void**)&gp.gn.b, (size_t)N * sizeof(float)));
      for (int j = 0; j < gp.nblk; ++j) {
        const CondW& d = c->conds[gp.cond[j]];
        if (d.proj_ctx.N != Cc || d.proj_ctx.Cin != Cc) return mvd_fail("build_unet_section: unexpected proj_context shape");
        HIP_CHECK_RET(hipMemcpyAsync(gp.w.w + (size_t)j * Cc * Cc, d.proj_ctx.w, (size_t)Cc * Cc * sizeof(half_t), hipMemcpyDeviceToDevice, c->bs));
        HIP_CHECK_RET(hipMemcpyAsync(gp.gn.g + (size_t)j * Cc, d.gn_ctx.g, (size_t)Cc * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
        HIP_CHECK_RET(hipMemcpyAsync(gp.gn.b + (size_t)j * Cc, d.gn_ctx.b, (size_t)Cc * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
      }
      c->ctx_groups.push_back(gp);
    }
  }
  return 0;
}

// ---------------- Lightning-module step embedding (morphable_diffusion.py:452-458) ----------------
int build_step_section(mvd_ctx* c) {
  RET_IF(pack_lin(c, "time_embed.0.weight", "time_embed.0.bias", &c->step_te0));
  RET_IF(pack_lin(c, "time_embed.2.weight", "time_embed.2.bias", &c->step_te2));
  return 0;
}

// ---------------- mesh conditioner ----------------
int build_condnet_section(mvd_ctx* c) {
  const std::string S = "spatial_volume.";
  RET_IF(pack_conv(c, S + "target_encoder.init_conv.weight", S + "target_encoder.init_conv.bias", false, false,
                   &c->enc_init, 8));
  for (int i = 0; i < 3; ++i) {
    const std::string p = S + "target_encoder.out_conv" + std::to_string(i) + ".";
    EncBlockW& e = c->enc_blocks[i];
    RET_IF(pack_lin(c, p + "time_embed.weight", p + "time_embed.bias", &e.t));
    RET_IF(pack_lin(c, p + "view_embed.weight", p + "view_embed.bias", &e.v));
    RET_IF(load_norm(c, p + "conv.0", &e.n1));
    RET_IF(pack_conv(c, p + "conv.2.weight", p + "conv.2.bias", false, false, &e.c1));
    RET_IF(load_norm(c, p + "conv.3", &e.n2));
    RET_IF(pack_conv(c, p + "conv.5.weight", p + "conv.5.bias", false, false, &e.c2));
  }
  RET_IF(load_norm(c, S + "target_encoder.final_out.0", &c->enc_final_norm));
  RET_IF(pack_conv(c, S + "target_encoder.final_out.2.weight", S + "target_encoder.final_out.2.bias", false, false,
                   &c->enc_final));
  RET_IF(copy_f32(c, S + "smpl_feature_extractor.conv0.weight", &c->fuse_w));
  RET_IF(copy_f32(c, S + "smpl_feature_extractor.conv0.bias", &c->fuse_b));
  {
    const char* blk[9] = {"conv0", "conv0", "down0", "conv1", "conv1", "down1", "conv2", "conv2", "conv2"};
    const int slot[9] = {0, 3, 0, 0, 3, 0, 0, 3, 6};
    for (int i = 0; i < 9; ++i)
      RET_IF(build_sparse_layer(c, S + "xyzc_net.", blk[i], slot[i], blk[i][0] == 'd', &c->sparse[i]));
  }
  const std::string F = S + "frustum_volume_feats.";
  RET_IF(pack_conv(c, F + "conv0.weight", F + "conv0.bias", false, false, &c->fr_conv0));
  for (int i = 0; i < 6; ++i)
    RET_IF(build_frustum_block(c, F + "conv" + std::to_string(i + 1) + ".", "bn", false, (i % 2 == 0) ? 2 : 1,
                               &c->fr_blocks[i]));
  for (int i = 0; i < 3; ++i)
    RET_IF(build_frustum_block(c, F + "up" + std::to_string(i) + ".", "norm", true, 2, &c->fr_up[i]));
  {  // stack the per-block FiLM projections: [sum cin][time_dim] and [sum cin][view_dim]
    auto stack = [&](const std::vector<std::string>& prefixes, const char* wn, LinW* out, int K) -> int {
      int total = 0;
      std::vector<RawTensor*> ws, bs;
      for (auto& p : prefixes) {
        RawTensor *wt, *bt;
        RET_IF(get_raw(c, p + wn + ".weight", &wt));
        RET_IF(get_raw(c, p + wn + ".bias", &bt));
        ws.push_back(wt);
        bs.push_back(bt);
        total += (int)wt->shape[0];
      }
      out->N = total;
      out->K = K;
      RET_IF(dmalloc(c, (void**)&out->w, (size_t)total * K * sizeof(half_t)));
      RET_IF(dmalloc(c, (void**)&out->bias, (size_t)total * sizeof(float)));
      size_t off = 0;
      for (size_t i = 0; i < ws.size(); ++i) {
        RET_IF(launch_f32_to_f16(ws[i]->d, out->w + off * K, ws[i]->numel, c->bs));
        HIP_CHECK_RET(hipMemcpyAsync(out->bias + off, bs[i]->d, bs[i]->numel * sizeof(float), hipMemcpyDeviceToDevice, c->bs));
        off += ws[i]->shape[0];
      }
      return 0;
    };
    std::vector<std::string> fp, ep;
    int off = 0;
    for (int i = 0; i < 9; ++i) {
      fp.push_back(i < 6 ? F + "conv" + std::to_string(i + 1) + "." : F + "up" + std::to_string(i - 6) + ".");
      c->film_off[i] = off;
      off += i < 6 ? c->fr_blocks[i].cin : c->fr_up[i - 6].cin;
    }
    c->film_total = off;
    RET_IF(stack(fp, "t_conv", &c->film_t, c->v.time_dim));
    RET_IF(stack(fp, "v_conv", &c->film_v, c->v.view_dim));
    for (int i = 0; i < 3; ++i) ep.push_back(S + "target_encoder.out_conv" + std::to_string(i) + ".");
    RET_IF(stack(ep, "time_embed", &c->enc_t, c->v.time_dim));
    RET_IF(stack(ep, "view_embed", &c->enc_v, c->v.view_dim));
  }
  return 0;
}

// everything that engine_repack re-derives, in a fixed order (the allocation replay depends on it)
int build_hot_sections(mvd_ctx* c) {
  if (c->has_unet) {
    RET_IF(build_unet_section(c));
    RET_IF(apply_xp_policy(c));
    RET_IF(engine_build_join(c));  // the adjoint packs and the conv3x streams read the forward packs
    RET_IF(build_conv3x_streams(c));
    RET_IF(build_rowchain_streams(c));
    RET_IF(build_ffp(c));
    if (c->train_mode) RET_IF(engine_build_dgrad(c));
  }
  if (c->has_step) RET_IF(build_step_section(c));
  if (c->has_cond) {
    RET_IF(build_condnet_section(c));
    RET_IF(engine_build_join(c));
    if (c->train_mode) RET_IF(engine_build_dgrad_cond(c));
  }
  return 0;
}

}  // namespace

int engine_finalize(mvd_ctx* c) {
  const std::string U = "model.diffusion_model.";
  HIP_CHECK_RET(hipSetDevice(c->device));
  // sections are optional: a stand-alone UNet (YAML unet_config.target) uploads only its own keys
  const bool has_unet = c->raw.count(U + "time_embed.0.weight") > 0;
  const bool has_cond = c->raw.count("spatial_volume.target_encoder.init_conv.weight") > 0;
  const bool has_step = c->raw.count("time_embed.0.weight") > 0;
  const bool has_vae = c->raw.count("first_stage_model.decoder.conv_in.weight") > 0;
  const bool has_vae_enc = c->raw.count("first_stage_model.encoder.conv_in.weight") > 0;
  const bool has_clip = c->raw.count("clip_image_encoder.model.visual.conv1.weight") > 0;
  if (!has_unet && !has_cond && !has_vae && !has_vae_enc && !has_clip)
    return mvd_fail("finalize: no UNet, spatial_volume, first-stage or CLIP weights were uploaded");
  if (has_clip) RET_IF(build_clip(c));
  if (has_vae) RET_IF(build_vae(c));
  if (has_vae_enc) RET_IF(build_vae_encoder(c));
  c->has_unet = has_unet;
  c->has_step = has_step;
  c->has_cond = has_cond;
  if (c->train_mode) RET_IF(engine_train_setup(c));  // masters into the arena first: the packs below read them from there
  for (auto& cc : c->cond_const) cc.valid = false;  // the DepthTransformers' context-free images depend on the weights
  c->sec_begin = c->owned.size();
  RET_IF(build_hot_sections(c));
  c->sec_end = c->owned.size();
  HIP_CHECK_RET(hipDeviceSynchronize());
  // training mode keeps the hot-path tensors: parameters live in the arena, buffers (BatchNorm running statistics) as they were
  // uploaded -- engine_repack reads both
  for (auto it = c->raw.begin(); it != c->raw.end();) {
    if (c->train_mode && engine_hot_key(it->first)) {
      ++it;
      continue;
    }
    hipFree(it->second.d);
    it = c->raw.erase(it);
  }
  if (c->has_unet) RET_IF(engine_side_init(c));  // side stream + events now, so that nothing is created on the step path
  c->finalized = true;
  return 0;
}

// After an optimiser step changed the master parameters: every packed / folded / stacked fp16 weight of the UNet, the step
// embedding and the conditioner is re-derived IN PLACE (same allocations, same pointers).
// after: the stream whose enqueued work (the optimiser update) the re-pack must follow and that later work (the next forward)
// is enqueued on, or null.  With a stream the re-pack is asynchronous: the build streams wait for an event on it and it waits
// for theirs -- no device synchronisation, the host enqueues the ~600 launches while the AdamW kernels still run.  Without one
// (mvd_train_repack, arena adoption) the device is synchronised before and after, as before.
int engine_repack(mvd_ctx* c, hipStream_t after, bool have_stream) {
  if (!c->finalized || !c->train_mode) return mvd_fail("engine_repack: the context was not finalized in training mode");
  HIP_CHECK_RET(hipSetDevice(c->device));
  // ~600 pack / fold launches of 5-50 us that do not fill the chip: four streams (MVD_REPACK_STREAMS=1: one stream)
  static const bool one_stream = getenv("MVD_REPACK_STREAMS") != nullptr && getenv("MVD_REPACK_STREAMS")[0] == '1';
  if (!c->bstreams[0])
    for (int i = 0; i < 4; ++i) {
      HIP_CHECK_RET(hipStreamCreateWithFlags(&c->bstreams[i], hipStreamNonBlocking));
      HIP_CHECK_RET(hipEventCreateWithFlags(&c->bevents[i], hipEventDisableTiming));
    }
  if (!c->bev_after) HIP_CHECK_RET(hipEventCreateWithFlags(&c->bev_after, hipEventDisableTiming));
  if (have_stream) {
    HIP_CHECK_RET(hipEventRecord(c->bev_after, after));
    for (int i = 0; i < 4; ++i) HIP_CHECK_RET(hipStreamWaitEvent(c->bstreams[i], c->bev_after, 0));
  } else {
    HIP_CHECK_RET(hipDeviceSynchronize());
  }
  c->bs_multi = !one_stream;
  c->bs_rr = 0;
  c->bs = c->bstreams[0];  // (one stream: every launch on build stream 0)
  c->repacking = true;
  c->repack_cursor = c->sec_begin;
  const int r = build_hot_sections(c);
  const bool complete = c->repack_cursor == c->sec_end;
  c->repacking = false;
  c->bs_multi = false;
  c->bs = 0;
  if (have_stream) {  // also on the error paths: nothing later on `after` may overtake what was enqueued
    for (int i = 0; i < 4; ++i) {
      hipEventRecord(c->bevents[i], c->bstreams[i]);
      hipStreamWaitEvent(after, c->bevents[i], 0);
    }
  } else {
    hipDeviceSynchronize();
  }
  RET_IF(r);
  if (!complete) return mvd_fail("engine_repack: allocation sequence shorter than the first build");
  return 0;
}
