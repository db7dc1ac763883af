// First-stage decoder executor: AutoencoderKL.decode of the reference
// (ldm/models/autoencoder.py:330-333; Decoder.forward ldm/modules/diffusionmodules/model.py:535-568) for a batch of
// latents, on the same kernels as the UNet: GroupNorm(32, eps 1e-6)+swish -> LDS-halo 3x3 conv with the residual in
// the epilogue, 1x1 shortcut convs, parity-folded upsample convs.  The single-head attention of the middle block has
// d = 512 (beyond the flash kernel's register tile), so it runs as GEMMs per sample: S = q k^T (alpha = C^-1/2) ->
// row softmax -> O = P V with V^T produced by the swapped GEMM; the v bias is folded into proj_out's bias (softmax
// rows sum to 1).  Layout: channels-last, residual stream fp32, operand-only tensors fp16 (as in the UNet).
#include "engine.h"

int bwd_cast_rows(const float* src, long ld, long rows, int C, int Cp, half_t* dst, hipStream_t s, int split);
int launch_softmax_rows_f32(const float* s, long rows, int cols, float* p, hipStream_t st);

namespace {

// ResnetBlock.forward (model.py:121-141, temb = None): x + conv2(swish(GN(conv1(swish(GN(x))))))
int vae_res(mvd_ctx* c, const VaeResW& r, const float* in, float* out, int B, int H, int W, hipStream_t s) {
  WsScope ws_scope(c);
  const size_t rows = (size_t)B * H * W;
  // exact mode (mvd_set_vae_precision): every operand is [hi | lo | hi] / [w_hi | w_hi | w_lo], see ConvW::xp
  const int w1 = r.c1.xp ? 3 : 1, w2 = r.c2.xp ? 3 : 1;
  half_t* a1 = ws_alloc<half_t>(c, rows * r.cin * w1);
  float* h1 = ws_alloc<float>(c, rows * r.cout);
  half_t* a2 = ws_alloc<half_t>(c, rows * r.cout * w2);
  WS_CHECK(a1 && h1 && a2);
  RET_IF(run_group_norm(c, in, r.cin, B, H * W, r.n1, 32, 1e-6f, ACT_SILU, nullptr, a1, r.cin * w1, s, 0, r.c1.xp));
  GemmArgs g;
  g.a = a1; g.lda = r.cin * w1; g.w = &r.c1; g.out = h1; g.ldc = r.cout;
  RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));
  RET_IF(run_group_norm(c, h1, r.cout, B, H * W, r.n2, 32, 1e-6f, ACT_SILU, nullptr, a2, r.cout * w2, s, 0, r.c2.xp));
  const float* resid = in;
  if (r.has_skip) {  // nin_shortcut: 1x1 conv on the fp32 input
    float* sk = ws_alloc<float>(c, rows * r.cout);
    WS_CHECK(sk);
    GemmArgs gs;
    gs.a = in; gs.a_f32 = 1; gs.lda = r.cin; gs.w = &r.skip; gs.out = sk; gs.ldc = r.cout;
    if (r.skip.xp) {
      half_t* as = ws_alloc<half_t>(c, rows * 3 * r.cin);
      WS_CHECK(as);
      RET_IF(launch_rows_f32_to_f16_split(in, r.cin, (long)rows, r.cin, as, s));
      gs.a = as; gs.a_f32 = 0; gs.lda = 3 * r.cin;
    }
    RET_IF(run_linear(c, gs, B, (int)rows, s));
    resid = sk;
  }
  g = GemmArgs();
  g.a = a2; g.lda = r.cout * w2; g.w = &r.c2; g.out = out; g.ldc = r.cout; g.resid = resid; g.ldr = r.cout;
  RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));
  return 0;
}

// AttnBlock.forward (model.py:178-202) with every product in extended precision: q, k, V^T and the probabilities are kept in
// fp32 and split into fp16 hi + lo operand images ([hi | lo | hi] on the row side, [hi | hi | lo] on the weight side)
int vae_attn_exact(mvd_ctx* c, const VaeAttnW& v, const float* in, float* out, int B, int HW, hipStream_t s) {
  WsScope ws_scope(c);
  const int C = v.norm.C;
  const size_t rows = (size_t)B * HW;
  half_t* hn = ws_alloc<half_t>(c, rows * 3 * C);
  float* qf = ws_alloc<float>(c, rows * C);
  float* kf = ws_alloc<float>(c, rows * C);
  half_t* q3 = ws_alloc<half_t>(c, rows * 3 * C);
  half_t* k3 = ws_alloc<half_t>(c, rows * 3 * C);
  float* vtf = ws_alloc<float>(c, rows * C);              // per sample [C][HW]
  half_t* vt3 = ws_alloc<half_t>(c, rows * 3 * C);        // per sample [C][3 HW]
  float* sc = ws_alloc<float>(c, rows * HW);
  half_t* p3 = ws_alloc<half_t>(c, rows * 3 * HW);
  float* aof = ws_alloc<float>(c, rows * C);
  half_t* ao3 = ws_alloc<half_t>(c, rows * 3 * C);
  WS_CHECK(hn && qf && kf && q3 && k3 && vtf && vt3 && sc && p3 && aof && ao3);
  RET_IF(run_group_norm(c, in, C, B, HW, v.norm, 32, 1e-6f, ACT_NONE, nullptr, hn, 3 * C, s, 0, 1));
  GemmArgs g;
  g.a = hn; g.lda = 3 * C; g.w = &v.q; g.out = qf; g.ldc = C;
  RET_IF(run_linear(c, g, B, (int)rows, s));
  g = GemmArgs();
  g.a = hn; g.lda = 3 * C; g.w = &v.k; g.out = kf; g.ldc = C;
  RET_IF(run_linear(c, g, B, (int)rows, s));
  RET_IF(bwd_cast_rows(qf, C, (long)rows, C, C, q3, s, 1));
  RET_IF(bwd_cast_rows(kf, C, (long)rows, C, C, k3, s, 2));
  const float scale = 1.0f / sqrtf((float)C);
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * HW;
    // V^T = W_v X^T: the (extended-precision) weight pack is the row operand, this sample's split tokens the weight operand:
    // w_hi x_hi + w_hi x_lo + w_lo x_hi, the same three products
    ConvW xw;
    xw.w = hn + o * 3 * C; xw.N = HW; xw.Cin = 3 * C; xw.taps = 1;
    g = GemmArgs();
    g.a = v.v.w; g.lda = 3 * C; g.w = &xw; g.out = vtf + o * C; g.ldc = HW; g.use_bias = false;
    RET_IF(run_linear(c, g, 1, C, s));
    ConvW kw;
    kw.w = k3 + o * 3 * C; kw.N = HW; kw.Cin = 3 * C; kw.taps = 1;
    g = GemmArgs();
    g.a = q3 + o * 3 * C; g.lda = 3 * C; g.w = &kw; g.out = sc + o * HW; g.ldc = HW; g.use_bias = false; g.alpha = scale;
    RET_IF(run_linear(c, g, 1, HW, s));
    RET_IF(bwd_cast_rows(vtf + o * C, HW, C, HW, HW, vt3 + o * 3 * C, s, 2));
  }
  RET_IF(launch_softmax_rows_f32(sc, (long)rows, HW, sc, s));
  RET_IF(bwd_cast_rows(sc, HW, (long)rows, HW, HW, p3, s, 1));
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * HW;
    ConvW vw;
    vw.w = vt3 + o * 3 * C; vw.N = C; vw.Cin = 3 * HW; vw.taps = 1;
    g = GemmArgs();
    g.a = p3 + o * 3 * HW; g.lda = 3 * HW; g.w = &vw; g.out = aof + o * C; g.ldc = C; g.use_bias = false;
    RET_IF(run_linear(c, g, 1, HW, s));
  }
  RET_IF(bwd_cast_rows(aof, C, (long)rows, C, C, ao3, s, 1));
  g = GemmArgs();
  g.a = ao3; g.lda = 3 * C; g.w = &v.proj; g.out = out; g.ldc = C; g.resid = in; g.ldr = C;
  return run_linear(c, g, B, (int)rows, s);
}

// AttnBlock.forward (model.py:178-202)
int vae_attn(mvd_ctx* c, const VaeAttnW& v, const float* in, float* out, int B, int HW, hipStream_t s) {
  if (v.q.xp) return vae_attn_exact(c, v, in, out, B, HW, s);
  WsScope ws_scope(c);
  const int C = v.norm.C;
  const size_t rows = (size_t)B * HW;
  half_t* hn = ws_alloc<half_t>(c, rows * C);
  half_t* q = ws_alloc<half_t>(c, rows * C);
  half_t* k = ws_alloc<half_t>(c, rows * C);
  half_t* vt = ws_alloc<half_t>(c, rows * C);            // per sample [C][HW]
  float* sc = ws_alloc<float>(c, rows * HW);              // per sample [HW][HW]
  half_t* pr = ws_alloc<half_t>(c, rows * HW);
  half_t* ao = ws_alloc<half_t>(c, rows * C);
  WS_CHECK(hn && q && k && vt && sc && pr && ao);
  RET_IF(run_group_norm(c, in, C, B, HW, v.norm, 32, 1e-6f, ACT_NONE, nullptr, hn, C, s));
  GemmArgs g;
  g.a = hn; g.lda = C; g.w = &v.q; g.out = q; g.out_f32 = 0; g.ldc = C;
  RET_IF(run_linear(c, g, B, (int)rows, s));
  g = GemmArgs();
  g.a = hn; g.lda = C; g.w = &v.k; g.out = k; g.out_f32 = 0; g.ldc = C;
  RET_IF(run_linear(c, g, B, (int)rows, s));
  const float scale = 1.0f / sqrtf((float)C);
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * HW;
    // V^T = W_v X^T: the weight matrix is the "activation" operand, this sample's tokens are the "weights"
    ConvW xw;
    xw.w = hn + o * C; xw.N = HW; xw.Cin = C; xw.taps = 1;
    g = GemmArgs();
    g.a = v.v.w; g.lda = C; g.w = &xw; g.out = vt + o * C; g.out_f32 = 0; g.ldc = HW; g.use_bias = false;
    RET_IF(run_linear(c, g, 1, C, s));
    // S = q k^T * C^-1/2
    ConvW kw;
    kw.w = k + o * C; kw.N = HW; kw.Cin = C; kw.taps = 1;
    g = GemmArgs();
    g.a = q + o * C; g.lda = C; g.w = &kw; g.out = sc + o * HW; g.ldc = HW; g.use_bias = false; g.alpha = scale;
    RET_IF(run_linear(c, g, 1, HW, s));
  }
  RET_IF(launch_softmax_rows(sc, (long)rows, HW, pr, s));
  for (int b = 0; b < B; ++b) {
    const size_t o = (size_t)b * HW;
    ConvW vw;
    vw.w = vt + o * C; vw.N = C; vw.Cin = HW; vw.taps = 1;
    g = GemmArgs();
    g.a = pr + o * HW; g.lda = HW; g.w = &vw; g.out = ao + o * C; g.out_f32 = 0; g.ldc = C; g.use_bias = false;
    RET_IF(run_linear(c, g, 1, HW, s));
  }
  g = GemmArgs();
  g.a = ao; g.lda = C; g.w = &v.proj; g.out = out; g.ldc = C; g.resid = in; g.ldr = C;
  RET_IF(run_linear(c, g, B, (int)rows, s));
  return 0;
}

// conv / linear on an fp32 source: in exact mode the source is first split into the [hi | lo | hi] operand
int vae_conv_f32(mvd_ctx* c, GemmArgs g, const ConvW& w, const float* src, int ld, int B, int H, int W, int stride, int ups, bool linear,
                 hipStream_t s) {
  WsScope ws_scope(c, WS_TEMP);
  const long rows = (long)B * H * W;
  g.w = &w;
  g.a = src; g.a_f32 = 1; g.lda = ld;
  if (w.xp) {
    half_t* as = ws_alloc<half_t>(c, (size_t)rows * 3 * w.cin_l);
    WS_CHECK(as);
    RET_IF(launch_rows_f32_to_f16_split(src, ld, rows, w.cin_l, as, s));
    g.a = as; g.a_f32 = 0; g.lda = 3 * w.cin_l;
  }
  if (linear) return run_linear(c, g, B, (int)rows, s);
  if (ups && w.w_up) return run_upconv2d(c, g, B, H, W, s);
  return run_conv2d(c, g, B, H, W, stride, ups, s);
}

}  // namespace

int engine_vae_decode(mvd_ctx* c, const float* z_nchw, int B, int h, int w, float* out_nchw, hipStream_t s) {
  const VaeW& v = c->vae;
  if (!v.present) return mvd_fail("first-stage decoder weights not uploaded / finalized");
  if ((h % 16) || (w % 16)) return mvd_fail("vae_decode: latent height and width must be multiples of 16");
  WsScope ws_scope(c);
  int H = h, W = w;
  size_t rows = (size_t)B * H * W;
  // post_quant_conv (1x1, embed -> z_channels), written into an 8-channel zero-padded tensor for conv_in
  float* zin = ws_alloc<float>(c, rows * 8);
  float* x0 = ws_alloc<float>(c, rows * 8);
  WS_CHECK(zin && x0);
  RET_IF(launch_nchw_to_nhwc(z_nchw, B, v.embed, H * W, zin, 8, 8, s));
  HIP_CHECK_RET(hipMemsetAsync(x0, 0, rows * 8 * sizeof(float), s));
  GemmArgs g;
  g.out = x0; g.ldc = 8;
  RET_IF(vae_conv_f32(c, g, v.post_quant, zin, 8, B, H, W, 1, 0, true, s));
  // ping-pong buffers sized for the largest level
  size_t maxel = 0;
  {
    int hh = H, ww = W, ch = v.block_in;
    maxel = (size_t)B * hh * ww * ch;
    for (int l = v.nlev - 1; l >= 0; --l) {
      for (auto& r : v.up[l]) {
        ch = r.cout;
        const size_t e = (size_t)B * hh * ww * (r.cin > r.cout ? r.cin : r.cout);
        if (e > maxel) maxel = e;
      }
      if (l > 0) {
        hh *= 2;
        ww *= 2;
        const size_t e = (size_t)B * hh * ww * ch;
        if (e > maxel) maxel = e;
      }
    }
  }
  float* bufA = ws_alloc<float>(c, maxel);
  float* bufB = ws_alloc<float>(c, maxel);
  WS_CHECK(bufA && bufB);
  float *cur = bufA, *nxt = bufB;
  auto swap = [&]() { float* t = cur; cur = nxt; nxt = t; };
  g = GemmArgs();
  g.out = cur; g.ldc = v.block_in;
  RET_IF(vae_conv_f32(c, g, v.conv_in, x0, 8, B, H, W, 1, 0, false, s));
  RET_IF(vae_res(c, v.mid1, cur, nxt, B, H, W, s)); swap();
  RET_IF(vae_attn(c, v.attn, cur, nxt, B, H * W, s)); swap();
  RET_IF(vae_res(c, v.mid2, cur, nxt, B, H, W, s)); swap();
  int ch = v.block_in;
  for (int l = v.nlev - 1; l >= 0; --l) {
    for (auto& r : v.up[l]) {
      RET_IF(vae_res(c, r, cur, nxt, B, H, W, s)); swap();
      ch = r.cout;
    }
    if (l > 0) {  // Upsample: nearest x2 + conv3x3 (model.py:53-57)
      g = GemmArgs();
      g.out = nxt; g.ldc = ch;
      // always the parity-folded form (results do not depend on the batch size); exact mode: the 9-tap form on the split operand
      RET_IF(vae_conv_f32(c, g, v.up_conv[l], cur, ch, B, H, W, 1, 1, false, s));
      swap();
      H *= 2;
      W *= 2;
    }
  }
  rows = (size_t)B * H * W;
  const int wo = v.conv_out.xp ? 3 : 1;
  half_t* a = ws_alloc<half_t>(c, rows * ch * wo);
  float* o4 = ws_alloc<float>(c, rows * 4);
  WS_CHECK(a && o4);
  RET_IF(run_group_norm(c, cur, ch, B, H * W, v.norm_out, 32, 1e-6f, ACT_SILU, nullptr, a, ch * wo, s, 0, v.conv_out.xp));
  g = GemmArgs();
  g.a = a; g.lda = ch * wo; g.w = &v.conv_out; g.out = o4; g.ldc = 4;
  RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));
  RET_IF(launch_nhwc_to_nchw(o4, 4, B, v.out_ch, H * W, out_nchw, s));
  return 0;
}

// Encoder.forward (model.py:434-459) + quant_conv (autoencoder.py:324-328); the posterior's sample()/mode() are host code
int engine_vae_encode(mvd_ctx* c, const float* x_nchw, int B, int H, int W, float* moments_nchw, hipStream_t s) {
  const VaeEncW& v = c->vae_enc;
  if (!v.present) return mvd_fail("first-stage encoder weights not uploaded / finalized");
  const int down = 1 << (v.nlev - 1);
  if ((H % (16 * down)) || (W % (16 * down))) return mvd_fail("vae_encode: image size must be a multiple of 16 x the downsampling factor");
  WsScope ws_scope(c);
  size_t rows = (size_t)B * H * W;
  float* x0 = ws_alloc<float>(c, rows * 8);
  size_t maxel = rows * (size_t)v.conv_in.N;
  for (auto& r : v.down[0]) maxel = rows * (size_t)(r.cout > r.cin ? r.cout : r.cin) > maxel ? rows * (size_t)(r.cout > r.cin ? r.cout : r.cin) : maxel;
  float* bufA = ws_alloc<float>(c, maxel);
  float* bufB = ws_alloc<float>(c, maxel);
  WS_CHECK(x0 && bufA && bufB);
  float *cur = bufA, *nxt = bufB;
  auto swap = [&]() { float* t = cur; cur = nxt; nxt = t; };
  RET_IF(launch_nchw_to_nhwc(x_nchw, B, v.in_ch, H * W, x0, 8, 8, s));
  GemmArgs g;
  g.out = cur; g.ldc = v.conv_in.N;
  RET_IF(vae_conv_f32(c, g, v.conv_in, x0, 8, B, H, W, 1, 0, false, s));
  int ch = v.conv_in.N;
  for (int l = 0; l < v.nlev; ++l) {
    for (auto& r : v.down[l]) {
      RET_IF(vae_res(c, r, cur, nxt, B, H, W, s)); swap();
      ch = r.cout;
    }
    if (l < v.nlev - 1) {  // Downsample: zero pad right/bottom, conv k3 s2 (model.py:72-76)
      g = GemmArgs();
      g.out = nxt; g.ldc = ch; g.tap_shift = 1;
      RET_IF(vae_conv_f32(c, g, v.down_conv[l], cur, ch, B, H, W, 2, 0, false, s));
      swap();
      H /= 2;
      W /= 2;
    }
  }
  RET_IF(vae_res(c, v.mid1, cur, nxt, B, H, W, s)); swap();
  RET_IF(vae_attn(c, v.attn, cur, nxt, B, H * W, s)); swap();
  RET_IF(vae_res(c, v.mid2, cur, nxt, B, H, W, s)); swap();
  rows = (size_t)B * H * W;
  const int wo = v.conv_out.xp ? 3 : 1;
  half_t* a = ws_alloc<half_t>(c, rows * ch * wo);
  float* h8 = ws_alloc<float>(c, rows * v.conv_out.N);
  float* mo = ws_alloc<float>(c, rows * v.mom);
  WS_CHECK(a && h8 && mo);
  RET_IF(run_group_norm(c, cur, ch, B, H * W, v.norm_out, 32, 1e-6f, ACT_SILU, nullptr, a, ch * wo, s, 0, v.conv_out.xp));
  g = GemmArgs();
  g.a = a; g.lda = ch * wo; g.w = &v.conv_out; g.out = h8; g.ldc = v.conv_out.N;
  RET_IF(run_conv2d(c, g, B, H, W, 1, 0, s));
  g = GemmArgs();
  g.out = mo; g.ldc = v.mom;
  RET_IF(vae_conv_f32(c, g, v.quant, h8, v.conv_out.N, B, H, W, 1, 0, true, s));
  RET_IF(launch_nhwc_to_nchw(mo, v.mom, B, v.mom, H * W, moments_nchw, s));
  return 0;
}
