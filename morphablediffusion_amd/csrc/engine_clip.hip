// CLIP image embedding executor: FrozenCLIPImageEmbedder.forward of the reference
// (ldm/modules/encoders/modules.py:363-379) = preprocess + `clip.load(...)`'s VisionTransformer.forward, as
// SyncMultiviewDiffusion.prepare calls it once per sample (morphable_diffusion.py:487-488).  Runs on the UNet's
// kernels: the patch convolution (k14 s14, no bias) is a GEMM over an im2col written by the preprocess kernel; each
// ResidualAttentionBlock is LayerNorm -> q|k|v GEMM -> flash attention (V transposed while staged) -> out_proj GEMM with the
// residual in the epilogue -> LayerNorm -> c_fc GEMM -> c_proj GEMM with the residual in the epilogue.
//   * token axis: 257 tokens live in 264 rows per sample (row counts stay multiples of 8 for the GEMM tiles); the pad
//     rows start at zero, never enter a softmax (keys >= T are masked) and stay finite
//   * the v bias is folded into out_proj's bias (softmax rows sum to 1)
//   * QuickGELU(v) = v sigmoid(1.702 v) = silu(1.702 v) / 1.702: c_fc runs with alpha = 1.702 (bias pre-scaled) and
//     the SiLU epilogue, c_proj with alpha = 1 / 1.702 -- no new epilogue in the shared GEMM kernels
// Residual stream fp32, operand-only tensors fp16 (as in the UNet).
#include "engine.h"

int engine_clip_encode(mvd_ctx* c, const float* x_nchw, int B, int H, int W, float* out, hipStream_t s) {
  const ClipW& k = c->clip;
  if (!k.present) return mvd_fail("clip_encode: no clip_image_encoder.model.visual.* weights were uploaded");
  if (B < 1 || H < 2 || W < 2) return mvd_fail("clip_encode: bad image shape");
  WsScope ws_scope(c);
  const int C = k.width, G2 = (k.image / k.patch) * (k.image / k.patch), T = k.T, Tp = k.Tp;
  const size_t rows = (size_t)B * Tp, prow = (size_t)B * G2;
  half_t* patches = ws_alloc<half_t>(c, prow * k.Kp);
  float* pe = ws_alloc<float>(c, prow * C);
  float* x0 = ws_alloc<float>(c, rows * C);
  float* x = ws_alloc<float>(c, rows * C);
  float* x2 = ws_alloc<float>(c, rows * C);
  half_t* l1 = ws_alloc<half_t>(c, rows * C);
  half_t* qkv = ws_alloc<half_t>(c, rows * 3 * C);
  half_t* ao = ws_alloc<half_t>(c, rows * C);
  half_t* hh = ws_alloc<half_t>(c, rows * 4 * C);
  float* cl = ws_alloc<float>(c, (size_t)B * C);
  WS_CHECK(patches && pe && x0 && x && x2 && l1 && qkv && ao && hh && cl);

  RET_IF(launch_clip_patches(x_nchw, B, H, W, k.image, k.patch, k.Kp, patches, s));
  GemmArgs g;
  g.a = patches; g.lda = k.Kp; g.w = &k.conv1; g.out = pe; g.ldc = C; g.use_bias = false;
  RET_IF(run_linear(c, g, B, (int)prow, s));
  RET_IF(launch_clip_tokens(pe, k.cls, k.pos, B, T, Tp, C, x0, s));
  RET_IF(launch_layernorm_f32(x0, C, (int)rows, C, k.ln_pre.g, k.ln_pre.b, 1e-5f, x, s));
  // the attention kernel never writes the pad rows of its output: they must not hold stale non-finite bits
  HIP_CHECK_RET(hipMemsetAsync(ao, 0, rows * C * sizeof(half_t), s));

  for (const ClipLayerW& L : k.blk) {
    RET_IF(launch_layernorm(x, (int)rows, C, L.ln1.g, L.ln1.b, 1e-5f, l1, s));
    g = GemmArgs();
    g.a = l1; g.lda = C; g.w = &L.qkv; g.out = qkv; g.out_f32 = 0; g.ldc = 3 * C;
    RET_IF(run_linear(c, g, B, (int)rows, s));
    RET_IF(launch_attention(qkv, 3 * C, qkv + 2 * C, 3 * C, ao, C, B, T, k.heads, C / k.heads, s, Tp));
    g = GemmArgs();
    g.a = ao; g.lda = C; g.w = &L.out; g.out = x2; g.ldc = C; g.resid = x; g.ldr = C;
    RET_IF(run_linear(c, g, B, (int)rows, s));
    RET_IF(launch_layernorm(x2, (int)rows, C, L.ln2.g, L.ln2.b, 1e-5f, l1, s));
    g = GemmArgs();
    g.a = l1; g.lda = C; g.w = &L.fc; g.out = hh; g.out_f32 = 0; g.ldc = 4 * C; g.alpha = 1.702f; g.act = ACT_SILU;
    RET_IF(run_linear(c, g, B, (int)rows, s));
    g = GemmArgs();
    g.a = hh; g.lda = 4 * C; g.w = &L.proj; g.out = x; g.ldc = C; g.alpha = 1.0f / 1.702f; g.resid = x2; g.ldr = C;
    RET_IF(run_linear(c, g, B, (int)rows, s));
  }
  // ln_post on the class token of every sample, then @ proj
  RET_IF(launch_layernorm_f32(x, (long)Tp * C, B, C, k.ln_post.g, k.ln_post.b, 1e-5f, cl, s));
  RET_IF(launch_small_linear(cl, C, B, C, k.proj.w, nullptr, k.embed, ACT_NONE, out, k.embed, 0, s));
  return 0;
}
