// UNet executor: DepthWiseAttention.forward (reference ldm/models/diffusion/attention.py:117-138) as a fixed
// launch sequence.  Data layout in HBM:
//   * residual stream / anything read by a norm or a residual add: fp32, channels-last [B*H*W, C]
//   * anything that is only ever an MFMA operand (norm outputs, q/k/v, attention output, GEGLU output,
//     depth-attention z): fp16
//   * skip connections are "concatenated by construction": the producer of each skip tensor writes into the
//     channel slice of the buffer the matching output block will read (row stride = concatenated width).
#include <string.h>

#include <algorithm>

#include "engine.h"

namespace {

void igemm_init(IGemm& g) {
  memset(&g, 0, sizeof(g));
  g.alpha = 1.0f;
  g.sz = g.sy = g.sx = 1;
  g.out_linear = 1;
  g.Z = g.Y = g.X = g.B = 1;
  g.IZ = g.IY = g.IX = 1;
  g.PZ = g.PY = g.PX = 1;
  for (int i = 0; i < MVD_MAX_TAPS; ++i) g.tap[i] = igemm_tap(0, 0, 0, i);
}

void igemm_fill(IGemm& g, const GemmArgs& ga) {
  g.a = ga.a;
  g.a_f32 = ga.a_f32;
  g.lda = ga.lda;
  g.Cin = ga.w->Cin;
  g.cin_alg = ga.w->xp ? ga.w->cin_l : ga.w->Cin;
  g.w = ga.w->w;
  g.wx = ga.w->wx;
  g.wx_bn = ga.w->wx_bn;
  g.N = ga.w->N;
  g.out = ga.out;
  g.out_f32 = ga.out_f32;
  g.ldc = ga.ldc;
  g.bias = ga.use_bias ? ga.w->bias : nullptr;
  g.rowbias = ga.rowbias;
  g.rb_ld = ga.rb_ld;
  g.alpha = ga.alpha;
  g.rowscale = ga.rowscale;
  g.rs_ld = ga.rs_ld;
  g.gn_partial = ga.gn_partial;
  g.gn_cpg = ga.gn_cpg;
  g.resid = ga.resid;
  g.resid_f32 = ga.resid_f32;
  g.ldr = ga.ldr;
  g.geglu = ga.geglu;
  g.act = ga.act;
  g.out_split = ga.out_split;
}

int igemm_go(mvd_ctx* c, IGemm& g, int force_splitk, hipStream_t s, const GemmArgs* defer = nullptr) {
  const int M = g.B * g.Z * g.Y * g.X;
  g.bn = igemm_pick_bn(g.N, g.geglu);
  // (fp32 sources keep the 160-wide tile too: 32768 x 960 x 320 runs 48 us as 3 x 128 columns -- 768 workgroups, one and
  //  a half rounds, A read three times -- and 37 us as 2 x 160; MVD_IGEMM_F32_BN128=1 restores the old choice)
  static const bool f32_bn128 = getenv("MVD_IGEMM_F32_BN128") != nullptr;
  if (g.a_f32 && g.bn == 160 && f32_bn128) g.bn = 128;
  static const bool no_cx = getenv("MVD_NO_CONV3X") != nullptr;
  if (!no_cx && c->use_halo && g.wx && conv3x_eligible(g, g.wx_bn)) {
    // conv3x (k_conv3x.hip): one workgroup per CU and 16 x 16 pixel tile x wx_bn columns; split over 64-channel chunks when the
    // tiles do not fill the chip (microseconds, as for the halo kernel below)
    const int ncc = g.Cin / 64;
    const int tiles = (g.X == 8 ? cdiv(g.B, 4) : g.B * (g.Y / 16) * (g.X / 16)) * (g.N / g.wx_bn);  // 8 x 8 images: four per tile
    double best = 1e30;
    int sk = 1;
    for (int s2 = 1; s2 <= 8 && s2 <= ncc; ++s2) {
      const int rounds = cdiv(tiles * s2, 256), steps = cdiv(ncc, s2) * 9;
      double t = rounds * (steps * (g.wx_bn == 160 ? 0.75 : 0.62) + 8.0);
      if (s2 > 1) t += 3.0 + (s2 + 1) * (double)M * g.N * 4.0 / 3.5e6;
      if (t < best) {
        best = t;
        sk = s2;
      }
    }
    if (force_splitk > 0) sk = force_splitk < ncc ? force_splitk : ncc;
    if (sk > 1) sk = cdiv(ncc, cdiv(ncc, sk));
    WsScope ws_scope(c, WS_TEMP);
    g.splitk = sk;
    g.partial = nullptr;
    g.bn = g.wx_bn;
    bool deferred = false;
    if (sk > 1) {
      static const int defer_max = getenv("MVD_DEFER_MAX") ? atoi(getenv("MVD_DEFER_MAX")) : 16;
      if (defer && defer->slabs && defer->sk_used && sk <= defer_max && (size_t)sk * M * g.N <= defer->slabs_cap &&
          (!g.resid || (defer->defer_epilogue && g.resid_f32))) {
        g.partial = defer->slabs;
        deferred = true;
      } else {
        g.partial = ws_alloc<float>(c, (size_t)sk * M * g.N);
        if (!g.partial) g.splitk = 1;
      }
    }
    if (defer && defer->sk_used) *defer->sk_used = deferred ? sk : 1;
    const double kalg = (double)(g.cin_alg ? g.cin_alg : g.Cin);
    const double flops = 2.0 * M * g.N * kalg * 9.0;
    double bytes = (double)M * kalg * 2 + 9.0 * g.N * kalg * 2 + (g.splitk > 1 ? 0.0 : (double)M * g.N * 4);
    if (g.resid && g.splitk <= 1) bytes += (double)M * g.N * 4;
    int r;
    {
      ProbeScope ps(c, s, g.wx_bn == 160 ? "conv3x_kernel<5>" : "conv3x_kernel<4>", flops, bytes);
      r = launch_conv3x(g, g.wx, g.wx_bn, s);
    }
    if (!r && g.splitk > 1 && !deferred) {
      ProbeScope ps(c, s, "splitk_reduce_kernel", 0.0, (double)M * g.N * 4.0 * (g.splitk + 1));
      r = launch_splitk_reduce(g, s);
    }
    return r;
  }
  const bool halo = c->use_halo && conv3_halo_eligible(g);
  static const bool use_dense = getenv("MVD_NO_GEMM_DMA") == nullptr;
  static const int dense_min_m = getenv("MVD_DENSE_MIN_M") ? atoi(getenv("MVD_DENSE_MIN_M")) : 64;
  // one 256-row workgroup per CU: wins for the Linear layers and wherever weights stream (small M, long K), down to one
  // quarter-filled row tile (2-views-per-rank step: 7.08 ms with a 512-row threshold, 6.97 ms with 64);
  // the big shallow 3-D convs keep the 128-row gather kernel (finer tiles, 2 workgroups per CU)
  if (g.npar > 0) {  // parity-batched launch (run_convT3d / run_upconv2d): LDS-DMA kernel, no split-K
    if (!gemm_dma_eligible(g)) return mvd_fail("igemm_go: parity batch needs the LDS-DMA kernel");
    int kmax = 0;
    for (int p = 0; p < g.npar; ++p) kmax = g.par_ntaps[p] > kmax ? g.par_ntaps[p] : kmax;
    int nch = 1, sk2 = 1;
    gemm_dma_plan(M * g.npar, g.N, kmax * cdiv(g.Cin, 64), g.bn, 1, &nch, &sk2);
    g.nch = nch;
    g.splitk = 1;
    g.partial = nullptr;
    // Parity walk (k_gemm.hip PWALK): every workgroup walks the parity classes of its tile instead of one workgroup per class --
    // meant for the frustum network's level-0 ConvTranspose3d (384 tiles x 8 classes of 2 ... 16 k-steps each: 282 us at 154
    // TFLOP/s).  MEASURED (profiles/r06_y_ab_pwalk.txt, threshold 192 tiles): 12.828 vs 12.811 ms per step, 6.169 vs 6.169 at 2
    // views per rank -- nothing: the launch sits on the side stream beside the trunk.  Tested form, off unless
    // MVD_PAR_WALK_MIN=<tiles> is set.
    static const int pwalk_min = getenv("MVD_PAR_WALK_MIN") ? atoi(getenv("MVD_PAR_WALK_MIN")) : 0;
    if (pwalk_min > 0 && cdiv(M, 256) * cdiv(g.N, g.bn) >= pwalk_min) {
      g.par_walk = 1;
      g.nch = 1;
    }
    double fl = 0.0;
    for (int p = 0; p < g.npar; ++p) fl += 2.0 * M * g.N * (double)(g.cin_alg ? g.cin_alg : g.Cin) * g.par_ntaps[p];
    const double by = (double)g.B * g.PZ * g.PY * g.PX * g.Cin * 2 + (double)g.npar * kmax * g.N * g.Cin * 2 +
                      (double)M * g.npar * g.N * (g.out_f32 ? 4 : 2) + (g.resid ? (double)M * g.npar * g.N * (g.resid_f32 ? 4 : 2) : 0.0);
    char fam[64];
    snprintf(fam, sizeof fam, "gemm_dma_kernel<%d,0>", g.bn);
    ProbeScope ps(c, s, fam, fl, by);
    return launch_gemm_dma(g, s);
  }
  const bool dense = use_dense && !halo && M >= dense_min_m && (g.ntaps == 1 || M <= 16384) && gemm_dma_eligible(g);
  if (g.gn_partial && !dense) return mvd_fail("igemm_go: the statistics-only pass needs the LDS-DMA kernel");
  int sk;
  if (halo) {
    // LDS-halo 3x3 kernel, one workgroup per CU: pick the column width and the split over 64-channel chunks that
    // minimise  rounds * (steps * step_cost + fill) + reduce pass  (microseconds, fitted to conv_bench3 sweeps)
    const int ncc = g.Cin / 64;
    double best = 1e30;
    int best_bn = 128, best_sk = 1;
    for (int bn = 128; bn <= 160; bn += 32) {
      const int tiles = conv3_halo_tiles(g, bn);
      for (int s2 = 1; s2 <= 8 && s2 <= ncc; ++s2) {
        const int rounds = cdiv(tiles * s2, 256), steps = cdiv(ncc, s2) * 9;
        double t = rounds * (steps * (bn == 160 ? 1.3 : 1.0) + 8.0);
        if (s2 > 1) t += 3.0 + (s2 + 1) * (double)M * g.N * 4.0 / 3.5e6;
        if (t < best) {
          best = t;
          best_bn = bn;
          best_sk = s2;
        }
      }
    }
    g.bn = best_bn;
    sk = force_splitk > 0 ? (force_splitk < ncc ? force_splitk : ncc) : best_sk;
    static const int tune_bn = getenv("MVD_HALO_BN") ? atoi(getenv("MVD_HALO_BN")) : 0;  // tools/conv_bench3.py sweeps
    static const int tune_sk = getenv("MVD_HALO_SK") ? atoi(getenv("MVD_HALO_SK")) : 0;
    if (tune_bn) g.bn = tune_bn;
    if (tune_sk) sk = tune_sk;
    if (sk > ncc) sk = ncc;
    if (sk > 1) sk = cdiv(ncc, cdiv(ncc, sk));  // no empty split (the kernels cut the chunk range in ceil(ncc/sk) pieces)
  } else if (dense) {
    int nch = 1, sk2 = 1;
    const int ksteps = g.ntaps * cdiv(g.Cin, 64);
    static const bool old_plan = getenv("MVD_OLD_PLAN") != nullptr;
    const bool plain_gemm = !g.gn_partial && !g.rowscale && g.ntaps == 1;
    if (plain_gemm && !old_plan) {
      int bn = 0, bm = 256;
      // (128-row tiles exist for the PLAIN instantiation only: one centre tap, linear rows)
      gemm_dma_plan_us(M, g.N, ksteps, g.geglu, g.out_f32 ? 4 : (g.out_split ? 6 : 2), g.resid ? (g.resid_f32 ? 4 : 2) : 0, &bn,
                       &nch, &sk2, gemm_dma_is_plain(g) ? &bm : nullptr);
      g.bn = bn;
      g.bm = bm;
    } else {
      gemm_dma_plan(M, g.N, ksteps, g.bn, g.geglu, &nch, &sk2);
    }
    if (g.gn_partial || g.rowscale) {  // folded-GroupNorm passes walk the whole tile grid: one round of workgroups
      if (g.bn == 160) g.bn = 128;
      const int tiles = cdiv(M, 256) * cdiv(g.N, g.bn);
      nch = cdiv(tiles, 256);
      if (nch > 64) nch = 64;
      force_splitk = 1;
    }
    sk = g.geglu ? 1 : (force_splitk > 0 ? force_splitk : sk2);
    {  // sweeps (tools/gemm_plan_sweep.py): column-tile width and split of the next dense launches
      static const int tune_bn = getenv("MVD_DENSE_BN") ? atoi(getenv("MVD_DENSE_BN")) : 0;
      static const int tune_sk = getenv("MVD_DENSE_SK") ? atoi(getenv("MVD_DENSE_SK")) : 0;
      static const int tune_bm = getenv("MVD_DENSE_BM") ? atoi(getenv("MVD_DENSE_BM")) : 0;
      if (tune_bn && !g.geglu && !g.gn_partial && !g.rowscale) {
        g.bn = tune_bn;
        g.bm = (tune_bm == 128 && gemm_dma_is_plain(g)) ? 128 : 256;
        gemm_dma_plan(M, g.N, ksteps, g.bn, g.geglu, &nch, &sk2);
        if (g.bm == 128) {  // no column walk with the swept 128-row tiles: one tile per workgroup
          nch = 1;
          sk2 = 1;
        }
        if (force_splitk <= 0) sk = sk2;
      }
      if (tune_sk && !g.geglu && force_splitk <= 0) sk = tune_sk;
    }
    if (sk > ksteps) sk = ksteps;
    if (sk > 1) sk = cdiv(ksteps, cdiv(ksteps, sk));  // no empty split
    g.nch = sk > 1 ? 1 : nch;
    static const bool plan_debug = getenv("MVD_PLAN_DEBUG") != nullptr;
    if (plan_debug) fprintf(stderr, "[plan] M=%d N=%d ksteps=%d -> bm=%d bn=%d nch=%d sk=%d\n", M, g.N, ksteps, g.bm ? g.bm : 256, g.bn, g.nch, sk);
  } else {
    const int ksteps = g.ntaps * cdiv(g.Cin, 64);
    sk = force_splitk > 0 ? force_splitk : igemm_pick_splitk(M, g.N, ksteps, g.bn);
    if (sk > ksteps) sk = ksteps;
    if (sk > 1) sk = cdiv(ksteps, cdiv(ksteps, sk));  // no empty split
  }
  WsScope ws_scope(c, WS_TEMP);
  g.splitk = sk;
  g.partial = nullptr;
  bool deferred = false;  // the caller's consumer adds the slabs (GemmArgs::slabs): no reduce pass
  if (sk > 1) {
    static const int defer_max = getenv("MVD_DEFER_MAX") ? atoi(getenv("MVD_DEFER_MAX")) : 16;  // A/B: 4 = the round-3 limit
    if (defer && defer->slabs && defer->sk_used && sk <= defer_max && (size_t)sk * M * g.N <= defer->slabs_cap && !g.geglu &&
        (!g.resid || (defer->defer_epilogue && g.resid_f32)) && g.act == 0 && g.alpha == 1.0f && g.out_linear && g.out_f32 &&
        !g.out_split) {
      g.partial = defer->slabs;
      deferred = true;
    } else {
      g.partial = ws_alloc<float>(c, (size_t)sk * M * g.N);
      if (!g.partial) g.splitk = 1;  // not enough scratch: single pass
    }
  }
  if (defer && defer->sk_used) *defer->sk_used = deferred ? sk : 1;
  static const bool timing = getenv("MVD_LAYER_TIMING") != nullptr;  // debugging aid: per-GEMM time on stderr
  static hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (timing) {
    if (!ev0) {
      hipEventCreate(&ev0);
      hipEventCreate(&ev1);
    }
    hipEventRecord(ev0, s);
  }
  int r;
  // algorithmic work of this launch: every operand and the result once (gathered taps re-read the same input pixels)
  // algorithmic FLOPs: the layer's own K (an extended-precision layer executes 3x that), every tap once
  const double flops = 2.0 * M * g.N * (double)(g.cin_alg ? g.cin_alg : g.Cin) * g.ntaps;
  const double in_rows = (double)g.B * g.PZ * g.PY * g.PX;
  const double out_cols = g.geglu ? g.N / 2 : g.N;
  const double kalg = (double)(g.cin_alg ? g.cin_alg : g.Cin);
  double bytes = in_rows * kalg * (g.a_f32 ? 4 : 2) + (double)g.ntaps * g.N * kalg * 2 +
                 (g.gn_partial ? 0.0 : (double)M * out_cols * (g.out_f32 ? 4 : 2));
  if (g.resid) bytes += (double)M * g.N * (g.resid_f32 ? 4 : 2);
  char fam[64];
  if (halo) {
    snprintf(fam, sizeof fam, "conv3_dma_kernel<%d,%d,%d>", g.bn == 160 ? 160 : 128, g.X % 16 == 0 ? 16 : 8, g.X % 16 == 0 ? 16 : 8);
    {
      ProbeScope ps(c, s, fam, flops, g.splitk > 1 ? bytes - (double)M * out_cols * (g.out_f32 ? 4 : 2) : bytes);
      r = launch_conv3_halo(g, s);
    }
  } else if (dense) {
    if (g.bm == 128) snprintf(fam, sizeof fam, "gemm_dma_kernel<128x%d>", g.bn);  // four-wave row tiles (round 6)
    else snprintf(fam, sizeof fam, "gemm_dma_kernel<%d,%d>", g.bn, g.gn_partial ? 1 : (g.rowscale ? 2 : 0));
    ProbeScope ps(c, s, fam, flops, bytes);
    r = launch_gemm_dma(g, s);
  } else {
    const int bn = g.bn ? g.bn : igemm_pick_bn(g.N, g.geglu);
    snprintf(fam, sizeof fam, "igemm_kernel<%d,%d>", g.a_f32, bn);
    ProbeScope ps(c, s, fam, flops, bytes);
    r = launch_igemm(g, s);
  }
  if (!r && g.splitk > 1 && !deferred) {  // the slabs are written once and read once: 2 x splitk x M x N x 4 bytes that no roofline needs
    ProbeScope ps(c, s, "splitk_reduce_kernel", 0.0, (double)M * g.N * 4.0 * (g.splitk + 1));
    r = launch_splitk_reduce(g, s);
  }
  if (timing) {
    hipEventRecord(ev1, s);
    hipEventSynchronize(ev1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    const double fl = 2.0 * M * g.N * g.Cin * g.ntaps;
    fprintf(stderr, "[gemm] M=%d N=%d Cin=%d taps=%d f32=%d halo=%d bn=%d sk=%d geglu=%d  %.1f us  %.0f TF\n", M, g.N, g.Cin,
            g.ntaps, g.a_f32, halo ? 1 : (dense ? 2 : 0), g.bn, g.splitk, g.geglu, ms * 1e3, fl / (ms * 1e-3) * 1e-12);
  }
  return r;
}

}  // namespace

int run_linear(mvd_ctx* c, const GemmArgs& ga, int B, int rows, hipStream_t s) {
  IGemm g;
  igemm_init(g);
  igemm_fill(g, ga);
  g.B = B;
  g.X = rows / B;
  g.IX = g.X;
  g.PX = g.X;
  g.ntaps = 1;
  return igemm_go(c, g, ga.force_splitk, s, &ga);
}

int run_conv2d(mvd_ctx* c, const GemmArgs& ga, int B, int H, int W, int stride, int ups, hipStream_t s) {
  IGemm g;
  igemm_init(g);
  igemm_fill(g, ga);
  g.B = B;
  g.PY = H;
  g.PX = W;
  g.ups = ups;
  g.IY = H << ups;
  g.IX = W << ups;
  g.sy = g.sx = stride;
  g.Y = (g.IY - 1) / stride + 1;
  g.X = (g.IX - 1) / stride + 1;
  if (ga.w->taps == 9) {
    g.ntaps = 9;
    for (int t = 0; t < 9; ++t) g.tap[t] = igemm_tap(0, t / 3 - 1 + ga.tap_shift, t % 3 - 1 + ga.tap_shift, t);
  } else if (ga.w->taps == 1) {
    g.ntaps = 1;
  } else {
    return mvd_fail("run_conv2d: kernel must be 1x1 or 3x3");
  }
  return igemm_go(c, g, ga.force_splitk, s, &ga);
}

int run_upconv2d(mvd_ctx* c, const GemmArgs& ga_in, int B, int H, int W, hipStream_t s) {
  if (!ga_in.w->w_up) return mvd_fail("run_upconv2d: weights were not folded");
  WsScope ws_scope(c, WS_TEMP);
  GemmArgs ga = ga_in;
  ConvW cw = *ga_in.w;
  cw.w = cw.w_up;
  cw.taps = 16;
  ga.w = &cw;
  if (ga.a_f32) {  // operand copy: the LDS-DMA kernel streams fp16
    half_t* ah = ws_alloc<half_t>(c, (size_t)B * H * W * cw.Cin);
    WS_CHECK(ah);
    RET_IF(launch_rows_f32_to_f16((const float*)ga.a, ga.lda, (long)B * H * W, cw.Cin, ah, s));
    ga.a = ah;
    ga.a_f32 = 0;
    ga.lda = cw.Cin;
  }
  static const bool no_batch = getenv("MVD_NO_PARITY_BATCH") != nullptr;
  IGemm gb;
  igemm_init(gb);
  igemm_fill(gb, ga);
  gb.B = B;
  gb.Y = gb.PY = gb.IY = H;
  gb.X = gb.PX = gb.IX = W;
  gb.out_linear = 0;
  gb.OZ = 1;
  gb.OY = 2 * H;
  gb.OX = 2 * W;
  gb.ozm = 1;
  gb.oym = gb.oxm = 2;
  const bool batched = !no_batch && ga.force_splitk <= 1 && gemm_dma_eligible(gb);
  for (int par = 0; par < 4; ++par) {
    const int py = par >> 1, px = par & 1;
    int taps[4];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        taps[a * 2 + b] = igemm_tap(0, py == 0 ? a - 1 : a, px == 0 ? b - 1 : b, par * 4 + a * 2 + b);
    if (batched) {
      gb.par_ntaps[par] = 4;
      for (int i = 0; i < 4; ++i) gb.par_tap[par][i] = taps[i];
      gb.par_oz[par] = 0;
      gb.par_oy[par] = py;
      gb.par_ox[par] = px;
      continue;
    }
    IGemm g = gb;
    g.ntaps = 4;
    for (int i = 0; i < 4; ++i) g.tap[i] = taps[i];
    g.oyo = py;
    g.oxo = px;
    RET_IF(igemm_go(c, g, ga.force_splitk, s));
  }
  if (batched) {
    gb.npar = 4;
    gb.ntaps = 4;
    gb.bn = igemm_pick_bn(gb.N, 0);
    RET_IF(igemm_go(c, gb, 0, s));
  }
  return 0;
}

int run_conv3d(mvd_ctx* c, const GemmArgs& ga, int B, int D, int H, int W, int stride, hipStream_t s) {
  IGemm g;
  igemm_init(g);
  igemm_fill(g, ga);
  g.B = B;
  g.PZ = g.IZ = D;
  g.PY = g.IY = H;
  g.PX = g.IX = W;
  g.sz = g.sy = g.sx = stride;
  g.Z = (D - 1) / stride + 1;
  g.Y = (H - 1) / stride + 1;
  g.X = (W - 1) / stride + 1;
  if (ga.w->taps == 27) {
    g.ntaps = 27;
    for (int t = 0; t < 27; ++t) g.tap[t] = igemm_tap(t / 9 - 1, (t / 3) % 3 - 1, t % 3 - 1, t);
  } else if (ga.w->taps == 1) {
    g.ntaps = 1;
  } else {
    return mvd_fail("run_conv3d: kernel must be 1 or 27 taps");
  }
  return igemm_go(c, g, ga.force_splitk, s);
}

// out[o] += in[i] * W[k] with o = 2 i - 1 + k  (ConvTranspose3d k3 s2 p1 op1).  For output parity p:
// p = 0 -> k = 1, i = q ;  p = 1 -> (k = 0, i = q + 1), (k = 2, i = q)   where o = 2 q + p.
int run_convT3d(mvd_ctx* c, const GemmArgs& ga, int B, int D, int H, int W, hipStream_t s) {
  if (ga.w->taps != 27) return mvd_fail("run_convT3d: expects a 3x3x3 kernel");
  static const bool no_batch = getenv("MVD_NO_PARITY_BATCH") != nullptr;
  IGemm gb;  // all 8 parity classes in one launch when the LDS-DMA kernel applies
  igemm_init(gb);
  igemm_fill(gb, ga);
  gb.B = B;
  gb.Z = gb.PZ = gb.IZ = D;
  gb.Y = gb.PY = gb.IY = H;
  gb.X = gb.PX = gb.IX = W;
  gb.out_linear = 0;
  gb.OZ = 2 * D;
  gb.OY = 2 * H;
  gb.OX = 2 * W;
  gb.ozm = gb.oym = gb.oxm = 2;
  const bool batched = !no_batch && !ga.a_f32 && ga.force_splitk <= 1 && B * D * H * W >= 512 && gemm_dma_eligible(gb);
  for (int par = 0; par < 8; ++par) {
    const int pz = par >> 2, py = (par >> 1) & 1, px = par & 1;
    const int kz[2] = {pz ? 0 : 1, 2}, dzv[2] = {pz ? 1 : 0, 0}, nz = pz ? 2 : 1;
    const int ky[2] = {py ? 0 : 1, 2}, dyv[2] = {py ? 1 : 0, 0}, ny = py ? 2 : 1;
    const int kx[2] = {px ? 0 : 1, 2}, dxv[2] = {px ? 1 : 0, 0}, nx = px ? 2 : 1;
    int taps[8], t = 0;
    for (int a = 0; a < nz; ++a)
      for (int b = 0; b < ny; ++b)
        for (int e = 0; e < nx; ++e) taps[t++] = igemm_tap(dzv[a], dyv[b], dxv[e], (kz[a] * 3 + ky[b]) * 3 + kx[e]);
    if (batched) {
      gb.par_ntaps[par] = t;
      for (int i = 0; i < t; ++i) gb.par_tap[par][i] = taps[i];
      gb.par_oz[par] = pz;
      gb.par_oy[par] = py;
      gb.par_ox[par] = px;
      continue;
    }
    IGemm g = gb;
    for (int i = 0; i < t; ++i) g.tap[i] = taps[i];
    g.ntaps = t;
    g.ozo = pz;
    g.oyo = py;
    g.oxo = px;
    RET_IF(igemm_go(c, g, ga.force_splitk, s));
  }
  if (batched) {
    gb.npar = 8;
    gb.ntaps = 8;
    gb.bn = igemm_pick_bn(gb.N, 0);
    RET_IF(igemm_go(c, gb, 0, s));
  }
  return 0;
}

int run_group_norm(mvd_ctx* c, const float* x, int ld, int B, int rows_per_sample, const NormW& n, int groups, float eps,
                   int act, const float* preadd, half_t* out, int ldo, hipStream_t s, int preadd_ld, int split, int nslab,
                   size_t slab_stride, const float* bias2, const float* resid, int ldr, float* mat, int ldm) {
  // algorithmic: x read once (fp32), fp16 result written once
  ProbeScope ps(c, s, "group_norm", 0.0, (double)B * rows_per_sample * n.C * 6.0);
  static const bool two_pass = getenv("MVD_GN_TWO_PASS") != nullptr;
  if (!two_pass && gn_group_eligible(ld, rows_per_sample, n.C, groups, preadd ? (preadd_ld ? preadd_ld : n.C) : 0, ldo))
    return launch_gn_group(x, ld, B, rows_per_sample, n.C, groups, preadd, preadd_ld ? preadd_ld : n.C, n.g, n.b, eps, act, out,
                           ldo, s, split, nslab, slab_stride, bias2, resid, ldr, mat, ldm);
  if (nslab > 1 || resid || mat) return mvd_fail("run_group_norm: slab input needs the single-pass form");
  WsScope ws_scope(c, WS_TEMP);
  float* partial = ws_alloc<float>(c, (size_t)B * gn_max_slabs() * groups * 2);
  WS_CHECK(partial);
  int nslabs = 0;
  RET_IF(launch_gn_stats(x, ld, B, rows_per_sample, n.C, groups, preadd, preadd_ld, partial, &nslabs, s));
  RET_IF(launch_gn_apply(x, ld, B, rows_per_sample, n.C, groups, preadd, preadd_ld, partial, nslabs, n.g, n.b, eps, act, out,
                         ldo, s, split));
  return 0;
}

namespace {

}  // namespace

// ResBlock._forward, openaimodel.py:256-276
int unet_do_res(Fwd& f, const ResW& r, View in, View out, int H, int W, ResSaved* sv, Carry* in_carry, Carry* out_carry) {
  mvd_ctx* c = f.c;
  WsScope ws_scope(c, WS_BLOCK);
  const int rows = f.Bv * H * W;
  // extended-precision layers (ConvW::xp) read [hi | lo | hi] operands: three times the logical width
  const int w1 = r.c1.xp ? 3 : 1, w2 = r.c2.xp ? 3 : 1;
  half_t* a1 = ws_alloc<half_t>(c, (size_t)rows * r.cin * w1);
  float* h1 = ws_alloc<float>(c, (size_t)rows * r.cout);
  half_t* a2 = ws_alloc<half_t>(c, (size_t)rows * r.cout * w2);
  WS_CHECK(a1 && h1 && a2);
  // The 1x1 skip convolution (openaimodel.py:273: skip_connection(x)) reads the block input only: on a helper stream it can run
  // beside GroupNorm1 -> conv1 -> GroupNorm2 instead of in front of conv2.  MEASURED (profiles/r06_z_ab_skip_side.txt): 12.24 vs
  // 12.245 ms per step, 6.11 vs 6.01 ms at 2 views per rank -- the chip gives two dependent-free kernels of one process no more
  // than their sum (the same answer as the two half-batch chains).  Tested form, on with MVD_SKIP_SIDE=1 only.
  // Its result buffer is allocated here (block scope, or the carry's storage); it takes no split-K scratch on the helper stream
  // (the workspace scopes are released in host order).
  static const bool no_skip_side = getenv("MVD_SKIP_SIDE") == nullptr;
  const bool skip_side = r.has_skip && !no_skip_side && !f.train && !sv;
  const bool sk_in_carry = r.has_skip && out_carry && out_carry->aux && out_carry->aux_cap >= (size_t)rows * r.cout;
  float* skbuf = nullptr;
  auto launch_skip = [&](hipStream_t ss, bool no_split) -> int {
    GemmArgs gs;
    gs.a = in.p; gs.a_f32 = 1; gs.lda = in.ld; gs.w = &r.skip; gs.out = skbuf; gs.ldc = r.cout;
    if (no_split) gs.force_splitk = 1;
    if (r.skip.xp) {  // fp32 source -> [hi | lo | hi] copy
      half_t* as = ws_alloc<half_t>(c, (size_t)rows * 3 * r.cin);
      WS_CHECK(as);
      RET_IF(launch_rows_f32_to_f16_split(in.p, in.ld, rows, r.cin, as, ss));
      gs.a = as; gs.a_f32 = 0; gs.lda = 3 * r.cin;
    }
    return run_linear(c, gs, f.Bv, rows, ss);
  };
  auto fork_skip = [&]() -> int {
    if (!c->side2) {
      HIP_CHECK_RET(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
      HIP_CHECK_RET(hipEventCreateWithFlags(&c->ev_s2_fork, hipEventDisableTiming));
      HIP_CHECK_RET(hipEventCreateWithFlags(&c->ev_s2_join, hipEventDisableTiming));
    }
    HIP_CHECK_RET(hipEventRecord(c->ev_s2_fork, f.s));  // the block input is final on the caller's stream here
    HIP_CHECK_RET(hipStreamWaitEvent(c->side2, c->ev_s2_fork, 0));
    RET_IF(launch_skip(c->side2, true));
    HIP_CHECK_RET(hipEventRecord(c->ev_s2_join, c->side2));
    return 0;
  };
  if (r.has_skip) {
    skbuf = sk_in_carry ? out_carry->aux : ws_alloc<float>(c, (size_t)rows * r.cout);
    WS_CHECK(skbuf);
  }
  const bool in_slabs = in_carry && in_carry->sk > 1;
  if (skip_side && !in_slabs) RET_IF(fork_skip());
  if (in_slabs)  // the previous block's last GEMM left its split-K slabs: sum + bias + residual -> in, then normalise
    RET_IF(run_group_norm(c, in_carry->slabs, r.cin, f.Bv, H * W, r.n1, 32, 1e-5f, ACT_SILU, nullptr, a1, r.cin * w1, f.s, 0, r.c1.xp,
                          in_carry->sk, in_carry->stride, in_carry->bias, in_carry->resid, in_carry->ldr, in.p, in.ld));
  else
    RET_IF(run_group_norm(c, in.p, in.ld, f.Bv, H * W, r.n1, 32, 1e-5f, ACT_SILU, nullptr, a1, r.cin * w1, f.s, 0, r.c1.xp));
  if (skip_side && in_slabs) RET_IF(fork_skip());  // (that GroupNorm materialised the block input: the skip conv reads it)
  GemmArgs g1;
  g1.a = a1; g1.lda = r.cin * w1; g1.w = &r.c1; g1.out = h1; g1.ldc = r.cout;
  g1.rowbias = f.emb_all + r.emb_off; g1.rb_ld = c->emb_total;
  // conv1's result only feeds GroupNorm 2: when the conv splits K, the norm adds the slabs (+ bias + emb) itself and the
  // reduce pass (a launch, a write and a read of h1) disappears
  static const bool no_defer = getenv("MVD_NO_DEFER_REDUCE") != nullptr || getenv("MVD_GN_TWO_PASS") != nullptr;
  int sk1 = 1;
  const size_t slab_elems = (size_t)rows * r.cout;
  if (!no_defer && !f.train && gn_group_eligible(r.cout, H * W, r.cout, 32, c->emb_total, r.cout * w2)) {
    const size_t nsl = rows > 8192 ? 4 : 16;  // the 4 x 4 level's convolutions split K twelve ways; full resolution never splits
    g1.slabs = ws_alloc<float>(c, nsl * slab_elems);
    g1.slabs_cap = g1.slabs ? nsl * slab_elems : 0;
    g1.sk_used = &sk1;
  }
  RET_IF(run_conv2d(c, g1, f.Bv, H, W, 1, 0, f.s));
  if (sk1 > 1)
    RET_IF(run_group_norm(c, g1.slabs, r.cout, f.Bv, H * W, r.n2, 32, 1e-5f, ACT_SILU, g1.rowbias, a2, r.cout * w2, f.s, g1.rb_ld,
                          r.c2.xp, sk1, slab_elems, r.c1.bias));
  else
    RET_IF(run_group_norm(c, h1, r.cout, f.Bv, H * W, r.n2, 32, 1e-5f, ACT_SILU, nullptr, a2, r.cout * w2, f.s, 0, r.c2.xp));
  const float* resid = in.p;
  int ldr = in.ld;
  bool skip_outlives_scope = true;  // no skip conv: the residual is the block input, owned by the caller
  if (r.has_skip) {
    // (a deferred conv2 hands the residual to the NEXT block's GroupNorm: the skip conv's result then lives in the carry)
    // A deferred conv2 hands `resid` to the NEXT block's GroupNorm, i.e. beyond this function's workspace scope: the skip conv's
    // result may then only live in the carry's own storage.  Without that storage (carry_storage could not get the second
    // buffer) conv2 must not defer.
    skip_outlives_scope = sk_in_carry;
    if (skip_side) HIP_CHECK_RET(hipStreamWaitEvent(f.s, c->ev_s2_join, 0));  // conv2 (or the next block's GroupNorm) reads it
    else RET_IF(launch_skip(f.s, false));
    resid = skbuf;
    ldr = r.cout;
  }
  GemmArgs g2;
  g2.a = a2; g2.lda = r.cout * w2; g2.w = &r.c2; g2.out = out.p; g2.ldc = out.ld; g2.resid = resid; g2.ldr = ldr;
  int sk2 = 1;
  if (out_carry && out_carry->slabs && skip_outlives_scope) {
    g2.slabs = out_carry->slabs; g2.slabs_cap = out_carry->cap; g2.sk_used = &sk2; g2.defer_epilogue = true;
  }
  RET_IF(run_conv2d(c, g2, f.Bv, H, W, 1, 0, f.s));
  if (out_carry) {
    out_carry->sk = sk2;
    out_carry->stride = (size_t)rows * r.cout;
    out_carry->bias = r.c2.bias;
    out_carry->resid = resid;
    out_carry->ldr = ldr;
  }
  if (sv) {
    sv->a1 = a1; sv->ld1 = r.cin * w1; sv->h1 = h1; sv->a2 = a2; sv->ld2 = r.cout * w2;
  }
  return 0;
}

// SpatialTransformer.forward modules/attention.py:325-336, BasicTransformerBlock._forward :265-269
int unet_do_st(Fwd& f, const STW& t, View in, View out, int H, int W, STSaved* sv, Carry* in_carry, Carry* out_carry) {
  mvd_ctx* c = f.c;
  WsScope ws_scope(c, WS_BLOCK);
  const int C = t.C, T = H * W, rows = f.Bv * T;
  const int wi = t.proj_in.xp ? 3 : 1, wo = t.proj_out.xp ? 3 : 1;  // extended precision: [hi | lo | hi] operands
  // Row-chain form (k_rowchain.hip; inference at full batch): everything behind the attention -- to_out + attn2 + t0, LayerNorm3,
  // FF1, GEGLU, FF2, the residual, proj_out + the block input -- is ONE launch with the rows resident in registers; the
  // intermediates t2 / l3 / gg (and t3, unless an extended-precision proj_out wants its [hi | lo | hi] operand) do not exist.
  // Below ~128 workgroups of 128 rows the layered GEMMs fill the chip better than one wave per 32 rows does.
  static const bool no_rc = getenv("MVD_NO_ROWCHAIN") != nullptr;
  static const int rc_min_rows = getenv("MVD_ROWCHAIN_MIN_ROWS") ? atoi(getenv("MVD_ROWCHAIN_MIN_ROWS")) : 16384;
  const bool rc = !no_rc && !f.train && !sv && t.rc_stream && rows >= rc_min_rows && rowchain_takes(C, rows, T) && !(in.ld & 3) && !(out.ld & 3);
  static const bool no_xpf = getenv("MVD_NO_XP_FUSE") != nullptr;  // A/B: extended-precision proj_in / proj_out as separate GEMMs
  // the stream was packed for this precision form of proj_out (rc_po 0: that form of the kernel does not exist at this width)
  const bool rc_po = rc && t.rc_po != 0 && t.rc_po == (t.proj_out.xp ? 2 : 1) && !(no_xpf && t.proj_out.xp) &&
                     rowchain_form_instantiated(C, 1, t.rc_po);
  half_t* n0 = ws_alloc<half_t>(c, (size_t)rows * C * wi);
  float* t0 = ws_alloc<float>(c, (size_t)rows * C);
  half_t* l1 = ws_alloc<half_t>(c, (size_t)rows * C);
  half_t* qkv = ws_alloc<half_t>(c, (size_t)rows * 3 * C);
  half_t* ao = ws_alloc<half_t>(c, (size_t)rows * C);
  float* t2 = rc ? nullptr : ws_alloc<float>(c, (size_t)rows * C);
  // Layered path, plain-precision proj_out, inference: FF2 and proj_out are ONE GEMM over [gg | t2] with K = 5C (STW::ffp); LayerNorm3
  // leaves the fp16 copy of t2 behind the 4C columns of gg, t3 does not exist (MVD_NO_FFP=1: the two layers)
  static const bool ln_scalar = getenv("MVD_LN_SCALAR") != nullptr;
  const bool ffp = !rc && t.ffp.w && !f.train && !sv && !t.proj_out.xp && !ln_scalar && layernorm_slabs_takes(C);
  const int ldg = ffp ? 5 * C : 4 * C;
  half_t* gg = rc ? nullptr : ws_alloc<half_t>(c, (size_t)rows * ldg);
  half_t* t3 = (rc_po || ffp) ? nullptr : ws_alloc<half_t>(c, (size_t)rows * C * wo);  // x + ff(x): only ever the proj_out operand -> fp16
  half_t* l3 = f.train ? ws_alloc<half_t>(c, (size_t)rows * C) : l1;  // the backward pass needs both LayerNorm outputs
  WS_CHECK(n0 && t0 && l1 && qkv && ao && (rc || (t2 && gg)) && (rc_po || ffp || t3) && l3);
  if (in_carry && in_carry->sk > 1)
    RET_IF(run_group_norm(c, in_carry->slabs, C, f.Bv, T, t.norm, 32, 1e-6f, ACT_NONE, nullptr, n0, C * wi, f.s, 0, t.proj_in.xp,
                          in_carry->sk, in_carry->stride, in_carry->bias, in_carry->resid, in_carry->ldr, in.p, in.ld));
  else
    RET_IF(run_group_norm(c, in.p, in.ld, f.Bv, T, t.norm, 32, 1e-6f, ACT_NONE, nullptr, n0, C * wi, f.s, 0, t.proj_in.xp));
  GemmArgs g;
  // A GEMM whose fp32 result is read by a LayerNorm first (proj_in -> LayerNorm1, to_out -> LayerNorm3) leaves its split-K slabs to
  // that LayerNorm when the plan splits K: launch_layernorm_slabs sums them, adds what the reduce pass would have added, writes the
  // finished fp32 row (t0 / t2: the later residuals) and the normalised fp16 row -- no reduce launch (the low-resolution blocks and
  // the few-views-per-rank step, where K is split).  Inference only.
  // MEASURED (profiles/r06_g_ab_ln_defer.txt): 2 launches fewer at the headline, 12 fewer at 2 views per rank, and no time gained
  // (13.20 vs 13.17 ms, 6.49 vs 6.45 ms): a dependent 5 us reduce launch with a warm L2 costs what the wider LayerNorm costs.  Kept
  // as a tested form behind MVD_LN_DEFER=1, off by default.
  static const bool no_ln_defer = getenv("MVD_LN_DEFER") == nullptr || getenv("MVD_NO_DEFER_REDUCE") != nullptr;
  const size_t ln_slab_elems = (size_t)rows * C;
  int sk_ln = 1;
  auto offer_ln_slabs = [&](GemmArgs& ga, int* sk) {
    *sk = 1;
    if (no_ln_defer || f.train || sv || !layernorm_slabs_takes(C) || (c->a2_total & 3) || (t.a2_off & 3)) return;
    const size_t nsl = rows > 8192 ? 4 : 16;
    ga.slabs = ws_alloc<float>(c, nsl * ln_slab_elems);
    ga.slabs_cap = ga.slabs ? nsl * ln_slab_elems : 0;
    ga.sk_used = sk;
  };
  static const bool no_rh = getenv("MVD_NO_ROWHEAD") != nullptr;
  if (rc && !no_rh && t.rh_stream && t.rh_xp == (t.proj_in.xp ? 1 : 0) && !(no_xpf && t.proj_in.xp)) {
    // row-head kernel: proj_in -> t0, LayerNorm1 and the q | k | v projection in one launch (k_rowchain.hip)
    RowHead hp;
    hp.stream = t.rh_stream; hp.rows = rows; hp.n0 = n0; hp.ld_n0 = C * wi; hp.b_pi = t.proj_in.bias; hp.t0 = t0; hp.ld_t0 = C;
    hp.qkv = qkv; hp.ld_qkv = 3 * C;
    const double cc = (double)C * C;
    ProbeScope ps(c, f.s, "rowhead_kernel", 2.0 * rows * cc * 4.0, (double)rows * C * (2.0 + 4.0 + 6.0) + cc * 2.0 * 4.0);
    RET_IF(launch_rowhead(hp, t.rh_xp, f.s));
  } else {
  g.a = n0; g.lda = C * wi; g.w = &t.proj_in; g.out = t0; g.ldc = C;
  offer_ln_slabs(g, &sk_ln);
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  {
    ProbeScope ps(c, f.s, "layernorm", 0.0, (double)rows * C * (6.0 + (sk_ln > 1 ? 4.0 * sk_ln : 0.0)));
    if (sk_ln > 1)  // proj_in split K: LayerNorm1 sums the slabs, adds the bias, writes t0 (the later residual) and l1
      RET_IF(launch_layernorm_slabs(g.slabs, sk_ln, ln_slab_elems, rows, C, t.proj_in.bias, nullptr, 0, T, nullptr, 0, t0, t.ln1.g, t.ln1.b,
                                    1e-5f, l1, f.s));
    else
      RET_IF(launch_layernorm(t0, rows, C, t.ln1.g, t.ln1.b, 1e-5f, l1, f.s));
  }
  // q | k | v projection in one GEMM; the attention kernel transposes V while staging it
  g = GemmArgs();
  g.a = l1; g.lda = C; g.w = &t.qkv; g.out = qkv; g.out_f32 = 0; g.ldc = 3 * C; g.use_bias = false;
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  }
  {
    ProbeScope ps(c, f.s, "attn_kernel", 4.0 * f.Bv * (double)T * T * C, (double)rows * C * 8.0);
    RET_IF(launch_attention(qkv, 3 * C, qkv + 2 * C, 3 * C, ao, C, f.Bv, T, t.heads, C / t.heads, f.s));
  }
  if (rc) {
    RowChain rp;
    memset(&rp, 0, sizeof rp);
    rp.stream = t.rc_stream; rp.rows = rows; rp.T = T;
    rp.ao = ao; rp.ld_ao = C; rp.xin = t0; rp.ld_x = C; rp.b_ao = t.attn_out.bias;
    rp.rowbias = f.a2_all + t.a2_off; rp.rb_ld = c->a2_total;
    if (rc_po) {
      rp.b_po = t.proj_out.bias; rp.resid = in.p; rp.ld_r = in.ld; rp.out = out.p; rp.ld_o = out.ld;
    } else {
      rp.out = t3; rp.ld_o = C * wo; rp.out_split = t.proj_out.xp ? C : 0;
    }
    {
      const double cc = (double)C * C, fl = 2.0 * rows * cc * (rc_po ? 14.0 : 13.0);
      const double by = (double)rows * C * (2.0 + 4.0 + (rc_po ? 8.0 : 2.0 * wo)) + cc * 2.0 * (rc_po ? 14.0 : 13.0);
      ProbeScope ps(c, f.s, "rowchain_kernel", fl, by);
      RET_IF(launch_rowchain(rp, C, 1, rc_po ? t.rc_po : 0, f.s));
    }
    if (rc_po) {
      if (out_carry) {
        out_carry->sk = 1;
        out_carry->stride = (size_t)rows * C;
        out_carry->bias = t.proj_out.bias;
        out_carry->resid = in.p;
        out_carry->ldr = in.ld;
      }
      return 0;
    }
  } else {
  // attn2 (single CLIP token -> per-sample constant, precomputed for all blocks in engine_unet) rides on the
  // attn1 output projection as a per-sample bias
  g = GemmArgs();
  g.a = ao; g.lda = C; g.w = &t.attn_out; g.out = t2; g.ldc = C; g.resid = t0; g.ldr = C;
  g.rowbias = f.a2_all + t.a2_off; g.rb_ld = c->a2_total;
  offer_ln_slabs(g, &sk_ln);
  g.defer_epilogue = g.slabs != nullptr;  // bias, attn2's per-sample row and the residual t0 are then LayerNorm3's to add
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  {
    ProbeScope ps(c, f.s, "layernorm", 0.0, (double)rows * C * (6.0 + (sk_ln > 1 ? 4.0 * (sk_ln + 1) : 0.0)));
    half_t* t2h = ffp ? gg + 4 * C : nullptr;  // fp16 t2 as columns [4C, 5C) of the folded GEMM's operand
    if (sk_ln > 1)
      RET_IF(launch_layernorm_slabs(g.slabs, sk_ln, ln_slab_elems, rows, C, t.attn_out.bias, g.rowbias, g.rb_ld, T, t0, C, t2, t.ln3.g,
                                    t.ln3.b, 1e-5f, l3, f.s, t2h, ldg));
    else
      RET_IF(launch_layernorm(t2, rows, C, t.ln3.g, t.ln3.b, 1e-5f, l3, f.s, t2h, ldg));
  }
  g = GemmArgs();
  g.a = l3; g.lda = C; g.w = &t.ff1; g.out = gg; g.out_f32 = 0; g.ldc = ldg; g.geglu = 1;
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  if (!ffp) {
  g = GemmArgs();
  g.a = gg; g.lda = 4 * C; g.w = &t.ff2; g.out = t3; g.out_f32 = 0; g.ldc = C * wo; g.resid = t2; g.ldr = C;
  g.out_split = t.proj_out.xp ? C : 0;
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  }
  }
  g = GemmArgs();
  g.a = t3; g.lda = C * wo; g.w = &t.proj_out; g.out = out.p; g.ldc = out.ld; g.resid = in.p; g.ldr = in.ld;
  if (ffp) {  // [gg | t2] [W_po W_2 | W_po]^T + (W_po b2 + b_po) + x
    g.a = gg; g.lda = 5 * C; g.w = &t.ffp;
  }
  int skp = 1;
  if (out_carry && out_carry->slabs) {
    g.slabs = out_carry->slabs; g.slabs_cap = out_carry->cap; g.sk_used = &skp; g.defer_epilogue = true;
  }
  RET_IF(run_linear(c, g, f.Bv, rows, f.s));
  if (out_carry) {
    out_carry->sk = skp;
    out_carry->stride = (size_t)rows * C;
    out_carry->bias = ffp ? t.ffp.bias : t.proj_out.bias;
    out_carry->resid = in.p;
    out_carry->ldr = in.ld;
  }
  if (sv) {
    sv->n0 = n0; sv->ldn0 = C * wi; sv->t0 = t0; sv->l1 = l1; sv->qkv = qkv; sv->ao = ao; sv->t2 = t2; sv->l3 = l3; sv->gg = gg;
    sv->t3 = t3; sv->ldt3 = C * wo;
  }
  return 0;
}

namespace {
// DepthTransformer._forward attention.py:78-84 with DepthAttention folded (see k_depth.hip)
// GroupNorm(proj_context(ctx)) without materialising the projection: pass 1 re-computes the 1x1x1 conv tile by tile and keeps
// only (sum, sumsq) per group, pass 2 re-computes it and applies scale/shift + ReLU in the epilogue.  Depends on the context
// volume and the weights only, so engine_unet issues it on the side stream for every DepthTransformer up front.
bool ctx_fold_ok(const Fwd& f, const CondW& d, int HW, int D, int level) {
  static const bool fold_off = getenv("MVD_NO_CTX_FOLD") != nullptr;
  const int rps = D * HW, cpg = d.Cc / 8;
  return !fold_off && f.n_ctx > 0 && f.src16[level] && rps % 256 == 0 && (cpg == 8 || cpg == 16 || cpg == 32) &&
         (long)f.n_ctx * HW * D >= 512;
}
// the same for every DepthTransformer of one context level at once (mvd_ctx::CtxGroup): cn_all [n_ctx * HW * D][nblk * Cc]
int ctx_fold_group(Fwd& f, const mvd_ctx::CtxGroup& gp, int HW, int D, half_t* cn_all, hipStream_t s) {
  mvd_ctx* c = f.c;
  const int Cc = gp.Cc, N = gp.nblk * Cc, crow = f.n_ctx * HW, rps = D * HW, cpg = Cc / 8, ntile = rps / 256, G = gp.nblk * 8;
  float* part = ws_alloc<float>(c, (size_t)f.n_ctx * ntile * G * 2);
  float* sc = ws_alloc<float>(c, (size_t)f.n_ctx * N);
  float* sh = ws_alloc<float>(c, (size_t)f.n_ctx * N);
  WS_CHECK(part && sc && sh);
  GemmArgs g;
  g.a = f.src16[gp.level]; g.lda = Cc; g.w = &gp.w; g.use_bias = false; g.gn_partial = part; g.gn_cpg = cpg; g.force_splitk = 1;
  RET_IF(run_linear(c, g, f.n_ctx, crow * D, s));
  RET_IF(launch_gn_finalize(part, f.n_ctx, ntile, rps, N, G, gp.gn.g, gp.gn.b, 1e-5f, sc, sh, N, s));
  g = GemmArgs();
  g.a = f.src16[gp.level]; g.lda = Cc; g.w = &gp.w; g.use_bias = false; g.out = cn_all; g.out_f32 = 0; g.ldc = N;
  g.rowscale = sc; g.rs_ld = N; g.rowbias = sh; g.rb_ld = N; g.act = ACT_RELU; g.force_splitk = 1;
  return run_linear(c, g, f.n_ctx, crow * D, s);
}

int ctx_fold(Fwd& f, const CondW& d, int HW, int D, int level, half_t* cn, hipStream_t s) {
  mvd_ctx* c = f.c;
  const int Cc = d.Cc, crow = f.n_ctx * HW, rps = D * HW, cpg = Cc / 8, ntile = rps / 256;
  float* part = ws_alloc<float>(c, (size_t)f.n_ctx * ntile * 8 * 2);
  float* sc = ws_alloc<float>(c, (size_t)f.n_ctx * Cc);
  float* sh = ws_alloc<float>(c, (size_t)f.n_ctx * Cc);
  WS_CHECK(part && sc && sh);
  GemmArgs g;
  g.a = f.src16[level]; g.lda = Cc; g.w = &d.proj_ctx; g.use_bias = false; g.gn_partial = part; g.gn_cpg = cpg; g.force_splitk = 1;
  RET_IF(run_linear(c, g, f.n_ctx, crow * D, s));
  RET_IF(launch_gn_finalize(part, f.n_ctx, ntile, rps, Cc, 8, d.gn_ctx.g, d.gn_ctx.b, 1e-5f, sc, sh, Cc, s));
  g = GemmArgs();
  g.a = f.src16[level]; g.lda = Cc; g.w = &d.proj_ctx; g.use_bias = false; g.out = cn; g.out_f32 = 0; g.ldc = Cc;
  g.rowscale = sc; g.rs_ld = Cc; g.rowbias = sh; g.rb_ld = Cc; g.act = ACT_RELU; g.force_splitk = 1;
  return run_linear(c, g, f.n_ctx, crow * D, s);
}

}  // namespace

int unet_do_cond(Fwd& f, const CondW& d, View in, View out, int H, int W, int level, int cond_idx) {
  mvd_ctx* c = f.c;
  WsScope ws_scope(c, WS_BLOCK);
  const int HW = H * W, I = d.I, Cc = d.Cc, D = f.depth0 >> level;
  const int crow = f.n_ctx * HW;
  // Samples without context (CFG's unconditional half) get x + K (mvd_ctx::CondConst): the block below then runs on the Bx = n_ctx
  // samples WITH context only -- half the rows of proj_in, the three GroupNorms, the output projection and both 3x3 convolutions.
  // MVD_NO_COND_CONST=1: the round-5 form (every layer over all samples, the context-free rows' z filled with relu(beta)).
  static const bool no_const = getenv("MVD_NO_COND_CONST") != nullptr;
  const int n_free = f.Bv - f.n_ctx;
  const bool use_const = !no_const && !f.train && !c->train_mode && cond_idx >= 0 && n_free > 0 && f.n_ctx > 0 && !(d.dim & 3) &&
                         !(in.ld & 3) && !(out.ld & 3);
  if (use_const) {
    if (c->cond_const.size() < c->conds.size()) c->cond_const.resize(c->conds.size());
    mvd_ctx::CondConst& cc = c->cond_const[cond_idx];
    if (!cc.valid || cc.H != H || cc.W != W) {
      if (cc.k && (cc.H != H || cc.W != W)) {
        HIP_CHECK_RET(hipFree(cc.k));
        cc.k = nullptr;
      }
      if (!cc.k) HIP_CHECK_RET(hipMalloc((void**)&cc.k, (size_t)HW * d.dim * sizeof(float)));
      float* zero = ws_alloc<float>(c, (size_t)HW * d.dim);
      WS_CHECK(zero);
      HIP_CHECK_RET(hipMemsetAsync(zero, 0, (size_t)HW * d.dim * sizeof(float), f.s));
      Fwd f0 = f;  // one context-free sample: the block's residual branch on its own (the input is zero, so out = K)
      f0.Bv = 1;
      f0.n_ctx = 0;
      View in0, out0;
      in0.p = zero; in0.ld = d.dim; in0.C = d.dim;
      out0.p = cc.k; out0.ld = d.dim; out0.C = d.dim;
      RET_IF(unet_do_cond(f0, d, in0, out0, H, W, level, -1));
      cc.H = H;
      cc.W = W;
      cc.valid = true;
    }
  }
  const int Bx = use_const ? f.n_ctx : f.Bv, rows = Bx * HW;
  // extended precision (ConvW::xp): operands are [hi | lo | hi]
  const int xq = d.wqk.xp, xo = d.wov.xp, x1 = d.conv1.xp, x2 = d.conv2.xp;
  const int wpn = (xq || x1 || x2) ? 3 : 1, wz = xo ? 3 : 1;
  float* p = ws_alloc<float>(c, (size_t)rows * I);
  half_t* pn = ws_alloc<half_t>(c, (size_t)rows * I * wpn);
  half_t* z = ws_alloc<half_t>(c, (size_t)rows * 4 * Cc * wz);
  float* o = ws_alloc<float>(c, (size_t)rows * I);
  float* o2 = ws_alloc<float>(c, (size_t)rows * I);
  WS_CHECK(p && pn && z && o && o2);
  // The three GEMMs of this block whose result only feeds a GroupNorm (proj_in -> gn_in, the folded output projection -> gn_o1,
  // conv1 -> gn_o2) leave their split-K slabs to that norm when the plan splits K (GemmArgs::slabs, as ResBlock conv1 does): the
  // reduce pass -- a launch, a write and a read of the tensor -- disappears (round 6: 24 of the headline step's 46 reduce launches
  // sat in front of a GroupNorm here; MVD_NO_COND_DEFER=1 restores them).  Inference only: the backward pass reads p / o / o2.
  static const bool no_cdefer = getenv("MVD_NO_COND_DEFER") != nullptr || getenv("MVD_NO_DEFER_REDUCE") != nullptr ||
                                getenv("MVD_GN_TWO_PASS") != nullptr;
  const size_t slab_elems = (size_t)rows * I;
  auto offer_slabs = [&](GemmArgs& ga, int* sk, int ldo) {
    *sk = 1;
    if (no_cdefer || f.train || !gn_group_eligible(I, HW, I, 8, 0, ldo)) return;
    const size_t nsl = rows > 8192 ? 4 : 16;
    ga.slabs = ws_alloc<float>(c, nsl * slab_elems);
    ga.slabs_cap = ga.slabs ? nsl * slab_elems : 0;
    ga.sk_used = sk;
  };
  GemmArgs g;
  int skd = 1;
  g.a = in.p; g.a_f32 = 1; g.lda = in.ld; g.w = &d.proj_in; g.out = p; g.ldc = I;
  offer_slabs(g, &skd, xq ? 3 * I : I);
  RET_IF(run_linear(c, g, Bx, rows, f.s));
  if (skd > 1)
    RET_IF(run_group_norm(c, g.slabs, I, Bx, HW, d.gn_in, 8, 1e-5f, ACT_SILU, nullptr, pn, xq ? 3 * I : I, f.s, 0, xq, skd, slab_elems,
                          d.proj_in.bias));
  else
    RET_IF(run_group_norm(c, p, I, Bx, HW, d.gn_in, 8, 1e-5f, ACT_SILU, nullptr, pn, xq ? 3 * I : I, f.s, 0, xq));
  if (f.n_ctx > 0) {
    float* qk = ws_alloc<float>(c, (size_t)crow * 4 * Cc);
    half_t* cn = cond_idx >= 0 && f.cn_pre[cond_idx] ? nullptr : ws_alloc<half_t>(c, (size_t)crow * D * Cc);
    WS_CHECK(qk && (cn || (cond_idx >= 0 && f.cn_pre[cond_idx])));
    g = GemmArgs();
    g.a = pn; g.lda = xq ? 3 * I : I; g.w = &d.wqk; g.out = qk; g.ldc = 4 * Cc; g.use_bias = false;
    RET_IF(run_linear(c, g, f.n_ctx, crow, f.s));
    const half_t* cnp = cond_idx >= 0 ? f.cn_pre[cond_idx] : nullptr;
    if (cnp) {  // prepared on the side stream
      HIP_CHECK_RET(hipStreamWaitEvent(f.s, c->ev_cond[cond_idx], 0));
      cn = const_cast<half_t*>(cnp);
    } else if (ctx_fold_ok(f, d, HW, D, level)) {
      if (f.ctx_side) HIP_CHECK_RET(hipStreamWaitEvent(f.s, c->ev_ctx, 0));
      RET_IF(ctx_fold(f, d, HW, D, level, cn, f.s));
    } else {
      if (f.ctx_side) HIP_CHECK_RET(hipStreamWaitEvent(f.s, c->ev_ctx, 0));
      float* pc = ws_alloc<float>(c, (size_t)crow * D * Cc);
      WS_CHECK(pc);
      g = GemmArgs();
      g.a = f.src[level].p; g.a_f32 = f.src[level].f32; g.lda = Cc; g.w = &d.proj_ctx; g.out = pc; g.ldc = Cc; g.use_bias = false;
      RET_IF(run_linear(c, g, f.n_ctx, crow * D, f.s));
      RET_IF(run_group_norm(c, pc, Cc, f.n_ctx, D * HW, d.gn_ctx, 8, 1e-5f, ACT_RELU, nullptr, cn, Cc, f.s));
    }
    {
      ProbeScope ps(c, f.s, "depth_attn_kernel", 4.0 * crow * (double)D * 4 * Cc,
                    (double)crow * D * Cc * 2.0 + (double)crow * 4 * Cc * 6.0);
      // the rows of the unconditional samples (all-zero context: GN(0) = beta, uniform softmax -> z = relu(beta) for every
      // head) are filled by the same launch
      RET_IF(launch_depth_attn(qk, cn, z, f.n_ctx, HW, D, Cc, 4, f.s, xo, rows - crow, d.relu_beta, cnp ? f.cn_ld[cond_idx] : Cc));
    }
  } else if (Bx > f.n_ctx) {
    RET_IF(launch_fill_rows_f16(z + (size_t)crow * 4 * Cc * wz, 4 * Cc * wz, rows - crow, d.relu_beta, 4 * Cc * wz, f.s));
  }
  g = GemmArgs();
  g.a = z; g.lda = 4 * Cc * wz; g.w = &d.wov; g.out = o; g.ldc = I; g.use_bias = false;
  offer_slabs(g, &skd, x1 ? 3 * I : I);
  RET_IF(run_linear(c, g, Bx, rows, f.s));
  if (skd > 1)
    RET_IF(run_group_norm(c, g.slabs, I, Bx, HW, d.gn_o1, 8, 1e-5f, ACT_RELU, nullptr, pn, x1 ? 3 * I : I, f.s, 0, x1, skd, slab_elems));
  else
    RET_IF(run_group_norm(c, o, I, Bx, HW, d.gn_o1, 8, 1e-5f, ACT_RELU, nullptr, pn, x1 ? 3 * I : I, f.s, 0, x1));
  g = GemmArgs();
  g.a = pn; g.lda = x1 ? 3 * I : I; g.w = &d.conv1; g.out = o2; g.ldc = I; g.use_bias = false;
  offer_slabs(g, &skd, x2 ? 3 * I : I);
  RET_IF(run_conv2d(c, g, Bx, H, W, 1, 0, f.s));
  if (skd > 1)
    RET_IF(run_group_norm(c, g.slabs, I, Bx, HW, d.gn_o2, 8, 1e-5f, ACT_RELU, nullptr, pn, x2 ? 3 * I : I, f.s, 0, x2, skd, slab_elems));
  else
    RET_IF(run_group_norm(c, o2, I, Bx, HW, d.gn_o2, 8, 1e-5f, ACT_RELU, nullptr, pn, x2 ? 3 * I : I, f.s, 0, x2));
  g = GemmArgs();
  g.a = pn; g.lda = x2 ? 3 * I : I; g.w = &d.conv2; g.out = out.p; g.ldc = out.ld; g.use_bias = false; g.resid = in.p; g.ldr = in.ld;
  RET_IF(run_conv2d(c, g, Bx, H, W, 1, 0, f.s));
  if (use_const) {  // the context-free samples (behind the others in the batch): x + K
    ProbeScope ps(c, f.s, "cond_const_add", 0.0, (double)n_free * HW * d.dim * 8.0);
    RET_IF(launch_add_image_rows(in.p + (size_t)crow * in.ld, in.ld, c->cond_const[cond_idx].k, d.dim, n_free, HW,
                                 out.p + (size_t)crow * out.ld, out.ld, f.s));
  }
  return 0;
}

int unet_do_op(Fwd& f, const UOp& op, View in, View out, int& H, int& W, StageRec* rec, Carry* in_carry, Carry* out_carry) {
  mvd_ctx* c = f.c;
  const bool keep = rec && c->ws.hold == 2;  // keep-all tape: the block's intermediates stay valid
  if (rec) rec->have_saved = keep;
  switch (op.kind) {
    case OP_RES: return unet_do_res(f, c->res[op.idx], in, out, H, W, keep ? &rec->rs : nullptr, in_carry, out_carry);
    case OP_ST: return unet_do_st(f, c->st[op.idx], in, out, H, W, keep ? &rec->ss : nullptr, in_carry, out_carry);
    case OP_CONV_IN:
    case OP_DOWN:
    case OP_UP: {
      if (in_carry && in_carry->sk > 1) return mvd_fail("unet_do_op: a convolution cannot take a deferred input");
      GemmArgs g;
      g.a = in.p; g.a_f32 = 1; g.lda = in.ld; g.w = &c->convs[op.idx]; g.out = out.p; g.ldc = out.ld;
      int skc = 1;
      const bool can_defer = out_carry && out_carry->slabs && op.kind == OP_DOWN && !c->convs[op.idx].xp;
      if (can_defer) {
        g.slabs = out_carry->slabs; g.slabs_cap = out_carry->cap; g.sk_used = &skc; g.defer_epilogue = true;
      }
      static const bool no_in32 = getenv("MVD_NO_CONV_IN_F32") != nullptr;
      if (op.kind == OP_CONV_IN && !f.train && !no_in32 && c->convs[op.idx].w32) {
        // inference: the first convolution in exact fp32 on the vector ALU (K = 72 is all staging for the MFMA forms)
        const ConvW& cw = c->convs[op.idx];
        ProbeScope ps(c, f.s, "conv_in_f32_kernel", 2.0 * f.Bv * H * W * 72.0 * cw.N,
                      (double)f.Bv * H * W * (cw.N + 8) * 4.0);
        RET_IF(launch_conv_in_f32(in.p, in.ld, cw.w32, cw.cin_src, cw.bias, cw.N, f.Bv, H, W, out.p, out.ld, f.s));
        if (out_carry) {
          out_carry->sk = 1;
          out_carry->stride = (size_t)f.Bv * H * W * op.cout;
          out_carry->bias = cw.bias;
          out_carry->resid = nullptr;
          out_carry->ldr = 0;
        }
        return 0;
      }
      const int stride = op.kind == OP_DOWN ? 2 : 1, ups = op.kind == OP_UP ? 1 : 0;
      WsScope ws_scope(c, WS_BLOCK);
      if (c->convs[op.idx].xp) {  // extended precision (conv_in): fp32 source -> [hi | lo | hi] copy
        const int Cl = c->convs[op.idx].cin_l;
        half_t* as = ws_alloc<half_t>(c, (size_t)f.Bv * H * W * 3 * Cl);
        WS_CHECK(as);
        RET_IF(launch_rows_f32_to_f16_split(in.p, in.ld, (long)f.Bv * H * W, Cl, as, f.s));
        g.a = as; g.a_f32 = 0; g.lda = 3 * Cl;
      }
      // weight-streaming regime (few pixels): the 9-tap form moves 9 slabs instead of 16
      static const bool no_up3x = getenv("MVD_NO_UP_CONV3X") != nullptr;
      const ConvW& cw = c->convs[op.idx];
      if (ups && cw.w_up && f.Bv * H * W >= 2048) {
        RET_IF(run_upconv2d(c, g, f.Bv, H, W, f.s));
      } else if (ups && !no_up3x && !f.train && cw.wx && !cw.xp && 2 * H == cw.res_out && 2 * W == cw.res_out && c->use_halo && !(in.ld & 3) &&
                 !(op.cin & 3)) {
        // 4 x 4 -> 8 x 8: the nearest-upsampled image as fp16 (one small launch), then conv3x over whole 8 x 8 images (round 6: the
        // register-staged 9-tap GEMM on the fp32 source ran this layer at 476 TFLOP/s, 127 us of the step)
        half_t* up = ws_alloc<half_t>(c, (size_t)f.Bv * 4 * H * W * op.cin);
        WS_CHECK(up);
        RET_IF(launch_upsample2_f16(in.p, in.ld, f.Bv, H, W, op.cin, up, f.s));
        g.a = up; g.a_f32 = 0; g.lda = op.cin;
        RET_IF(run_conv2d(c, g, f.Bv, 2 * H, 2 * W, 1, 0, f.s));
      } else {
        RET_IF(run_conv2d(c, g, f.Bv, H, W, stride, ups, f.s));
      }
      if (op.kind == OP_DOWN) { H = (H - 1) / 2 + 1; W = (W - 1) / 2 + 1; }
      if (out_carry) {
        out_carry->sk = skc;
        out_carry->stride = (size_t)f.Bv * H * W * op.cout;
        out_carry->bias = c->convs[op.idx].bias;
        out_carry->resid = nullptr;
        out_carry->ldr = 0;
      }
      if (op.kind == OP_UP) { H *= 2; W *= 2; }
      return 0;
    }
  }
  return mvd_fail("unknown op");
}

namespace {
int out_res_of(const std::vector<UOp>& ops, int H) {
  for (auto& o : ops) {
    if (o.kind == OP_DOWN) H = (H - 1) / 2 + 1;
    if (o.kind == OP_UP) H *= 2;
  }
  return H;
}

}  // namespace

// Per-sample constants of a forward: the timestep embedding through the time_embed MLP and every ResBlock's emb projection
// (openaimodel.py:728-729, 264-269: [Bv][emb_total]) and the folded attn2 output of every SpatialTransformer for the single CLIP
// token ([Bv][a2_total]).  Allocated in the caller's workspace scope.  t / context may be null (the part is skipped).
int unet_embeddings(mvd_ctx* c, const int64_t* t, const float* context, int Bv, hipStream_t s, float** e0_out, float** e1_out,
                    float** e2_out, float** ea_out, float** a2_out) {
  const mvd_unet_config& u = c->u;
  const int mc = u.model_channels, temb = 4 * mc;
  if (t) {
    float* e0 = ws_alloc<float>(c, (size_t)Bv * mc);
    float* e1 = ws_alloc<float>(c, (size_t)Bv * temb);
    float* e2 = ws_alloc<float>(c, (size_t)Bv * temb);
    float* ea = ws_alloc<float>(c, (size_t)Bv * c->emb_total);
    WS_CHECK(e0 && e1 && e2 && ea);
    RET_IF(launch_timestep_embedding(t, Bv, mc, e0, s));
    // time_embed MLP and every ResBlock's emb projection as three weight-streaming GEMMs (M = Bv rows)
    ConvW w0, w2, wa;
    w0.w = c->te0.w; w0.bias = c->te0.bias; w0.N = temb; w0.Cin = mc;
    w2.w = c->te2.w; w2.bias = c->te2.bias; w2.N = temb; w2.Cin = temb;
    wa.w = c->emb_all.w; wa.bias = c->emb_all.bias; wa.N = c->emb_total; wa.Cin = temb;
    GemmArgs g;
    g.a = e0; g.a_f32 = 1; g.lda = mc; g.w = &w0; g.out = e1; g.ldc = temb; g.act = ACT_SILU;
    RET_IF(run_linear(c, g, Bv, Bv, s));
    g = GemmArgs();
    g.a = e1; g.a_f32 = 1; g.lda = temb; g.w = &w2; g.out = e2; g.ldc = temb; g.act = ACT_SILU;  // emb is only used as silu(emb)
    RET_IF(run_linear(c, g, Bv, Bv, s));
    g = GemmArgs();
    g.a = e2; g.a_f32 = 1; g.lda = temb; g.w = &wa; g.out = ea; g.ldc = c->emb_total;
    RET_IF(run_linear(c, g, Bv, Bv, s));
    if (e0_out) *e0_out = e0;
    if (e1_out) *e1_out = e1;
    if (e2_out) *e2_out = e2;
    *ea_out = ea;
  }
  if (context) {
    float* a2 = ws_alloc<float>(c, (size_t)Bv * c->a2_total);
    WS_CHECK(a2);
    GemmArgs g;
    g.a = context; g.a_f32 = 1; g.lda = u.context_dim; g.w = &c->a2_all; g.out = a2; g.ldc = c->a2_total;
    RET_IF(run_linear(c, g, Bv, Bv, s));
    *a2_out = a2;
  }
  return 0;
}

int engine_side_init(mvd_ctx* c) {
  if (c->side) return 0;
  HIP_CHECK_RET(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  for (hipEvent_t* ev : {&c->ev_fork, &c->ev_join, &c->ev_join2, &c->ev_ctx, &c->ev_emb0, &c->ev_emb})
    HIP_CHECK_RET(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  c->ev_cond.resize(c->conds.size());
  for (auto& ev : c->ev_cond) HIP_CHECK_RET(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  return 0;
}

int engine_unet(mvd_ctx* c, const float* x_nhwc, int x_ld, const int64_t* t, const float* context, int Bv, int n_ctx,
                int depth0, const Ctx5 src[4], float* eps_nhwc, hipStream_t s, const CtxProducer* produce, TrainTape* tape) {
  if (!c->finalized || !c->has_unet) return mvd_fail("UNet weights not uploaded / finalized");
  if (tape && (produce || n_ctx != Bv)) return mvd_fail("engine_unet: the training forward takes every sample's context volumes");
  const mvd_unet_config& u = c->u;
  const int mc = u.model_channels;
  WsScope ws_scope0(c);
  Fwd f{c, s, Bv, n_ctx, depth0, nullptr, context, nullptr, src, {nullptr, nullptr, nullptr, nullptr}};
  f.train = tape != nullptr;
  // The context volumes (frustum network) and the context half of every DepthTransformer feed nothing before the middle
  // block.  Issued on the side stream they run beside the trunk (which leaves CUs idle at the lower resolutions and whenever
  // a rank holds few views); each DepthTransformer waits for its own event just before its depth attention.  The guard joins
  // the side stream back on every exit path, so the workspace these launches use is never handed out again while they may
  // still be running.
  SideJoin join(s);
  static const bool side_off = getenv("MVD_NO_SIDE_STREAM") != nullptr;
  // (the training forward stays on one stream: its intermediates are the tape, ordered with the backward pass that follows)
  const bool use_side = !side_off && !tape && n_ctx > 0 && src && c->conds.size() <= 16;
  bool forked = false;
  // fork_ctx: everything that produces or only reads the context volumes.  With a producer (the frustum network of
  // mvd_denoise_views) it is called after the full-resolution input blocks, so the side stream shares the CUs with the
  // lower-resolution blocks (which do not fill them) and not with the level-0 convs; without one, at the start.
  auto fork_ctx = [&]() -> int {
    forked = true;
    hipStream_t cs = s;  // the stream the context volumes are produced / read on
    if (use_side) {
      RET_IF(engine_side_init(c));
      // two-way handshake: the side stream starts after everything enqueued on the caller's stream so far, and the caller's
      // stream resumes only once the side stream has passed that point (without the acknowledgement the step
      // was not bit-reproducible in ~5 % of runs)
      // MVD_ONE_WAY_FORK re-creates the round-1 non-determinism on demand (tools/det_fork.sh; the pad /
      // spin / host-sync variants of that investigation are in the history: commit 88b2973)
      static const bool one_way = getenv("MVD_ONE_WAY_FORK") != nullptr;
      HIP_CHECK_RET(hipEventRecord(c->ev_fork, s));
      HIP_CHECK_RET(hipStreamWaitEvent(c->side, c->ev_fork, 0));
      if (!one_way) {
        HIP_CHECK_RET(hipEventRecord(c->ev_join2, c->side));
        HIP_CHECK_RET(hipStreamWaitEvent(s, c->ev_join2, 0));
      }
      join.side = c->side;
      join.ev = c->ev_join;
      cs = c->side;
    }
    f.ctx_side = use_side;
    if (produce) {
      // the producer's scratch is freed scope by scope on the host while the side stream may still be using it: keep it
      // allocated until this forward's own scope ends (after the join)
      c->ws.hold = use_side ? 1 : 0;
      const int r = (*produce)(cs);
      c->ws.hold = 0;
      RET_IF(r);
    }
    // fp16 view of the context volumes (operand-only tensors); fp32 sources are copied once per forward
    for (int l = 0; l < 4 && n_ctx > 0 && src; ++l) {
      if (!src[l].p) continue;
      if (!src[l].f32) {
        f.src16[l] = (const half_t*)src[l].p;
        continue;
      }
      const int Dl = depth0 >> l, Sl = u.image_size >> l;
      const size_t n = (size_t)n_ctx * Dl * Sl * Sl * u.volume_dims[l];
      half_t* h = ws_alloc<half_t>(c, n);
      WS_CHECK(h);
      RET_IF(launch_f32_to_f16((const float*)src[l].p, h, n, cs));
      f.src16[l] = h;
    }
    if (!use_side) return 0;
    HIP_CHECK_RET(hipEventRecord(c->ev_ctx, c->side));  // volumes + fp16 views complete: inline consumers wait for this
    auto level_of = [&](int Hx) {
      int l = 0;
      for (int r = u.image_size; r > Hx; r >>= 1) ++l;
      return l;
    };
    std::vector<int> cH(c->conds.size(), 0);
    int Hc = u.image_size;
    for (auto& blk : c->in_blocks) Hc = out_res_of(blk, Hc);
    cH[0] = out_res_of(c->mid_block, Hc);
    for (size_t i = 0; i < c->out_blocks.size(); ++i) {
      Hc = out_res_of(c->out_blocks[i], Hc);
      if (i >= 3 && 1 + (i - 3) < cH.size()) cH[1 + (i - 3)] = Hc;
    }
    // the blocks of one level together (mvd_ctx::CtxGroup: the volume is read once per pass and level); MVD_NO_CTX_GROUP=1: one by one
    static const bool no_group = getenv("MVD_NO_CTX_GROUP") != nullptr;
    std::vector<char> grouped(c->conds.size(), 0);
    for (const mvd_ctx::CtxGroup& gp : c->ctx_groups) {
      if (no_group) break;
      bool ok = gp.nblk >= 2;
      int Hk = 0;
      for (int j = 0; j < gp.nblk && ok; ++j) {
        const int k = gp.cond[j];
        ok = k < (int)cH.size() && cH[k] > 0 && level_of(cH[k]) == gp.level && (Hk == 0 || cH[k] == Hk) && c->conds[k].Cc == gp.Cc;
        if (ok) Hk = cH[k];
      }
      if (!ok) continue;
      const int HW = Hk * Hk, D = depth0 >> gp.level, N = gp.nblk * gp.Cc;
      if (!ctx_fold_ok(f, c->conds[gp.cond[0]], HW, D, gp.level) || gp.nblk * 8 > 32) continue;
      half_t* cn_all = ws_alloc<half_t>(c, (size_t)n_ctx * HW * D * N);
      WS_CHECK(cn_all);
      RET_IF(ctx_fold_group(f, gp, HW, D, cn_all, c->side));
      for (int j = 0; j < gp.nblk; ++j) {
        const int k = gp.cond[j];
        HIP_CHECK_RET(hipEventRecord(c->ev_cond[k], c->side));
        f.cn_pre[k] = cn_all + (size_t)j * gp.Cc;
        f.cn_ld[k] = N;
        grouped[k] = 1;
      }
    }
    for (size_t k = 0; k < c->conds.size(); ++k) {
      if (cH[k] <= 0 || grouped[k]) continue;
      const CondW& d = c->conds[k];
      const int lv = level_of(cH[k]), HW = cH[k] * cH[k], D = depth0 >> lv;
      if (lv > 3 || !ctx_fold_ok(f, d, HW, D, lv)) continue;
      half_t* cn = ws_alloc<half_t>(c, (size_t)n_ctx * HW * D * d.Cc);
      WS_CHECK(cn);
      RET_IF(ctx_fold(f, d, HW, D, lv, cn, c->side));
      HIP_CHECK_RET(hipEventRecord(c->ev_cond[k], c->side));
      f.cn_pre[k] = cn;
      f.cn_ld[k] = d.Cc;
    }
    return 0;
  };
  if (!produce) RET_IF(fork_ctx());
  // timestep embedding -> MLP -> every ResBlock's emb projection in one pass
  // Nothing before the first ResBlock reads them (the input convolution does not), and they are five weight-streaming GEMMs
  // of Bv rows on a few CUs: on the side stream they run beside the input convolution instead of in front of it (the caller's
  // stream waits for them behind the first input block).  Their split-K scratch is freed by host scopes while the side
  // stream may still use it: held, like the context producer's (Workspace::hold).
  float *e0 = nullptr, *e1 = nullptr, *e2 = nullptr, *ea = nullptr, *a2 = nullptr;
  static const bool side_emb_off = getenv("MVD_NO_SIDE_EMB") != nullptr;
  const bool side_emb = use_side && produce && !side_emb_off && c->in_blocks.size() > 1 && c->ws.hold == 0;  // (without a producer the side stream is already busy with the context folds)
  if (side_emb) {
    RET_IF(engine_side_init(c));
    HIP_CHECK_RET(hipEventRecord(c->ev_emb0, s));  // t / context were written on the caller's stream
    HIP_CHECK_RET(hipStreamWaitEvent(c->side, c->ev_emb0, 0));
    join.side = c->side;
    join.ev = c->ev_join;
    c->ws.hold = 1;
    const int r = unet_embeddings(c, t, context, Bv, c->side, &e0, &e1, &e2, &ea, &a2);
    c->ws.hold = 0;
    RET_IF(r);
    HIP_CHECK_RET(hipEventRecord(c->ev_emb, c->side));
  } else {
    RET_IF(unet_embeddings(c, t, context, Bv, s, &e0, &e1, &e2, &ea, &a2));
  }
  f.emb_all = ea;
  f.a2_all = a2;
  if (tape) {
    tape->e0 = e0; tape->e1 = e1; tape->e2 = e2; tape->ea = ea; tape->context = context;
    tape->Bv = Bv; tape->depth0 = depth0; tape->src = src;
    tape->a2 = a2;
  }

  // shapes of the concat buffers
  const int nb = (int)c->in_blocks.size();
  std::vector<int> in_ch(nb), in_res(nb);
  {
    int H = u.image_size;
    for (int j = 0; j < nb; ++j) {
      H = out_res_of(c->in_blocks[j], H);
      in_ch[j] = c->in_blocks[j].back().cout;
      in_res[j] = H;
    }
  }
  std::vector<int> h_ch(nb + 1), cat_C(nb);
  std::vector<float*> cat(nb);
  h_ch[0] = in_ch[nb - 1];
  for (int i = 0; i < nb; ++i) {
    const int j = nb - 1 - i;
    cat_C[i] = h_ch[i] + in_ch[j];
    cat[i] = ws_alloc<float>(c, (size_t)Bv * in_res[j] * in_res[j] * cat_C[i]);
    WS_CHECK(cat[i]);
    h_ch[i + 1] = c->out_blocks[i].back().cout;
  }
  float* final_h = ws_alloc<float>(c, (size_t)Bv * u.image_size * u.image_size * mc);
  WS_CHECK(final_h);
  if (tape) {
    tape->cat = cat; tape->cat_C = cat_C; tape->h_ch = h_ch; tape->in_ch = in_ch; tape->in_res = in_res; tape->final_h = final_h;
  }
  int chain_id = 0;

  // A block whose first layer is a single-pass GroupNorm over exactly the previous block's output can take that output as
  // split-K slabs (Carry): the previous block's reduce pass and this GroupNorm become one launch.
  static const bool no_carry = getenv("MVD_NO_CARRY") != nullptr;
  auto takes_carry = [&](const UOp& nx, int Hn) -> bool {
    if (no_carry || tape || (long)Bv * Hn * Hn > 8192) return false;  // (full resolution never splits K)
    const int C = nx.cin;
    if (nx.kind == OP_RES) return gn_group_eligible(C, Hn * Hn, C, 32, 0, C * (c->res[nx.idx].c1.xp ? 3 : 1));
    if (nx.kind == OP_ST) return gn_group_eligible(C, Hn * Hn, C, 32, 0, C * (c->st[nx.idx].proj_in.xp ? 3 : 1));
    return false;
  };
  auto carry_storage = [&](Carry& cy, size_t elems) {  // may leave the pointers null (no workspace): then nothing is deferred
    cy = Carry();
    cy.slabs = ws_alloc<float>(c, 16 * elems);
    cy.cap = cy.slabs ? 16 * elems : 0;
    cy.aux = cy.slabs ? ws_alloc<float>(c, elems) : nullptr;
    cy.aux_cap = cy.aux ? elems : 0;
  };
  // in_carry: the chain's input was left as slabs by the previous chain; out_carry: the chain's last block may leave its
  // output as slabs for the next chain's first block (the caller checked takes_carry).  keep_temps: the chain's
  // intermediate tensors stay allocated after it returns (a carried output's residual operand may be one of them).
  auto run_chain = [&](const std::vector<UOp>& ops, const CondW* cond, View in, View dst, int& H, int& W, Carry* in_carry,
                       Carry* out_carry, bool keep_temps) -> int {
    struct OptScope {
      mvd_ctx* c;
      size_t mark;
      bool on;
      ~OptScope() {
        if (on && !c->ws.hold) c->ws.off = mark;
      }
    } ws_scope{c, c->ws.off, !keep_temps};
    View cur = in;
    const int nstage = (int)ops.size() + (cond ? 1 : 0);
    Carry local[2];
    Carry* prev = in_carry;
    for (int k = 0; k < nstage; ++k) {
      const bool last = k == nstage - 1;
      const bool is_cond = cond && k == (int)ops.size();
      const int cout = is_cond ? cond->dim : ops[k].cout;
      int Ho = H;
      if (!is_cond) Ho = out_res_of({ops[k]}, H);
      View o;
      if (last) o = dst;
      else {
        o.p = ws_alloc<float>(c, (size_t)Bv * Ho * Ho * cout);
        WS_CHECK(o.p);
        o.ld = cout;
      }
      o.C = cout;
      StageRec rec;
      rec.chain = chain_id;
      rec.in = cur;
      rec.out = o;
      rec.H = H;
      rec.W = W;
      for (int r = u.image_size; r > H; r >>= 1) ++rec.level;
      Carry* oc = nullptr;
      if (is_cond) {
        if (prev && prev->sk > 1) return mvd_fail("run_chain: a DepthTransformer cannot take a deferred input");
        rec.kind = OP_COND;
        rec.idx = (int)(cond - c->conds.data());
        RET_IF(unet_do_cond(f, *cond, cur, o, H, W, rec.level, rec.idx));
      } else {
        if (!last) {
          const bool next_is_cond = cond && k + 1 == (int)ops.size();
          if (!next_is_cond && takes_carry(ops[k + 1], Ho)) {
            oc = &local[k & 1];
            carry_storage(*oc, (size_t)Bv * Ho * Ho * cout);
          }
        } else if (out_carry) {
          oc = out_carry;
        }
        if (oc) {
          oc->sk = 1;
          oc->resid = oc->bias = nullptr;
        }
        rec.kind = ops[k].kind;
        rec.idx = ops[k].idx;
        rec.in.C = ops[k].cin;
        RET_IF(unet_do_op(f, ops[k], cur, o, H, W, tape ? &rec : nullptr, prev, oc));
      }
      prev = oc;
      if (tape) tape->stages.push_back(rec);
      cur = o;
    }
    return 0;
  };

  int H = u.image_size, W = u.image_size;
  View cur;
  cur.p = const_cast<float*>(x_nhwc);
  cur.ld = x_ld;
  cur.C = u.in_channels;
  // input blocks: one block's output goes to the next block only (its other reader, the skip connection, reads the finished
  // tensor much later), so the carry crosses the chain boundaries here; two storages alternate
  Carry xcarry[2];
  {
    size_t emax = 0;
    for (int j = 0; j < nb; ++j)
      if ((long)Bv * in_res[j] * in_res[j] <= 8192) emax = std::max(emax, (size_t)Bv * in_res[j] * in_res[j] * in_ch[j]);
    if (emax && !tape && !no_carry) {
      carry_storage(xcarry[0], emax);
      carry_storage(xcarry[1], emax);
    }
  }
  Carry* prev_carry = nullptr;
  for (int j = 0; j < nb; ++j) {
    const int i = nb - 1 - j;
    View dst;
    dst.p = cat[i] + h_ch[i];
    dst.ld = cat_C[i];
    dst.C = in_ch[j];
    chain_id = j;
    const std::vector<UOp>& next_ops = j + 1 < nb ? c->in_blocks[j + 1] : c->mid_block;
    Carry* outc = (xcarry[j & 1].slabs && !next_ops.empty() && takes_carry(next_ops[0], in_res[j])) ? &xcarry[j & 1] : nullptr;
    if (side_emb && j == 1) HIP_CHECK_RET(hipStreamWaitEvent(s, c->ev_emb, 0));  // the first ResBlock needs the embeddings
    RET_IF(run_chain(c->in_blocks[j], nullptr, cur, dst, H, W, prev_carry, outc, true));
    prev_carry = outc;
    cur = dst;
    if (!forked && (H < u.image_size || j == nb - 1)) RET_IF(fork_ctx());
  }
  {
    View dst;
    dst.p = cat[0];
    dst.ld = cat_C[0];
    chain_id = nb;
    RET_IF(run_chain(c->mid_block, &c->conds[0], cur, dst, H, W, prev_carry, nullptr, false));
  }
  for (int i = 0; i < nb; ++i) {
    View in;
    in.p = cat[i];
    in.ld = cat_C[i];
    in.C = cat_C[i];
    View dst;
    if (i + 1 < nb) {
      dst.p = cat[i + 1];
      dst.ld = cat_C[i + 1];
    } else {
      dst.p = final_h;
      dst.ld = mc;
    }
    const CondW* cond = i >= 3 ? &c->conds[1 + (i - 3)] : nullptr;  // attention.py:100
    chain_id = nb + 1 + i;
    RET_IF(run_chain(c->out_blocks[i], cond, in, dst, H, W, nullptr, nullptr, false));
  }
  // out: GroupNorm32 + SiLU + zero-init conv (openaimodel.py:717-721)
  {
    const int rows = Bv * H * W;
    static const bool no_head32 = getenv("MVD_NO_OUT_CONV_F32") != nullptr || getenv("MVD_GN_TWO_PASS") != nullptr;
    if (!tape && !no_head32 && c->out_conv.w32 && !(W & 15) && gn_group_eligible(mc, H * W, mc, 32, 0, 2 * mc)) {
      // inference: GroupNorm + SiLU written in fp32, the 3 x 3 convolution onto 4 channels in exact fp32 on the vector ALU
      float* a32 = ws_alloc<float>(c, (size_t)rows * mc);
      WS_CHECK(a32);
      RET_IF(run_group_norm(c, final_h, mc, Bv, H * W, c->out_norm, 32, 1e-5f, ACT_SILU, nullptr, (half_t*)a32, 2 * mc, s, 0, 2));
      ProbeScope ps(c, s, "out_conv_f32_kernel", 2.0 * rows * 9.0 * mc * c->out_conv.N, (double)rows * (mc + u.out_channels) * 4.0);
      RET_IF(launch_out_conv_f32(a32, mc, c->out_conv.w32, c->out_conv.bias, c->out_conv.N, Bv, H, W, eps_nhwc, u.out_channels, s));
    } else {
    const int wx = c->out_conv.xp ? 3 : 1;  // extended precision: [hi | lo | hi] operand
    half_t* a = ws_alloc<half_t>(c, (size_t)rows * mc * wx);
    WS_CHECK(a);
    RET_IF(run_group_norm(c, final_h, mc, Bv, H * W, c->out_norm, 32, 1e-5f, ACT_SILU, nullptr, a, mc * wx, s, 0, c->out_conv.xp));
    if (tape) {
      tape->head_a = a;
      tape->head_ld = mc * wx;
    }
    GemmArgs g;
    g.a = a; g.lda = mc * wx; g.w = &c->out_conv; g.out = eps_nhwc; g.ldc = u.out_channels;
    RET_IF(run_conv2d(c, g, Bv, H, W, 1, 0, s));
    }
  }
  static const bool dbg_sum = getenv("MVD_DEBUG_SUM") != nullptr;  // investigation aid: which buffers differ between repeats?
  if (dbg_sum) {
    hipStreamSynchronize(s);
    if (c->side) hipStreamSynchronize(c->side);
    unsigned long long* d = ws_alloc<unsigned long long>(c, 1);
    WS_CHECK(d);
    auto sum = [&](const void* p, size_t bytes) -> unsigned long long {
      unsigned long long h = 0;
      if (!p || launch_bits_checksum(p, bytes, d, s)) return 0;
      hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      return h;
    };
    fprintf(stderr, "[sum]");
    for (auto& b : c->dbg) fprintf(stderr, " %s=%016llx", b.name.c_str(), sum(b.p, b.bytes));
    c->dbg.clear();
    for (int l = 0; l < 4 && src && n_ctx > 0; ++l) {
      const int Dl = depth0 >> l, Sl = u.image_size >> l;
      const size_t n = (size_t)n_ctx * Dl * Sl * Sl * u.volume_dims[l];
      fprintf(stderr, " src%d=%016llx", l, sum(src[l].p, n * (src[l].f32 ? 4 : 2)));
    }
    for (size_t k2 = 0; k2 < c->conds.size() && k2 < 16; ++k2)
      if (f.cn_pre[k2]) fprintf(stderr, " cn%zu=%016llx", k2, sum(f.cn_pre[k2], 64));  // first bytes only: cheap marker
    for (int i = 0; i < nb; ++i)
      fprintf(stderr, " cat%d=%016llx", i, sum(cat[i], (size_t)Bv * in_res[nb - 1 - i] * in_res[nb - 1 - i] * cat_C[i] * 4));
    fprintf(stderr, " final=%016llx eps=%016llx\n", sum(final_h, (size_t)Bv * u.image_size * u.image_size * mc * 4),
            sum(eps_nhwc, (size_t)Bv * u.image_size * u.image_size * u.out_channels * 4));
  }
  return 0;
}

// One block of the UNet on its own input (include/mvd.h: mvd_unet_block): the block-level parity tests of SURVEY section 8
// rows a19-a22 run the production block code -- same plans, same kernels -- against the reference's block goldens.
int engine_unet_block(mvd_ctx* c, const char* path, const float* x_nhwc, int B, int C, int H, int W, const int64_t* t,
                      const float* context, const float* vol_ndhwc, int D, float* out_nhwc, int* Cout, int* Hout, hipStream_t s) {
  if (!c->finalized || !c->has_unet) return mvd_fail("UNet weights not uploaded / finalized");
  const mvd_unet_config& u = c->u;
  int level = 0;
  for (int r = u.image_size; r > H; r >>= 1) ++level;
  if (H != W || level > 3 || (u.image_size >> level) != H) return mvd_fail("mvd_unet_block: the resolution is not a UNet level");
  const std::string p(path);
  const UOp* op = nullptr;
  const CondW* cond = nullptr;
  int cond_idx = -1;
  {
    int i = -1, j = -1;
    if (p == "middle_conditions") cond_idx = 0;
    else if (sscanf(path, "output_conditions.%d", &i) == 1) cond_idx = i >= 0 ? 1 + i : -1;  // ModuleList index (attention.py:98-113)
    else if (sscanf(path, "input_blocks.%d.%d", &i, &j) == 2 && i >= 0 && i < (int)c->in_blocks.size() && j >= 0 &&
             j < (int)c->in_blocks[i].size()) op = &c->in_blocks[i][j];
    else if (sscanf(path, "middle_block.%d", &j) == 1 && j >= 0 && j < (int)c->mid_block.size()) op = &c->mid_block[j];
    else if (sscanf(path, "output_blocks.%d.%d", &i, &j) == 2 && i >= 0 && i < (int)c->out_blocks.size() && j >= 0 &&
             j < (int)c->out_blocks[i].size()) op = &c->out_blocks[i][j];
    if (cond_idx >= 0 && cond_idx < (int)c->conds.size()) cond = &c->conds[cond_idx];
    if (!op && !cond) return mvd_fail("mvd_unet_block: no such block");
  }
  WsScope ws_scope(c);
  Ctx5 src[4];
  Fwd f{c, s, B, cond ? B : 0, cond ? D << level : 0, nullptr, context, nullptr, src, {nullptr, nullptr, nullptr, nullptr}};
  View in, out;
  in.p = const_cast<float*>(x_nhwc); in.ld = C; in.C = C;
  out.p = out_nhwc;
  if (cond) {
    if (C != cond->dim) return mvd_fail("mvd_unet_block: channel count does not match the DepthTransformer");
    if (!vol_ndhwc || D <= 0) return mvd_fail("mvd_unet_block: a DepthTransformer needs its context volume");
    src[level].p = vol_ndhwc;
    src[level].f32 = 1;
    const size_t n = (size_t)B * D * H * W * cond->Cc;
    half_t* h = ws_alloc<half_t>(c, n);  // the forward's fp16 view of the context volume (engine_unet: fork_ctx)
    WS_CHECK(h);
    RET_IF(launch_f32_to_f16(vol_ndhwc, h, n, s));
    f.src16[level] = h;
    out.ld = out.C = cond->dim;
    *Cout = cond->dim;
    *Hout = H;
    return unet_do_cond(f, *cond, in, out, H, W, level, -1);
  }
  if (C != op->cin) return mvd_fail("mvd_unet_block: channel count does not match the block");
  if (op->kind == OP_RES && !t) return mvd_fail("mvd_unet_block: a ResBlock needs the timesteps");
  if (op->kind == OP_ST && !context) return mvd_fail("mvd_unet_block: a SpatialTransformer needs the context");
  float *ea = nullptr, *a2 = nullptr;
  RET_IF(unet_embeddings(c, op->kind == OP_RES ? t : nullptr, op->kind == OP_ST ? context : nullptr, B, s, nullptr, nullptr, nullptr, &ea, &a2));
  f.emb_all = ea;
  f.a2_all = a2;
  out.ld = out.C = op->cout;
  *Cout = op->cout;
  int Ho = H, Wo = W;
  RET_IF(unet_do_op(f, *op, in, out, Ho, Wo, nullptr));
  *Hout = Ho;
  return 0;
}
