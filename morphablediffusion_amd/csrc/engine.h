// Host-side engine: weight store, block plan, workspace, and the executors that turn one denoising step
// into a fixed sequence of kernel launches on one HIP stream (no Python in the loop, no allocation).
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/mvd.h"
#include "common.h"
#include "rowchain.h"

struct RawTensor {
  float* d = nullptr;  // device fp32 copy in reference layout
  std::vector<int64_t> shape;
  size_t numel = 0;
};

struct ConvW {  // packed for the implicit GEMM: fp16 [taps][N][Cin]
  // Extended precision (xp): fp16 MFMA operands carry 11 significand bits, which puts ~4e-4 of relative error on every
  // GEMM output.  For the few layers whose error reaches the UNet output almost undamped (tools/precision_probe4.py: the
  // output conv, conv_in and the cheap layers of the last output block(s) carry half of the end-to-end error variance) both
  // operands are split into fp16 hi + lo parts and three products are accumulated: a_hi w_hi + a_lo w_hi + a_hi w_lo.  This
  // is done by CONCATENATION along K, so every kernel runs unchanged: Cin is the packed width 3 * cin_l, weights hold
  // [w_hi | w_hi | w_lo], the producer of the activation writes [a_hi | a_lo | a_hi] (IGemm::out_split, the GroupNorm /
  // depth-attention kernels' split mode, or launch_rows_f32_to_f16_split for fp32 sources).
  int xp = 0;
  int cin_l = 0;  // logical input width when xp (Cin / 3)
  half_t* w = nullptr;
  half_t* wx = nullptr;    // 3x3 ResBlock convs at 16-divisible resolutions: the weights as a conv3x fragment stream (k_conv3x.hip)
  int wx_bn = 0;           // ... packed for column tiles of this width
  half_t* w_up = nullptr;  // upsample convs only: the 16 parity-folded 2x2 slabs (k_misc.hip: pack_upconv_weight_kernel)
  int res_out = 0;         // upsample convs only: the resolution they produce (decides whether they get a conv3x stream)
  float* w32 = nullptr;    // conv_in / the output conv: the fp32 weights [N][cin_src][3][3] for the exact vector-ALU forms (k_misc.hip)
  int cin_src = 0;         // ... their (unpadded) input width
  float* bias = nullptr;
  int N = 0, Cin = 0, taps = 1;
  // training mode only (engine_train.hip: build_dgrad): the adjoint's weights, fp16 [taps, flipped][cin_l][Np] with Np = N padded
  // to a multiple of 8 -- the same packed layout with the roles of N and Cin swapped, so dgrad runs on the forward kernels
  half_t* wT = nullptr;
  int Np = 0;
  std::string key;   // state_dict key of the weight tensor (gradient bookkeeping); empty for stacked / folded weights
  std::string bkey;  // ... of the bias
};
struct NormW {
  float* g = nullptr;
  float* b = nullptr;
  int C = 0;
  std::string key;  // state_dict prefix ("....norm"): <key>.weight / <key>.bias
};
struct LinW {  // small per-sample linear: fp16 [N][K]
  std::string key;  // state_dict prefix (<key>.weight / <key>.bias); empty for stacked matrices
  half_t* w = nullptr;
  float* bias = nullptr;
  int N = 0, K = 0;
};

struct ResW {
  NormW n1, n2;
  ConvW c1, c2, skip;
  bool has_skip = false;
  int cin = 0, cout = 0, emb_off = 0;
  int res = 0;      // resolution (pixels per side) the block runs at: decides which kernel forms its weights are packed for
  std::string key;  // state_dict prefix (for the extended-precision re-pack)
};
struct STW {
  NormW norm, ln1, ln3;
  ConvW proj_in, qkv, attn_out, ff1, ff2, proj_out;  // qkv: to_q | to_k | to_v rows stacked
  int C = 0, heads = 8, a2_off = 0;
  std::string key;
  // row-chain kernel (k_rowchain.hip): to_out | LayerNorm3-folded FF1 | FF2 | proj_out as one pre-packed fragment stream
  // (null when the width is not one of its instantiated forms)
  half_t* rc_stream = nullptr;
  half_t* rh_stream = nullptr;  // ... and proj_in | LayerNorm1-folded q|k|v for the row-head kernel (C = 320)
  int rc_po = 1, rh_xp = 0;     // forms the streams were packed for: proj_out plain (1) / extended precision (2); proj_in likewise
  // FF2 and proj_out as ONE GEMM (inference, plain-precision proj_out, layered path): both are linear and nothing sits between
  // them, so  proj_out(t2 + FF2(g) + b2) + b_po = [g | t2] [W_po W_2 | W_po]^T + (W_po b2 + b_po)  -- K = 5C instead of two
  // launches with K = 4C and K = C and the [rows][C] intermediate in between.  ffp: [C][5C] fp16 (W_po W_2 folded in fp64 at
  // finalize), ffp.bias the folded bias; null weights = not built (training contexts, extended-precision proj_out).
  ConvW ffp;
};
struct CondW {
  ConvW proj_in, proj_ctx, wqk, wov, conv1, conv2;
  NormW gn_in, gn_ctx, gn_o1, gn_o2;
  half_t* relu_beta = nullptr;  // [4*Cc] z row for an all-zero context (CFG uncond half); [hi | lo | hi] when wov.xp
  int dim = 0, Cc = 0, I = 0;
  int res = 0;  // resolution the block runs at (a conditioner behind an Upsample runs at the new resolution)
  std::string key;
};
// first-stage decoder (AutoencoderKL.decode): ResnetBlock / AttnBlock / Upsample of
// ldm/modules/diffusionmodules/model.py
struct VaeResW {
  NormW n1, n2;
  ConvW c1, c2, skip;
  bool has_skip = false;
  int cin = 0, cout = 0;
};
struct VaeAttnW {  // AttnBlock (model.py:150-202)
  NormW norm;
  ConvW q, k, v, proj;  // v: used as the A operand of the swapped GEMM (V^T = W_v X^T); its bias is folded into
                        // proj.bias (softmax rows sum to 1)
};
struct VaeW {  // Decoder + post_quant_conv
  bool present = false;
  int ch = 0, out_ch = 0, zc = 0, embed = 0, block_in = 0, nlev = 0;
  ConvW post_quant;                // 1x1 conv embed -> zc (Cin padded to 8)
  ConvW conv_in, conv_out;         // conv_in: Cin padded to 8; conv_out: N padded to 4
  NormW norm_out;
  VaeResW mid1, mid2;
  VaeAttnW attn;
  std::vector<std::vector<VaeResW>> up;     // up[level][block], level = index in ch_mult
  std::vector<ConvW> up_conv;               // up_conv[level] (level > 0), parity-folded
};
struct VaeEncW {  // Encoder (double_z) + quant_conv
  bool present = false;
  int in_ch = 0, nlev = 0, mom = 0;         // mom = 2 * embed_dim output channels
  ConvW conv_in, conv_out, quant;           // conv_in: Cin padded to 8
  NormW norm_out;
  VaeResW mid1, mid2;
  VaeAttnW attn;
  std::vector<std::vector<VaeResW>> down;   // down[level][block]
  std::vector<ConvW> down_conv;             // down_conv[level] (level < nlev-1): k3 s2 on the input padded right/bottom
};

// CLIP vision tower (FrozenCLIPImageEmbedder, ldm/modules/encoders/modules.py:343-382; openai/CLIP VisionTransformer)
struct ClipLayerW {
  NormW ln1, ln2;
  ConvW qkv;   // attn.in_proj_weight (q | k | v rows) with the q and k biases; b_v is folded into out.bias
  ConvW out;   // attn.out_proj, bias = b_o + W_o b_v (softmax rows sum to 1)
  ConvW fc;    // mlp.c_fc; QuickGELU(v) = silu(1.702 v) / 1.702: run with alpha = 1.702 and the bias pre-scaled
  ConvW proj;  // mlp.c_proj, run with alpha = 1 / 1.702
};
struct ClipW {
  bool present = false;
  int width = 0, layers = 0, heads = 0, patch = 0, image = 0, embed = 0;
  int T = 0, Tp = 0, Kp = 0;  // tokens, tokens padded to 8, patch-conv K padded to 8
  ConvW conv1;                // [width][Kp], no bias
  float* cls = nullptr;       // class_embedding [width]
  float* pos = nullptr;       // positional_embedding [T][width]
  NormW ln_pre, ln_post;
  LinW proj;                  // proj^T as [embed][width]
  std::vector<ClipLayerW> blk;
};

struct UOp {
  int kind = 0;  // 0 conv_in, 1 res, 2 st, 3 down, 4 up
  int idx = 0;   // index into the per-kind weight vectors
  int cin = 0, cout = 0;
};
enum { OP_CONV_IN = 0, OP_RES = 1, OP_ST = 2, OP_DOWN = 3, OP_UP = 4 };

struct FrustumBlockW {
  LinW t_conv, v_conv;
  NormW gn;
  ConvW conv;
  int cin = 0, cout = 0, stride = 1;
};
struct SparseLayerW {
  std::string wkey, bnkey;  // state_dict keys of the conv weight / of its BatchNorm1d (prefix)
  int layout = 0;           // layout of the uploaded weight (build_sparse_layer): the gradient goes back in the same one
  float* w = nullptr;      // [27][Cin][Cout] fp32
  float* wp = nullptr;     // the same as B fragments of sparse_mfma_kernel (k_cond.hip); null: channel counts it does not take
  float* wd = nullptr;     // training contexts: fragments of the data-gradient's layer (transposed; tap-flipped when submanifold)
  float* scale = nullptr;  // folded eval BatchNorm
  float* shift = nullptr;
  float* gamma = nullptr;  // BatchNorm weight / bias themselves (train mode: batch statistics, engine_volume_from_fused)
  float* beta = nullptr;
  float* rmean = nullptr;  // training contexts: the running_mean / running_var buffers themselves -- a train-mode forward updates
  float* rvar = nullptr;   // them as nn.BatchNorm1d(momentum=0.01) does (network.py:105), the re-pack folds the updated values
  int cin = 0, cout = 0;
  bool strided = false;
};
struct EncBlockW {
  LinW t, v;
  NormW n1, n2;
  ConvW c1, c2;
};

struct Workspace {
  char* base = nullptr;
  size_t size = 0, off = 0, peak = 0;
  // 0: every scope releases its allocations when it ends.
  // 1: none does (work enqueued on the side stream is still using them); the enclosing scope opened before the hold releases
  //    everything.
  // 2: training forward, keep-all: scratch scopes (WS_TEMP: split-K slabs, operand copies) release, everything a block
  //    computes stays for the backward pass.
  // 3: training forward with per-block recompute: block scopes (WS_BLOCK) release too, only the chain level (block inputs /
  //    outputs) stays.
  int hold = 0;
  void* alloc(size_t bytes) {
    size_t o = (off + 255) & ~(size_t)255;
    if (o + bytes > size) return nullptr;
    off = o + bytes;
    if (off > peak) peak = off;
    return base + o;
  }
};

struct MeshTables {
  int Nv = 0;
  float* verts = nullptr;  // device [Nv][3]
  int n_sites[3] = {0, 0, 0};
  int shape[3][3];           // (d,h,w) per level
  int* nbr_subm[3] = {nullptr, nullptr, nullptr};
  int* nbr_down[2] = {nullptr, nullptr};  // level l -> l+1, indexed by output site
  int* grid2 = nullptr;                    // coarse index grid (level 2)
  float min_xyz[3];
  int out_sh[3];
  float* feat[2] = {nullptr, nullptr};  // ping-pong feature buffers [max_sites][64]
  // storage: verts / nbr_* / grid2 point into ONE device pool that is filled by one stream-ordered copy from a pinned host
  // image, so a rebuild (a new mesh in this slot: every training step) neither allocates nor synchronises once the pools have
  // grown to the mesh size; `staged` marks the end of the last copy out of `h_pool`
  int* pool = nullptr;
  int* h_pool = nullptr;
  size_t pool_cap = 0, feat_cap = 0;  // ints / floats per feat buffer
  hipEvent_t staged = nullptr;
};

struct mvd_ctx {
  mvd_unet_config u;
  mvd_volume_config v;
  int device = 0;
  // RCCL communicator of the view-sharded step's one exchange (mvd_comm_init; librccl.so is opened on demand, c_api.hip)
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  hipEvent_t ev_grad_sync[2] = {nullptr, nullptr};  // mvd_train_sync_gradients: main -> comm, comm -> main
  bool grad_sync_started = false;                   // phase 0 ran for the current step (its buckets are already reduced)
  bool finalized = false;
  bool has_unet = false, has_cond = false, has_step = false;
  bool vae_exact = false;   // first-stage encoder / decoder with every conv and the attention in extended precision (mvd_set_vae_precision)
  int precision_level = 3;  // extended-precision policy (engine_weights.hip: apply_xp_policy), mvd_set_precision_level
  bool use_halo = true;  // route eligible 3x3 convs through the LDS-halo kernel (MVD_NO_HALO=1 disables)
  // side stream: the context halves of the DepthTransformers (GroupNorm(proj_context(volume)), ready as soon as the frustum
  // volumes are) run beside the UNet trunk instead of inside it (engine_unet.hip)
  // second helper stream: a ResBlock's 1x1 skip convolution reads only the block input, so it runs beside GroupNorm1 -> conv1 ->
  // GroupNorm2 instead of in front of conv2 (unet_do_res); forked and joined inside the block
  hipStream_t side2 = nullptr;
  hipEvent_t ev_s2_fork = nullptr, ev_s2_join = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr, ev_ctx = nullptr, ev_emb0 = nullptr, ev_emb = nullptr;
  std::vector<hipEvent_t> ev_cond;
  // DepthTransformer on a sample WITHOUT context (the unconditional half of classifier-free guidance: all-zero frustum volumes):
  // GroupNorm(proj_context(0)) = beta, k is constant along the depth axis, the softmax uniform, so depth_attn's output -- and with
  // it everything proj_out computes -- does not depend on x: the block is x + K with an image K [H*W][dim] that depends on the
  // weights and the resolution only (reference ldm/models/diffusion/attention.py:26-47, 78-84).  K is computed once per context
  // (first use, by running the block on one context-free sample) and the step runs the block on the samples WITH context only.
  // The DepthTransformers that attend to the SAME context level (4 at level 0, 3 at level 1, 2 at level 2 with the reference's
  // volume_dims) each normalise their own 1x1x1 projection of that volume: stacked along N the projections are ONE GEMM over the
  // volume (the statistics pass and the apply pass read it once per LEVEL instead of once per block: 100 MB at level 0 and 16
  // views), the GroupNorms one finalize over nblk * 8 groups.  Built with the weights (engine_weights.hip), used by the side-stream
  // context fold of the inference forward (engine_unet.hip: fork_ctx).
  struct CtxGroup {
    int level = 0, Cc = 0, nblk = 0;
    int cond[4] = {0, 0, 0, 0};
    ConvW w;      // stacked proj_context weights [nblk * Cc][Cc]
    NormW gn;     // stacked GroupNorm gain / bias [nblk * Cc]
  };
  std::vector<CtxGroup> ctx_groups;
  float* enc_scratch = nullptr;  // 2-D encoder output + FiLM rows of mvd_vertex_view_features (engine_cond.hip: stream-safe form)
  size_t enc_scratch_cap = 0;    // floats
  struct CondConst {
    float* k = nullptr;
    int H = 0, W = 0;
    bool valid = false;
  };
  std::vector<CondConst> cond_const;
  // in-situ per-kernel-family timing (bench.py's roofline object): HIP events on the launch stream around launches, keyed
  // by the kernel's template instance.  mode 0 off; 1 every launch of every family; 2 only family `probe_only`, a
  // pseudo-random 1-in-`probe_stride` sample of its launches (keeps the timed region undisturbed)
  int probe_mode = 0;
  int probe_stride = 1;
  unsigned probe_counter = 0;
  std::string probe_only;
  struct ProbeFam {
    long launches = 0, sampled = 0;
    double flops = 0.0, bytes = 0.0;      // algorithmic work of ALL launches seen
    double s_flops = 0.0, s_bytes = 0.0;  // ... of the bracketed ones
    std::vector<size_t> ev;               // indices into probe_ev (start event; stop = +1)
  };
  std::map<std::string, ProbeFam> probe_fam;
  std::vector<hipEvent_t> probe_ev;   // pool, two events per bracketed launch
  size_t probe_used = 0;
  std::vector<size_t> probe_null;     // survey mode: brackets around a null kernel (event overhead + dispatch latency)
  std::vector<size_t> probe_empty;    // survey mode: brackets with NOTHING between the two events (what a bracket itself costs)
  std::map<std::string, RawTensor> raw;
  std::vector<void*> owned;  // packed device allocations

  VaeW vae;
  VaeEncW vae_enc;
  ClipW clip;
  // UNet
  LinW te0, te2, emb_all;
  int emb_total = 0;
  ConvW a2_all;  // folded attn2 (W_o W_v, b_o) of every SpatialTransformer stacked: [a2_total][context_dim]
  int a2_total = 0;
  std::vector<std::vector<UOp>> in_blocks, out_blocks;
  std::vector<UOp> mid_block;
  std::vector<ResW> res;
  std::vector<STW> st;
  std::vector<ConvW> convs;  // conv_in / down / up
  std::vector<CondW> conds;  // [0] middle, [1+k] output_conditions.k
  NormW out_norm;
  ConvW out_conv;
  // step embedding MLP of the Lightning module
  LinW step_te0, step_te2;
  // conditioner
  ConvW enc_init, enc_final;
  NormW enc_final_norm;
  EncBlockW enc_blocks[3];
  float* fuse_w = nullptr;
  float* fuse_b = nullptr;
  SparseLayerW sparse[9];
  ConvW fr_conv0;
  FrustumBlockW fr_blocks[6], fr_up[3];
  // FiLM projections (t_conv / v_conv, time_embed / view_embed) of all blocks stacked: one launch each per step
  LinW film_t, film_v, enc_t, enc_v;
  int film_total = 0, film_off[9] = {0};

  // training (engine_train.hip).  With train_mode set before finalize the uploaded fp32 tensors of the hot path
  // (model.diffusion_model.* | spatial_volume.* | time_embed.*) are kept as the MASTER parameters in one flat arena (sorted by
  // key, so the reference's optimiser groups are contiguous ranges); `raw` keeps pointing into it, gradients / Adam moments
  // live in arenas of the same layout, and engine_repack re-derives every packed fp16 weight from the masters in place.
  bool train_mode = false;
  // which optimiser groups received a gradient since the last mvd_train_zero_grad (1: the UNet, 2: time_embed / spatial_volume):
  // mvd_train_adamw_step leaves the others alone, as torch.optim.AdamW skips parameters whose .grad is None
  bool grad_touched[3] = {false, false, false};
  bool repacking = false;
  // Stream of the weight-build launches: 0 for the first build; engine_repack spreads the ~600 small pack / fold launches of
  // a training step's re-pack over four streams (bs rotates per packed tensor, joins between the sections whose packs read
  // earlier packs), they are latency- not bandwidth-bound
  hipStream_t bs = 0;
  hipStream_t bstreams[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t bevents[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t bev_after = nullptr;  // recorded on the caller's stream: what the re-pack follows
  unsigned bs_rr = 0;
  bool bs_multi = false;
  size_t repack_cursor = 0, sec_begin = 0, sec_end = 0;  // the `owned` allocations [sec_begin, sec_end) belong to the re-packable sections
  std::vector<size_t> owned_bytes;
  struct ParamRec {
    std::string key;
    size_t off = 0, numel = 0;
    std::vector<int64_t> shape;
  };
  std::vector<ParamRec> params;
  std::map<std::string, size_t> param_index;
  size_t arena_n = 0;
  float *arena_p = nullptr, *arena_g = nullptr, *arena_m = nullptr, *arena_v = nullptr;
  bool arena_owned[4] = {false, false, false, false};  // hipMalloc'ed here (true) or adopted from the caller (mvd_train_adopt_arena)
  int* found_inf = nullptr;  // device flag of the last gradient finite-check
  // Gradient buckets of the last mvd_train_unet_step, in the order their gradients become final during the backward pass (one per
  // chain of UNet blocks, then one for everything that completes at the end: the stacked embedding / attn2 projections, the
  // head): arena ranges + an event recorded behind the last kernel that writes them.  The caller starts one all-reduce per
  // bucket on its communication stream as soon as the event allows (DDP's bucketed overlap, train_morphable_diffusion.py:302-303).
  struct GradBucket {
    hipEvent_t ev = nullptr;
    std::vector<size_t> off, len;
  };
  std::vector<GradBucket> buckets;
  int n_buckets = 0;
  bool buckets_cached = false;        // the ranges are a property of the loaded weights: built during the first step
  std::vector<char> bucket_done;      // per parameter: already in a bucket (while the ranges are being built)
  float* bucket_snapshot = nullptr;   // test hook: every bucket's ranges are copied here right behind its event

  struct DbgBuf {
    std::string name;
    const void* p;
    size_t bytes;
  };
  std::vector<DbgBuf> dbg;  // MVD_DEBUG_SUM: buffers to checksum at the end of the forward (investigation aid)

  Workspace ws;
  // per-sample, step-invariant tables.  `mesh` / `cams` / `n_cams` are the ACTIVE sample's; mvd_select_sample parks them
  // in `slots[cur_slot]` and activates another slot, so a batch of B samples keeps B sets resident across the steps
  // instead of rebuilding them (hipFree / hipMalloc / host hash maps) for every sample at every step.
  MeshTables mesh;
  ViewCam* cams = nullptr;  // device [n_cams]
  int n_cams = 0;
  struct CamStage {  // pinned host image of `cams` + the end of its last upload (same scheme as MeshTables::h_pool)
    ViewCam* h = nullptr;
    int cap = 0;
    hipEvent_t staged = nullptr;
  };
  CamStage cam_stage;
  struct SampleSlot {
    MeshTables mesh;
    ViewCam* cams = nullptr;
    int n_cams = 0;
    CamStage cam_stage;
    float* volume = nullptr;  // the slot's 32^3 latent volume (mvd_denoise_views_batch reads one per sample)
  };
  std::vector<SampleSlot> slots;
  long bn_train_calls = 0;  // train-mode forwards of the sparse CNN since the weights were loaded (num_batches_tracked)
  int cur_slot = 0;
  float* volume = nullptr;  // device [V][V][V][64] fp32 (channels-last)
  hipEvent_t vol_ready = nullptr;  // caller-owned: recorded after the mvd_volume_from_fused that the next readers need
};

// engine_weights.hip
int engine_finalize(mvd_ctx* c);
// engine_vae.hip: AutoencoderKL.decode on z [B, embed, h, w] (NCHW fp32) -> [B, out_ch, 8h, 8w]
int engine_vae_decode(mvd_ctx* c, const float* z_nchw, int B, int h, int w, float* out_nchw, hipStream_t s);
// AutoencoderKL.encode(x).parameters: x [B, 3, H, W] -> moments [B, 2*embed, H/8, W/8]
int engine_vae_encode(mvd_ctx* c, const float* x_nchw, int B, int H, int W, float* moments_nchw, hipStream_t s);
// engine_clip.hip: FrozenCLIPImageEmbedder.forward on x [B, 3, H, W] in [-1, 1] -> [B, embed]
int engine_clip_encode(mvd_ctx* c, const float* x_nchw, int B, int H, int W, float* out, hipStream_t s);
// engine_unet.hip
struct Ctx5 {  // channels-last context volume of one level for the first n_ctx samples
  const void* p = nullptr;
  int f32 = 1;
};
// ---- training tape (engine_train.hip): what the forward pass leaves behind for the backward pass -------------------------
struct View {
  float* p = nullptr;
  int ld = 0, C = 0;
};
struct ResSaved {  // ResBlock._forward intermediates
  const half_t* a1 = nullptr;  // silu(GN1(x)) [rows][ld1] (first cin columns; [hi | lo | hi] when the conv is extended-precision)
  const half_t* a2 = nullptr;  // silu(GN2(h1))
  const float* h1 = nullptr;   // conv1(a1) + bias + emb: the input of GN2
  int ld1 = 0, ld2 = 0;
};
struct STSaved {  // SpatialTransformer / BasicTransformerBlock intermediates
  const half_t *n0 = nullptr, *l1 = nullptr, *qkv = nullptr, *ao = nullptr, *l3 = nullptr, *gg = nullptr, *t3 = nullptr;
  const float *t0 = nullptr, *t2 = nullptr;
  int ldn0 = 0, ldt3 = 0;
};
enum { OP_COND = 5 };
struct StageRec {
  int kind = 0, idx = 0, chain = 0;  // chain: 0..nb-1 input blocks, nb middle, nb+1+i output block i
  View in, out;
  int H = 0, W = 0, level = 0;       // input resolution
  bool have_saved = false;
  ResSaved rs;
  STSaved ss;
};
struct TrainTape {
  bool recompute = false;  // true: only block inputs / outputs are kept, each block is re-run before its backward
  std::vector<StageRec> stages;  // forward order
  const float *e0 = nullptr, *e1 = nullptr, *e2 = nullptr, *ea = nullptr, *a2 = nullptr, *context = nullptr;
  float* final_h = nullptr;
  const half_t* head_a = nullptr;
  int head_ld = 0;
  std::vector<float*> cat;
  std::vector<int> cat_C, h_ch, in_ch, in_res;
  int Bv = 0, depth0 = 0;
  const struct Ctx5* src = nullptr;
};
// produce (optional): fills src[] by enqueueing the context-volume producer (the frustum network) on the stream it is given;
// engine_unet calls it after its full-resolution input blocks, on the side stream (see engine_unet.hip)
typedef std::function<int(hipStream_t)> CtxProducer;
int engine_unet(mvd_ctx* c, const float* x_nhwc, int x_ld, const int64_t* t, const float* context, int Bv, int n_ctx,
                int depth0, const Ctx5 src[4], float* eps_nhwc, hipStream_t s, const CtxProducer* produce = nullptr,
                TrainTape* tape = nullptr);
// executor state of one UNet forward (engine_unet.hip); the block functions are shared with the backward pass, which re-runs
// single blocks (TrainTape::recompute)
struct Fwd {
  mvd_ctx* c;
  hipStream_t s;
  int Bv, n_ctx, depth0;
  const float* emb_all;  // [Bv][emb_total]
  const float* context;  // [Bv][context_dim]
  const float* a2_all;   // [Bv][a2_total] folded attn2 output per SpatialTransformer
  const Ctx5* src;
  const half_t* src16[4];  // fp16 view of each context level (the source itself or a copy made once per forward)
  // relu(GroupNorm(proj_context(volume))) of every DepthTransformer, produced on the side stream (nullptr: inline)
  const half_t* cn_pre[16] = {nullptr};
  int cn_ld[16] = {0};     // row stride (halfs) of cn_pre[k]: Cc, or the stacked width when the blocks of one level were folded together
  bool ctx_side = false;  // the volumes were produced / converted on the side stream: inline readers wait for ev_ctx
  bool train = false;     // training forward: no deferred split-K slabs, LN1 / LN3 outputs in separate buffers
};
struct Carry;
int unet_embeddings(mvd_ctx* c, const int64_t* t, const float* context, int Bv, hipStream_t s, float** e0_out, float** e1_out,
                    float** e2_out, float** ea_out, float** a2_out);
int engine_unet_block(mvd_ctx* c, const char* path, const float* x_nhwc, int B, int C, int H, int W, const int64_t* t,
                      const float* context, const float* vol_ndhwc, int D, float* out_nhwc, int* Cout, int* Hout, hipStream_t s);
int unet_do_res(Fwd& f, const ResW& r, View in, View out, int H, int W, ResSaved* sv = nullptr, Carry* in_carry = nullptr,
                Carry* out_carry = nullptr);
int unet_do_st(Fwd& f, const STW& t, View in, View out, int H, int W, STSaved* sv = nullptr, Carry* in_carry = nullptr,
               Carry* out_carry = nullptr);
int unet_do_cond(Fwd& f, const CondW& d, View in, View out, int H, int W, int level, int cond_idx = -1);
int unet_do_op(Fwd& f, const UOp& op, View in, View out, int& H, int& W, StageRec* rec = nullptr, Carry* in_carry = nullptr,
               Carry* out_carry = nullptr);
// engine_weights.hip: re-derive the packed weights of the UNet / step embedding / conditioner from the master parameters
int engine_repack(mvd_ctx* c, hipStream_t after = nullptr, bool have_stream = false);
// engine_train.hip
int engine_train_setup(mvd_ctx* c);     // finalize, train mode: masters into the arena
void engine_build_rotate(mvd_ctx* c);  // next build stream (no-op outside a multi-stream re-pack)
int engine_build_join(mvd_ctx* c);     // every build stream waits for all of them
bool engine_hot_key(const std::string& k);  // a key of the re-packable sections (UNet / conditioner / step embedding)
int engine_build_dgrad(mvd_ctx* c);     // adjoint weights of every UNet GEMM (ConvW::wT)
int engine_build_dgrad_cond(mvd_ctx* c);  // ... of the conditioner's dense convolutions
float* engine_grad(mvd_ctx* c, const std::string& key);         // gradient / master of a state_dict entry (nullptr: not a parameter)
const float* engine_master(mvd_ctx* c, const std::string& key);
int engine_dmalloc(mvd_ctx* c, void** p, size_t bytes);         // owned allocation; replays the recorded one while re-packing
// One training step of the UNet (training_step morphable_diffusion.py:520-549 from `self.model(...)` on + loss.backward()):
// forward with the tape, loss = mean((pred - target)^2), dL/dpred * loss_scale back through every block; parameter gradients
// are ACCUMULATED into the gradient arena (x loss_scale).  dsrc[l] (may be null): gradient w.r.t. the context volumes,
// channels-last like src[l].  recompute: keep only block inputs and re-run each block before its backward.
// Backward of the mesh conditioner for ONE sample (its mesh / cameras active): re-runs construct_spatial_volume +
// construct_view_frustum_volume (morphable_diffusion.py:203-320) for the sample's N noisy views and target view with every
// intermediate kept, then back-propagates dsrc[l] = dL/d(frustum volume l) (channels-last) into the parameters of
// spatial_volume.* and time_embed.*.  dbg_*: optional outputs for the parity tests.
int engine_train_conditioner_backward(mvd_ctx* c, const float* x_noisy_nchw, int64_t timestep, const float* v_embed, int n_views,
                                      int target_idx, float* const dsrc[4], float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats,
                                      float* dbg_dtembed, hipStream_t s);
int engine_train_conditioner_backward_batch(mvd_ctx* c, int B, const int* slots, const float* x_noisy_all, const int64_t* timesteps,
                                            const float* v_embed_all, int n_views, const int* target_idx, float* const dsrc[4],
                                            float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, hipStream_t s);
int engine_train_cond_backward(mvd_ctx* c, int cond_idx, const float* x, const float* ctx_vol, const float* dout, int B, int H, int W,
                               int level, int depth0, float* dx, float* dctx, hipStream_t s);
int engine_train_step(mvd_ctx* c, const float* x_nhwc, int x_ld, const int64_t* t, const float* context, int B, int depth0,
                      const Ctx5 src[4], const float* target_nhwc, float loss_scale, int recompute, float* pred_nhwc, float* loss_out,
                      float* const dsrc[4], hipStream_t s);
// engine_cond.hip
int engine_vertex_features(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed,
                           const int32_t* view_idx_dev, int n_local, int add_bias, float* fused_out, hipStream_t s,
                           float* vf_out = nullptr);
// NoisyTargetViewEncoder alone: x_noisy [n_local,4,S,S] -> feats channels-last [n_local*S*S][16]
int engine_target_encoder(mvd_ctx* c, const float* x_noisy, const float* t_embed, const float* v_embed, int n_local, float* feats,
                          hipStream_t s, float* pre_own = nullptr);
bool engine_encoder_is_fused(const mvd_ctx* c);
// the sparse voxel CNN alone: *rows_out = feature rows [n_sites[2]][64] of the coarsest level (mesh ping-pong buffer)
int engine_sparse_net(mvd_ctx* c, const float* fused, hipStream_t s, bool bn_batch_stats, const float** rows_out);
int engine_fuse_vertex_features(mvd_ctx* c, const float* vf_all, int n_views, float* fused_out, hipStream_t s);
// bn_batch_stats: the sparse CNN's BatchNorm1d layers normalise with the statistics of the active rows (the module in train
// mode, as during the reference's training_step) instead of the running buffers
int engine_volume_from_fused(mvd_ctx* c, const float* fused, hipStream_t s, bool bn_batch_stats = false);
struct FrustumOut {
  half_t* lvl0_half = nullptr;  // level 0 in fp16 instead of lvl[0] when engine_frustum(..., half0 = true)
  float* lvl[4];  // channels-last fp32 [TN, D_l, s_l, s_l, C_l]
};
int engine_frustum(mvd_ctx* c, const float* t_embed, const float* v_embed, const int32_t* view_idx_dev, int TN,
                   FrustumOut* out, hipStream_t s, bool half0 = false);
// construct_view_frustum_volume for TN views of EACH of B samples (slots[b]: that sample's cameras and 32^3 volume), the network
// once over the B * TN volumes; t_embed [B][time_dim], v_embed [B][TN][view_dim], view_idx_dev [TN] (the same views per sample)
int engine_frustum_multi(mvd_ctx* c, int B, const int* slots, const float* t_embed, const float* v_embed,
                         const int32_t* view_idx_dev, int TN, FrustumOut* out, hipStream_t s, bool half0);
int engine_frustum_batch(mvd_ctx* c, int B, const int* slots, const float* volumes, const float* t_embed, const float* v_rows,
                         const int32_t* view_idx_dev, FrustumOut* out, hipStream_t s);

// helpers shared by the executors
struct GemmArgs {
  const void* a = nullptr;
  int a_f32 = 0, lda = 0;
  const ConvW* w = nullptr;
  void* out = nullptr;
  int out_f32 = 1, ldc = 0;
  const float* rowbias = nullptr;
  int rb_ld = 0;
  float alpha = 1.0f;  // scale on the accumulator before the biases
  int tap_shift = 0;   // run_conv2d 3x3: taps at offsets {-1,0,1} + tap_shift (1 = zero padding on the right/bottom only,
                       // the first-stage encoder's Downsample: F.pad(x,(0,1,0,1)) + conv k3 s2 p0, model.py:72-76)
  const float* rowscale = nullptr;  // per-sample per-column scale on the accumulator (folded GroupNorm)
  int rs_ld = 0;
  float* gn_partial = nullptr;      // statistics-only pass: per-tile (sum, sumsq) per group, nothing stored
  int gn_cpg = 0;
  const void* resid = nullptr;
  int resid_f32 = 1, ldr = 0;
  int geglu = 0;
  int act = 0;
  bool use_bias = true;
  int force_splitk = 0;
  int out_split = 0;  // fp16 output as [hi | lo | hi] of this logical width (IGemm::out_split)
  // Deferred split-K reduction (run_conv2d): when the plan splits K and the fp32 slabs [sk][rows][N] fit in `slabs`
  // (capacity slabs_cap floats), the GEMM writes them there, NO reduce pass runs and *sk_used = sk; the consumer adds the
  // slabs, the bias and the per-sample bias itself (run_group_norm with nslab > 1).  Otherwise *sk_used = 1 and `out` holds
  // the finished result as usual.
  float* slabs = nullptr;
  size_t slabs_cap = 0;
  int* sk_used = nullptr;
  // with `slabs`: the deferral may also leave the bias and an fp32 residual to the consumer (run_group_norm: bias2 / resid);
  // without this flag a GEMM with a residual always reduces its slabs itself
  bool defer_epilogue = false;
};
// A block's output left as the split-K slabs of its last GEMM: the NEXT block's first GroupNorm sums them, adds the bias and
// the residual, writes the finished tensor to its own input view (for the later readers: residual adds, skip connections)
// and normalises it -- the GEMM's reduce pass and the GroupNorm in one launch (engine_unet.hip: run_chain).
struct Carry {
  float* slabs = nullptr;  // storage provided by the caller; outlives the producing block
  size_t cap = 0;          // floats
  float* aux = nullptr;    // storage for a residual operand that must outlive the producing block (the skip conv's result)
  size_t aux_cap = 0;
  // filled by the producer; sk == 1: nothing was deferred, the output view holds the finished tensor
  int sk = 1;
  size_t stride = 0;       // floats between slabs (rows * N)
  const float* bias = nullptr;
  const float* resid = nullptr;
  int ldr = 0;
};
// plain GEMM / 1x1 conv over `rows` rows grouped in `B` samples (rows % B == 0)
int run_linear(mvd_ctx* c, const GemmArgs& ga, int B, int rows, hipStream_t s);
// 3x3 conv (stride 1/2, optional nearest x2 upsample of the input) on [B,H,W,*] channels-last
int run_conv2d(mvd_ctx* c, const GemmArgs& ga, int B, int H, int W, int stride, int ups, hipStream_t s);
// nearest x2 upsample + 3x3 conv as four parity-folded 2x2 convs on the [B,H,W] input (needs ConvW::w_up)
int run_upconv2d(mvd_ctx* c, const GemmArgs& ga, int B, int H, int W, hipStream_t s);
// 3x3x3 conv stride 1/2 on [B,D,H,W,*]
int run_conv3d(mvd_ctx* c, const GemmArgs& ga, int B, int D, int H, int W, int stride, hipStream_t s);
// ConvTranspose3d(k3,s2,p1,op1): 8 output-parity classes
int run_convT3d(mvd_ctx* c, const GemmArgs& ga, int B, int D, int H, int W, hipStream_t s);
// GroupNorm(+act) -> fp16
// split: out rows are [hi | lo | hi] (3 * n.C halfs), the operand of an extended-precision consumer
// nslab > 1 (single-pass form only, nslab <= 4): x is nslab split-K slabs `slab_stride` floats apart whose sum, plus bias2[c]
// and the pre-add, is the tensor to normalise (GemmArgs::slabs)
int run_group_norm(mvd_ctx* c, const float* x, int ld, int B, int rows_per_sample, const NormW& n, int groups, float eps,
                   int act, const float* preadd, half_t* out, int ldo, hipStream_t s, int preadd_ld = 0, int split = 0,
                   int nslab = 1, size_t slab_stride = 0, const float* bias2 = nullptr, const float* resid = nullptr, int ldr = 0,
                   float* mat = nullptr, int ldm = 0);
// restores the workspace bump pointer when the scope is left, on the error returns too (a failed call must not leak
// workspace into the calls that follow it)
enum { WS_CHAIN = 0, WS_BLOCK = 1, WS_TEMP = 2 };
struct WsScope {
  Workspace& w;
  const size_t mark;
  const int kind;
  explicit WsScope(mvd_ctx* c, int kind_ = WS_CHAIN) : w(c->ws), mark(c->ws.off), kind(kind_) {}
  ~WsScope() {
    if (!w.hold || (kind == WS_TEMP && w.hold >= 2) || (kind == WS_BLOCK && w.hold == 3)) w.off = mark;
  }
  WsScope(const WsScope&) = delete;
  WsScope& operator=(const WsScope&) = delete;
};
// Joins the side stream back into the caller's stream when the scope is left (on the error returns too): nothing enqueued
// after that point can overtake side-stream work that still reads workspace memory.  Declare it AFTER the WsScope whose
// memory the side stream uses.
struct SideJoin {
  hipStream_t main;
  hipStream_t side = nullptr;
  hipEvent_t ev = nullptr;
  explicit SideJoin(hipStream_t m) : main(m) {}
  ~SideJoin() {
    if (!side) return;
    hipEventRecord(ev, side);
    hipStreamWaitEvent(main, ev, 0);
  }
};
// Brackets the launches enqueued on `s` during its lifetime with two HIP events and books them under `family` (see
// mvd_ctx::probe_*).  flops / bytes: the ALGORITHMIC work of the bracketed launch (each operand and the result once).
struct ProbeScope {
  mvd_ctx* c;
  hipStream_t s;
  size_t slot = (size_t)-1;
  ProbeScope(mvd_ctx* c_, hipStream_t s_, const char* family, double flops, double bytes) : c(c_), s(s_) {
    if (!c->probe_mode) return;
    if (c->probe_mode == 2 && c->probe_only != family) return;
    mvd_ctx::ProbeFam& f = c->probe_fam[family];
    ++f.launches;
    f.flops += flops;
    f.bytes += bytes;
    if (c->probe_mode == 2 && c->probe_stride > 1) {
      unsigned h = ++c->probe_counter * 2654435761u;  // deterministic, unbiased 1-in-stride sample
      h ^= h >> 15;
      if (h % (unsigned)c->probe_stride) return;
    }
    if (c->probe_used + 2 > c->probe_ev.size())
      for (int i = 0; i < 2; ++i) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        c->probe_ev.push_back(ev);
      }
    slot = c->probe_used;
    c->probe_used += 2;
    ++f.sampled;
    f.s_flops += flops;
    f.s_bytes += bytes;
    f.ev.push_back(slot);
    hipEventRecord(c->probe_ev[slot], s);
  }
  ~ProbeScope() {
    if (slot == (size_t)-1) return;
    hipEventRecord(c->probe_ev[slot + 1], s);
    // survey mode: after every 8th bracketed launch an EMPTY bracket (two events, nothing between) on the same stream -- its
    // elapsed time is what the bracket adds to a launch's own duration (the second event's barrier packet); reported as the
    // pseudo-family "(empty bracket)" so that the caller can subtract it
    if (c->probe_mode == 1 && (slot & 14) == 0) {
      if (c->probe_used + 2 > c->probe_ev.size())
        for (int i = 0; i < 2; ++i) {
          hipEvent_t ev;
          if (hipEventCreate(&ev) != hipSuccess) return;
          c->probe_ev.push_back(ev);
        }
      const size_t e = c->probe_used;
      c->probe_used += 2;
      c->probe_empty.push_back(e);
      hipEventRecord(c->probe_ev[e], s);
      hipEventRecord(c->probe_ev[e + 1], s);
    }
    // ... and, offset by four brackets, one around a NULL kernel (one wave, no memory access): event overhead PLUS the dispatch
    // latency every bracketed launch pays between the first event's completion and its own first wave -- which rocprofv3's kernel
    // durations (begin to end of the kernel itself) do not contain.  Round 5 subtracted the empty bracket only and its
    // dominant-kernel time sat 16 % above rocprofv3's; pseudo-family "(null-kernel bracket)".
    if (c->probe_mode == 1 && (slot & 14) == 8) {
      if (c->probe_used + 2 > c->probe_ev.size())
        for (int i = 0; i < 2; ++i) {
          hipEvent_t ev;
          if (hipEventCreate(&ev) != hipSuccess) return;
          c->probe_ev.push_back(ev);
        }
      const size_t e = c->probe_used;
      c->probe_used += 2;
      c->probe_null.push_back(e);
      hipEventRecord(c->probe_ev[e], s);
      launch_probe_null(s);
      hipEventRecord(c->probe_ev[e + 1], s);
    }
  }
  ProbeScope(const ProbeScope&) = delete;
  ProbeScope& operator=(const ProbeScope&) = delete;
};
// creates the side stream and its events on first use
int engine_side_init(mvd_ctx* c);
template <typename T>
inline T* ws_alloc(mvd_ctx* c, size_t n) {
  return (T*)c->ws.alloc(n * sizeof(T));
}
#define WS_CHECK(p) \
  if (!(p)) return mvd_fail("workspace exhausted: create the context with a larger workspace_bytes")
#define RET_IF(x)          \
  do {                     \
    int _r = (x);          \
    if (_r) return _r;     \
  } while (0)
