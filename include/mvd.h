/* libmvd_hip.so -- C ABI of the MI355X-native multi-view denoiser (Morphable Diffusion hot path).
 *
 * The reference has no FFI: its plug-in surface is Python (YAML ``target:`` classes + method signatures,
 * SURVEY.md section 8(b)).  This header is the boundary a binding for that surface talks to; each entry
 * point names the reference interface it replaces.  Plain pointers and sizes only, no torch types.
 * Device pointers are HIP device memory on the context's device; ``stream`` is a hipStream_t (NULL = the
 * default stream).  Tensors at the boundary use the reference's own layouts (NCHW / NCDHW, fp32).
 * All work of a call is ordered on ``stream``: mvd_denoise_views / mvd_unet_forward overlap part of it on one internal side
 * stream per context, forked from and joined back into ``stream`` with events inside the call (MVD_NO_SIDE_STREAM=1 in the
 * environment disables that).
 * Every function returns 0 on success, non-zero on failure; mvd_last_error() gives the text.
 * A context is thread-compatible (one thread at a time per context).  No allocation happens on the step
 * path: weights and the workspace are allocated at create / finalize / set_mesh time.
 */
#ifndef MVD_H
#define MVD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvd_ctx mvd_ctx;

/* kwargs of ldm.models.diffusion.attention.DepthWiseAttention (reference configs/facescape.yaml:28-42,
 * ctor ldm/modules/diffusionmodules/openaimodel.py:444-727, ldm/models/diffusion/attention.py:87-115) */
typedef struct {
  int image_size;       /* latent resolution (32) */
  int in_channels;      /* 8 */
  int out_channels;     /* 4 */
  int model_channels;   /* 320 */
  int num_res_blocks;   /* 2 */
  int channel_mult[4];  /* 1,2,4,4 */
  int num_heads;        /* 8 */
  int context_dim;      /* 768 */
  int volume_dims[4];   /* 64,128,256,512 */
  int attention_levels; /* bit i set: attention at downsample rate 2^i (attention_resolutions [4,2,1] -> 0x7) */
} mvd_unet_config;

/* SpatialVolumeNet ctor (ldm/models/diffusion/morphable_diffusion.py:152-180) */
typedef struct {
  int time_dim;             /* 256 */
  int view_dim;             /* 4 */
  int num_views;            /* N: SMPLFeatureExtractor.num_views (hard-coded 16 in the reference) */
  int input_image_size;     /* 256 */
  int frustum_volume_depth; /* 48 */
  int spatial_volume_size;  /* 32 */
  float spatial_volume_length; /* 0.5 */
  float frustum_volume_length; /* 0.86603 */
  int projection;           /* 0 perspective, 1 orthographic */
  int frustum_dims[4];      /* 64,128,256,512 */
  float voxel_size;         /* 0.005 */
} mvd_volume_config;

int mvd_create(const mvd_unet_config* ucfg, const mvd_volume_config* vcfg, int device, size_t workspace_bytes,
               mvd_ctx** out);
void mvd_destroy(mvd_ctx* ctx);
const char* mvd_last_error(void);
/* "f16" (libmvd_hip.so) or "bf16" (libmvd_hip_bf16.so: the same sources built with bfloat16 MFMA operands and storage, the
 * training dtype BASELINE configs[3] names; the reference trainer's `precision` key, configs/facescape.yaml:65). */
const char* mvd_compute_dtype(void);

/* Replaces load_state_dict (reference generate_face.py:75-76): one call per state_dict entry, keyed by the
 * reference's own names.  data is fp32, contiguous, reference layout; on_device
 * selects device or host memory.  Unknown keys (VAE, CLIP, schedule buffers) are ignored and return 0. */
int mvd_upload_weight(mvd_ctx* ctx, const char* name, const float* data, const int64_t* shape, int ndim, int on_device);
/* Numerical policy, to be set before mvd_finalize_weights.  MFMA operands are fp16 (11 significand bits: ~4e-4 relative
 * error per GEMM output); the layers whose error reaches the UNet output almost undamped run in EXTENDED precision: both
 * operands split into fp16 hi + lo parts, three products accumulated in fp32 (a_hi w_hi + a_lo w_hi + a_hi w_lo).
 * level 0: none.  1: output conv + conv_in.  2: + the cheap layers (skip conv, transformer proj_in / proj_out, DepthTransformer)
 * of the last output block.  3 (default since round 6): + the same layers of the block before it.  4: + the last block's ResBlock
 * 3x3 convs.  5: all such layers of every full-resolution output block.  6: + of the full-resolution input blocks.  The ladder is
 * ordered by error removed per unit of time (until round 5 levels 3 and 4 added these two steps in the other order and the
 * default was 2).  Measured guided-eps error of the full step against the fp32 reference and step time (DESIGN.md section 2,
 * profiles/r06_k_xp_levels.txt): level 2 7.3-8.4e-4, level 4 6.0-7.1e-4 at +3.6 %; level 3 sits between at +1.2 %. */
int mvd_set_precision_level(mvd_ctx* ctx, int level);
/* First-stage model (mvd_vae_decode / mvd_vae_encode), to be set before mvd_finalize_weights.  0 (default): fp16 MFMA operands
 * like the UNet -- ~30 convolutions in series leave 1.8-2.0e-3 relative L2 in the decoded image (at most 0.8 of an 8-bit step).
 * 1 ("exact"): every convolution, the attention projections and both attention products run in extended precision (fp16 hi + lo
 * operand split, three products accumulated in fp32): the reference's fp32 result to ~1e-5, at about three times the cost. */
int mvd_set_vae_precision(mvd_ctx* ctx, int exact);
/* Packs/folds the uploaded tensors into their MFMA layouts; fails (listing the key) if one is missing. */
int mvd_finalize_weights(mvd_ctx* ctx);

/* DepthWiseAttention.forward(x, timesteps, context, source_dict) -- ldm/models/diffusion/attention.py:117-138.
 * x [Bv,in_channels,s,s]; timesteps [Bv] int64; context [Bv,1,context_dim]; src{32,16,8,4}: the source_dict
 * volumes [n_ctx,C_l,D_l,s_l,s_l] fp32 for the FIRST n_ctx samples, the remaining Bv-n_ctx samples are
 * treated as all-zero volumes (the classifier-free-guidance half, morphable_diffusion.py:137-139);
 * depth0 = D of the finest level (48); out [Bv,out_channels,s,s]. */
int mvd_unet_forward(mvd_ctx* ctx, const float* x, const int64_t* timesteps, const float* context, int Bv, int n_ctx,
                     const float* src0, const float* src1, const float* src2, const float* src3, int depth0, float* out,
                     void* stream);
/* ONE block of DepthWiseAttention on its own input, through the production block code (same plans and kernels as
 * mvd_unet_forward): `path` is the reference module path below model.diffusion_model --
 *   "input_blocks.I.J" / "middle_block.J" / "output_blocks.I.J": ResBlock._forward (openaimodel.py:256-276; needs timesteps [B]),
 *      SpatialTransformer.forward (modules/attention.py:325-336; needs context [B,1,context_dim]), Downsample / Upsample /
 *      the input convolution (openaimodel.py:100-157);
 *   "middle_conditions" / "output_conditions.K": DepthTransformer.forward (ldm/models/diffusion/attention.py:78-84; needs
 *      volume [B,C_l,D,H,W], every sample conditional).
 * x [B,C,H,W] at a UNet resolution (image_size >> level); out receives [B,Cout,Ho,Wo] (at most out_capacity floats) and
 * out_shape[4] its shape.  Used by the block-level parity tests. */
int mvd_unet_block(mvd_ctx* ctx, const char* path, const float* x, int B, int C, int H, int W, const int64_t* timesteps,
                   const float* context, const float* volume, int D, float* out, int out_capacity, int* out_shape, void* stream);


/* SyncMultiviewDiffusion.embed_time -- morphable_diffusion.py:491-494. t [B] int64 -> out [B,time_dim] */
int mvd_embed_time(mvd_ctx* ctx, const int64_t* t, int B, float* out, void* stream);

/* Step-invariant per-sample metadata (batch dict, generate_face.py:227-241), HOST pointers:
 * vertices [Nv,3] fp32 (+-0.5 cube frame), coord [Nv,3] int32 (z,y,x), out_sh [3] int32, bounds [2,3] fp32.
 * Builds the sparse-convolution neighbour tables (replaces spconv's indice generation, morphable_diffusion.py:245-254).
 * Transactional: on failure the previously set mesh stays active and intact. */
int mvd_set_mesh(mvd_ctx* ctx, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds,
                 int Nv);
/* A context keeps up to 64 independent sets of per-sample tables (mesh + cameras): mvd_select_sample makes set `slot` the
 * active one -- mvd_set_mesh / mvd_set_cameras write it, the stage calls read it.  A batch of B samples (the reference's
 * ``for bi in range(B)`` loop, morphable_diffusion.py:245) uploads each sample's tables once and switches between them at
 * every step instead of rebuilding them.  Slot 0 is active after mvd_create. */
int mvd_select_sample(mvd_ctx* ctx, int slot);
/* target_K [N,4,4], target_RT [N,3,4] fp32 HOST pointers (batch['target_K'], batch['target_RT']) */
int mvd_set_cameras(mvd_ctx* ctx, const float* K, const float* RT, int N);
/* The same two uploads in the order of `stream`, for callers whose mesh changes at every step (a training step sees a new
 * batch, generate_face.py one mesh per trajectory): the host pointers are read before the call returns, the tables go to the
 * device as ONE copy enqueued on `stream` -- launches enqueued on `stream` earlier keep reading the previous tables, later
 * ones read the new tables; nothing is allocated and nothing synchronises once the slot's pools have grown to the mesh size.
 * All launches of this context that read the slot must be on `stream` (or ordered after it by the caller). */
int mvd_set_mesh_async(mvd_ctx* ctx, const float* vertices, const int32_t* coord, const int32_t* out_sh, const float* bounds,
                       int Nv, void* stream);
int mvd_set_cameras_async(mvd_ctx* ctx, const float* K, const float* RT, int N, void* stream);
/* A whole batch at once (the reference's ``for bi in range(B)`` loop over a NEW batch, morphable_diffusion.py:245-254: spconv
 * regenerates its indices for every sample at every call): sample i's mesh and cameras go to slot slots[i] (distinct). The B
 * rule books are built on B host threads, every sample is validated before the first table is replaced, the uploads are
 * enqueued on `stream` as in the two calls above.  The active slot is unchanged on return. */
int mvd_set_samples_async(mvd_ctx* ctx, int B, const int* slots, const float* const* vertices, const int32_t* const* coord,
                          const int32_t* const* out_sh, const float* const* bounds, const int* Nv, const float* const* K,
                          const float* const* RT, int N, void* stream);
/* Host-only view of the rule book (no context, no GPU: used by the CPU tests): builds the tables of one mesh on the calling
 * thread -- n_sites[3] active sites per level, lens[6] = ints in nbr_subm[0..2] ([n_sites][27], -1 = inactive), nbr_down[0..1]
 * ([n_sites of the next level][27]) and the coarse index grid -- and mvd_rulebook_table copies table `which` (0..5) of that
 * build.  force_hash != 0 takes the open-addressing path that grids of more than 2^25 cells use. */
int mvd_rulebook_build(const int32_t* coord, const int32_t* out_sh, int Nv, int force_hash, int32_t* n_sites, int64_t* lens);
int mvd_rulebook_table(int which, int32_t* out);

/* First half of SpatialVolumeNet.construct_spatial_volume (morphable_diffusion.py:203-231) for the views
 * view_idx[0..n_local): x_noisy [n_local,4,s,s], t_embed [time_dim], v_embed [n_local,view_dim];
 * fused_out [Nv,16] = this rank's share of conv0(mean over ALL num_views views); with add_bias != 0 the
 * conv bias is included (exactly one rank adds it).  Summing fused_out over ranks gives the full tensor. */
int mvd_vertex_features(mvd_ctx* ctx, const float* x_noisy, const float* t_embed, const float* v_embed,
                        const int32_t* view_idx, int n_local, int add_bias, float* fused_out, void* stream);
/* The same stage split for the exact view exchange ("equivalently an all-gather of [N,16,Nv] if bit-exact
 * view-order summation is wanted"): mvd_vertex_view_features writes the per-view features vf_out [n_local,Nv,16]
 * (morphable_diffusion.py:203-229) without reducing them; after the ranks' slices are gathered in view order,
 * mvd_fuse_vertex_features applies SMPLFeatureExtractor (network.py:41-72) to vf_all [num_views,Nv,16] -> fused_out
 * [Nv,16].  Summation runs over the views in index order, so a sharded step is bit-identical to the single-GPU one. */
int mvd_vertex_view_features(mvd_ctx* ctx, const float* x_noisy, const float* t_embed, const float* v_embed,
                             const int32_t* view_idx, int n_local, float* vf_out, void* stream);
int mvd_fuse_vertex_features(mvd_ctx* ctx, const float* vf_all, int n_views, float* fused_out, void* stream);
/* 1 when mvd_vertex_view_features touches no memory of the context's shared workspace (the 2-D encoder runs as its one-launch
 * form out of a scratch of its own): the call may then be enqueued on the caller's communication stream WHILE mvd_denoise_views of
 * the same context runs on another stream -- the sampler moves the step's head (encoder, vertex gather, exchange, sparse CNN) off
 * the UNet's stream entirely.  0: keep it on the stream the UNet is launched on (64 x 64 latents, MVD_NO_FUSED_ENC). */
int mvd_vertex_features_stream_safe(mvd_ctx* ctx);
/* Stage probes for the parity tests (each stage of the conditioner on its own, in the reference's layouts):
 *   mvd_stage_target_encoder   NoisyTargetViewEncoder.forward (ldm/models/diffusion/network.py:181-207) for n_local views of one
 *                              sample: x_noisy [n_local,4,s,s], t_embed [time_dim], v_embed [n_local,view_dim] -> feats [n_local,16,s,s];
 *   mvd_stage_sparse_dense     SparseConvNet.forward(...) (network.py:74-96, the xyzc_net call of morphable_diffusion.py:253-254):
 *                              fused [Nv,16] -> dense [C,d,h,w] of the coarsest level (spconv's .dense(), batch 1); train_mode != 0
 *                              uses batch statistics in the BatchNorm1d layers.  shape_out (may be NULL) receives {C,d,h,w};
 *                              dense_out == NULL only queries the shape. */
int mvd_stage_target_encoder(mvd_ctx* ctx, const float* x_noisy, const float* t_embed, const float* v_embed, int n_local,
                             float* feats, void* stream);
int mvd_stage_sparse_dense(mvd_ctx* ctx, const float* fused, int train_mode, float* dense_out, int32_t* shape_out, void* stream);
/* The view-sharded step's ONE collective behind the C ABI (replaces nothing in the reference, whose sampler
 * is single-GPU: ldm/models/diffusion/morphable_diffusion.py:701-739).  Rank g owns views [g N/G, (g+1) N/G); every step each
 * rank computes the per-view vertex features of its views (mvd_vertex_view_features) and ALL-GATHERS them, so that every rank
 * sums the N views in index order (mvd_fuse_vertex_features): bit-identical to the single-GPU step.
 *   mvd_comm_unique_id : one rank creates the RCCL id (128 bytes) and hands it to the others over any side channel
 *   mvd_comm_init      : collective; creates the communicator this context owns (librccl.so is opened on first use)
 *   mvd_exchange_view_features : ncclAllGather of [n_local][Nv][16] fp32 from `local` into `all` ([N][Nv][16], rank r's slice
 *                        at views [r n_local, (r+1) n_local)) on `stream` -- no Python and no torch.distributed on the step path
 *   mvd_comm_destroy   : also done by mvd_destroy */
typedef struct { char internal[128]; } mvd_rccl_id;
int mvd_comm_unique_id(mvd_rccl_id* id_out);
int mvd_comm_init(mvd_ctx* ctx, const mvd_rccl_id* id, int rank, int world);
int mvd_comm_destroy(mvd_ctx* ctx);
int mvd_exchange_view_features(mvd_ctx* ctx, const float* local, float* all, int n_local, void* stream);
/* In-place sum all-reduce of `count` device floats over the context's communicator on `stream` (ncclAllReduce): the
 * "all_reduce" form of the step's exchange (321 KB of per-vertex sums) and the building block of the gradient
 * averaging below. */
int mvd_comm_all_reduce(mvd_ctx* ctx, float* buf, size_t count, void* stream);
/* Cross-stream hand-over of the volume.  The exchange + mvd_fuse_vertex_features + mvd_volume_from_fused may run on a
 * communication stream of the caller while `stream` of mvd_denoise_views already executes the UNet's input blocks (which
 * need none of it): record a hipEvent_t after mvd_volume_from_fused on that stream and register it here; every later reader
 * of the volume inside the library (the frustum stage of mvd_denoise_views / mvd_frustum_volumes) makes ITS stream wait for
 * the event before the first read.  The event is caller-owned and must stay alive while registered; NULL unregisters. */
int mvd_set_volume_ready_event(mvd_ctx* ctx, void* event);
/* Second half (morphable_diffusion.py:232-257): sparse voxel CNN + lattice gather. fused [Nv,16] ->
 * volume_out [64,V,V,V] (reference layout, may be NULL); the result is also kept in the context for
 * mvd_frustum_volumes / mvd_denoise_views. */
int mvd_volume_from_fused(mvd_ctx* ctx, const float* fused, float* volume_out, void* stream);

/* Training-step variants (reference training_step morphable_diffusion.py:520-549).  The Lightning module
 * is in train mode there, so the sparse CNN's BatchNorm1d layers (network.py:105) normalise with the statistics of the
 * sample's active rows instead of their running buffers: mvd_volume_from_fused_train is mvd_volume_from_fused with that
 * behaviour.  mvd_mse_loss: out[0] = mean((a - b)^2) over n device floats (the "loss_simple" of :541-542). */
int mvd_volume_from_fused_train(mvd_ctx* ctx, const float* fused, float* volume_out, void* stream);
int mvd_mse_loss(mvd_ctx* ctx, const float* a, const float* b, size_t n, float* out, void* stream);
/* Training step of the UNet (reference training_step morphable_diffusion.py:520-549 from `self.model(...)`
 * on, loss.backward(), configure_optimizers :627-646).
 *   mvd_train_enable(ctx, 1)        BEFORE mvd_finalize_weights: the uploaded fp32 tensors of model.diffusion_model.* /
 *                                   spatial_volume.* / time_embed.* are kept as master parameters in ONE flat arena (sorted by
 *                                   key, each tensor aligned to 64 floats); gradients and Adam moments use the same layout.
 *   mvd_train_param_count / _info   enumerate the parameters: name (state_dict key), offset and numel in the arena (floats),
 *                                   shape (up to 8 dims) -- what nn.Module.named_parameters() is to the reference.
 *   mvd_train_arena_size            floats per arena.
 *   mvd_train_adopt_arena(which, ptr, numel)   moves arena `which` (0 parameters, 1 gradients, 2 / 3 Adam moments) into
 *                                   caller-owned device memory of the same size (its content is copied over, a not yet
 *                                   existing arena is zeroed): the caller's tensor framework then sees parameters / gradients
 *                                   as views of ONE buffer (one RCCL all-reduce for the DDP gradient averaging of
 *                                   train_morphable_diffusion.py:302-303).  The memory must outlive the context.
 *   mvd_train_zero_grad             optimizer.zero_grad().
 *   mvd_train_unet_step             x [B,in_channels,s,s], timesteps [B], context [B,1,context_dim], src{0..3} the
 *                                   source_dict volumes [B,C_l,D_l,s_l,s_l] (after the condition dropout), target [B,out,s,s]:
 *                                   pred = UNet(x, ...), loss = mean((target - pred)^2) (morphable_diffusion.py:541-542), then
 *                                   dL/dpred * loss_scale is back-propagated through every block; parameter gradients are
 *                                   ACCUMULATED (x loss_scale) into the gradient arena.  dsrc{0..3} (each may be NULL) receive
 *                                   the gradient w.r.t. the volumes (x loss_scale).  recompute != 0: activation checkpointing
 *                                   per block (ldm/modules/diffusionmodules/util.py:102-148 -- only block inputs are kept, a
 *                                   block is re-run before its backward); 0 keeps every intermediate of the forward pass.
 *                                   fp16 MFMA operands, fp32 accumulation / master weights; pick loss_scale so that the scaled
 *                                   gradients stay inside the fp16 range (mvd_train_adamw_step reports overflow).
 *   mvd_train_get_grad              copy of one parameter's gradient (PyTorch layout), still multiplied by the loss scale(s).
 *   mvd_train_adamw_step            torch.optim.AdamW on the arena: the UNet group (all of model.diffusion_model.* when
 *                                   finetune_unet != 0, else the DepthTransformers of attention.py:140-142) at `lr`,
 *                                   time_embed.* and spatial_volume.* at `lr_aux` (the reference: 10 lr); gradients are
 *                                   multiplied by inv_scale first.  `step` counts from 1.  If any gradient is inf / nan the
 *                                   whole update is skipped (*skipped_out = 1; reading it back synchronises the stream; NULL
 *                                   = no read-back).  A group no backward call has written to since the last
 *                                   mvd_train_zero_grad is left alone -- parameters AND moments -- as torch.optim.AdamW skips
 *                                   parameters whose .grad is None (e.g. spatial_volume.* when only mvd_train_unet_step ran).
 *                                   With inv_scale == 1 (no loss scaling: the bfloat16 build) the overflow
 *                                   check is not run -- torch.optim.AdamW's own behaviour -- and *skipped_out stays 0.
 *   mvd_train_repack                after the parameters changed: re-derive every packed fp16 weight in place. */
int mvd_train_enable(mvd_ctx* ctx, int on);
int mvd_train_param_count(mvd_ctx* ctx);
int mvd_train_param_info(mvd_ctx* ctx, int index, char* name, size_t name_cap, int64_t* offset, int64_t* numel, int64_t* shape,
                         int* ndim);
int64_t mvd_train_arena_size(mvd_ctx* ctx);
int mvd_train_adopt_arena(mvd_ctx* ctx, int which, float* ptr, int64_t numel);
int mvd_train_zero_grad(mvd_ctx* ctx, void* stream);
int mvd_train_unet_step(mvd_ctx* ctx, const float* x, const int64_t* timesteps, const float* context, int B, const float* src0,
                        const float* src1, const float* src2, const float* src3, int depth0, const float* target, float loss_scale,
                        int recompute, float* pred_out, float* loss_out, float* dsrc0, float* dsrc1, float* dsrc2, float* dsrc3,
                        void* stream);
/* The conditioner's backward for ONE sample (its mesh / cameras active: mvd_select_sample): re-runs construct_spatial_volume and
 * construct_view_frustum_volume (morphable_diffusion.py:203-320: step embedding, 2-D encoder on the N noisy views, vertex gather,
 * view fusion, sparse voxel CNN in train mode, lattice gather, frustum gather and FrustumTV3DNet for the target view) with every
 * intermediate kept, then back-propagates dsrc{0..3} = dL/d(frustum volume l) [1,C_l,D_l,s_l,s_l] (what mvd_train_unet_step
 * returns for that sample, after the condition-dropout mask) into the gradients of spatial_volume.* and time_embed.*
 * (accumulated into the arena).  x_noisy [N,4,s,s], v_embed [N,view_dim] device pointers; timestep / target_index host values.
 * dbg_* (each may be NULL): dL/d(32^3 volume) [64,V,V,V], dL/d(fused vertex features) [Nv,16], dL/d(2-D encoder output)
 * [N,16,s,s], dL/d(step embedding) [time_dim] -- for the parity tests.  The three gather adjoints use fp32 atomic adds. */
int mvd_train_conditioner_backward(mvd_ctx* ctx, const float* x_noisy, int64_t timestep, const float* v_embed, int n_views,
                                   int target_index, const float* dsrc0, const float* dsrc1, const float* dsrc2, const float* dsrc3,
                                   float* dbg_dvolume, float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, void* stream);
/* The same for the B samples whose tables sit in slots[0..B) (distinct; uploaded by mvd_set_samples_async): the per-sample
 * stages run sample by sample, FrustumTV3DNet -- shared weights, same-shaped volumes -- runs once with the samples as its batch.
 * x_noisy [B,N,4,s,s], v_embed [B,N,view_dim], dsrc{l} [B,C_l,D_l,s_l,s_l] device pointers; timesteps [B] / target_index [B] HOST
 * arrays; dbg_* only with B = 1.  The active slot is unchanged on return. */
int mvd_train_conditioner_backward_batch(mvd_ctx* ctx, int B, const int* slots, const float* x_noisy, const int64_t* timesteps,
                                         const float* v_embed, int n_views, const int* target_index, const float* dsrc0,
                                         const float* dsrc1, const float* dsrc2, const float* dsrc3, float* dbg_dvolume,
                                         float* dbg_dfused, float* dbg_dfeats, float* dbg_dtembed, void* stream);
/* Parity hook: the backward pass of ONE DepthTransformer (attention.py:49-84; cond_index 0 = middle_conditions, 1 + k =
 * output_conditions.k) given its input x [B,dim,H,W], its context volume [B,C_l,D_l,H,W] and dL/d(output) [B,dim,H,W]:
 * dx, dcontext (may be NULL) are written, parameter gradients accumulated into the arena.  depth0 = D of the finest level. */
int mvd_train_cond_backward(mvd_ctx* ctx, int cond_index, const float* x, const float* context, const float* d_out, int B, int H,
                            int W, int depth0, float* dx, float* dcontext, void* stream);
int mvd_train_get_grad(mvd_ctx* ctx, const char* name, float* out, size_t numel, void* stream);
/* Checkpoint export of a training context (what Lightning's ModelCheckpoint gets from ``state_dict()``): the current value of a
 * resident tensor by its state_dict key -- a master parameter, or a BatchNorm running_mean / running_var buffer of the sparse
 * CNN, which train-mode forwards (mvd_volume_from_fused_train) update with momentum 0.01 like nn.BatchNorm1d (network.py:105).
 * out: device pointer, numel floats.  mvd_train_bn_calls: the number of such forwards since the weights were loaded (what
 * ``num_batches_tracked`` advanced by). */
int mvd_train_get_tensor(mvd_ctx* ctx, const char* name, float* out, size_t numel, void* stream);
int64_t mvd_train_bn_calls(mvd_ctx* ctx);
int mvd_train_adamw_step(mvd_ctx* ctx, float lr, float lr_aux, float beta1, float beta2, float eps, float weight_decay, int step,
                         float inv_scale, int finetune_unet, int* skipped_out, void* stream);
int mvd_train_repack(mvd_ctx* ctx);
/* The same in the order of `stream` (the stream the optimiser update was enqueued on and the next forward will be): no device
 * synchronisation -- the pack launches wait for an event on `stream`, later work on `stream` waits for them. */
int mvd_train_repack_async(mvd_ctx* ctx, void* stream);
/* DDP's bucketed, overlapped gradient averaging (train_morphable_diffusion.py:302-303: Lightning wraps the module in
 * DistributedDataParallel, whose reducer all-reduces a bucket as soon as its gradients are ready).  mvd_train_unet_step leaves
 * its UNet gradients as buckets in the order they become final during the backward pass -- one per chain of blocks (output
 * blocks 11..0, middle block, input blocks 11..0), then one for what completes at the end (the stacked emb_layers / attn2
 * projections, the head, time_embed) -- each a list of gradient-arena ranges (floats) plus an event recorded behind the last
 * kernel that writes them; together they cover model.diffusion_model.* exactly once.
 *   mvd_train_grad_bucket_count   buckets of the last step (0 before the first).
 *   mvd_train_grad_bucket         ranges of bucket k (offs / lens NULL: only *n_ranges).
 *   mvd_train_grad_bucket_wait    makes `stream` (the caller's communication stream) wait for bucket k's event.
 *   mvd_train_set_bucket_snapshot test hook: while set, every bucket's ranges are copied into `arena` (gradient-arena layout,
 *                                 device memory) right behind its event -- equal to the final gradients iff the ranges were final. */
int mvd_train_grad_bucket_count(mvd_ctx* ctx);
int mvd_train_grad_bucket(mvd_ctx* ctx, int k, int max_ranges, int64_t* offs, int64_t* lens, int* n_ranges);
int mvd_train_grad_bucket_wait(mvd_ctx* ctx, int k, void* stream);
int mvd_train_set_bucket_snapshot(mvd_ctx* ctx, float* arena);
/* The whole reducer behind the C ABI, on the communicator of mvd_comm_init (replaces DistributedDataParallel's reducer,
 * train_morphable_diffusion.py:302-303; no Python and no torch.distributed per bucket).  phase 0, called once right after
 * mvd_train_unet_step: every bucket's ranges are all-reduced (sum, in place in the gradient arena) on `comm_stream`, each bucket
 * behind its event.  phase 1, after the conditioner's backward was enqueued on `stream`: `comm_stream` waits for `stream`,
 * every arena range outside the buckets is reduced, `stream` waits for `comm_stream` and the arena is scaled by 1 / world on
 * `stream`.  phase 1 without a phase 0 reduces the whole arena (the flat all-reduce). */
int mvd_train_sync_gradients(mvd_ctx* ctx, int phase, void* comm_stream, void* stream);
/* Puts a volume [64,V,V,V] (reference layout, e.g. one sample of construct_spatial_volume's [B,64,V,V,V] result) back into
 * the context for mvd_frustum_volumes / mvd_denoise_views: with B > 1 samples per step (training_step) the per-sample
 * volumes are built first and the frustum stage runs afterwards (morphable_diffusion.py:531-533). */
int mvd_set_volume(mvd_ctx* ctx, const float* volume, void* stream);

/* SpatialVolumeNet.construct_view_frustum_volume (morphable_diffusion.py:265-320) for TN views of the
 * volume held in the context.  out_l: [TN,C_l,D_l,s_l,s_l] fp32 (reference layout), any may be NULL. */
int mvd_frustum_volumes(mvd_ctx* ctx, const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN,
                        float* out0, float* out1, float* out2, float* out3, void* stream);
/* The same for ONE target view of each of B samples (the training step, morphable_diffusion.py:496-518 with TN = 1): sample b's
 * frustum is gathered from volumes[b] [64,V,V,V] with the cameras of slot slots[b], then FrustumTV3DNet runs once with the B
 * volumes as its batch.  t_embed [B,time_dim], v_rows [B,view_dim] (the target view's embedding), view_idx [B] int32 -- device
 * pointers; out_l [B,C_l,D_l,s_l,s_l].  The active slot is unchanged on return. */
int mvd_frustum_volumes_batch(mvd_ctx* ctx, int B, const int* slots, const float* volumes, const float* t_embed, const float* v_rows,
                              const int32_t* view_idx, float* out0, float* out1, float* out2, float* out3, void* stream);

/* The per-view part of SyncDDIMSampler.denoise_apply (morphable_diffusion.py:721-738) for TN views, fully on
 * the device without leaving the library's layouts: frustum volumes -> CFG-batched UNet
 * (predict_with_unconditional_scale :132-149) -> DDIM update (:675-698).
 * x_noisy [TN,4,s,s], x_input [4,s,s], clip [context_dim], noise [TN,4,s,s] or NULL (is_step0),
 * coefficients from the DDIM tables; eps_out / x_prev [TN,4,s,s] (either may be NULL). */
int mvd_denoise_views(mvd_ctx* ctx, const float* x_noisy, const float* x_input, const float* clip, int64_t timestep,
                      const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN, float cfg_scale,
                      const float* noise, float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev, float dir_coef,
                      float sigma, float* eps_out, float* x_prev, void* stream);
/* The same step for B samples in ONE UNet pass (batch 2 * B * TN with guidance): the reference's own batching of
 * eval/generate_all_facescape.py:106-108,128-129,184-187, where denoise_apply (morphable_diffusion.py:701-739) receives B > 1.
 * slots[b]: the sample slot (mvd_select_sample) that holds sample b's mesh tables, cameras AND 32^3 volume (the volume built
 * by mvd_volume_from_fused while that slot was active); x_noisy / noise / eps_out / x_prev [B,TN,4,h,w]; x_input [B,4,h,w];
 * clip [B,context_dim]; timesteps [B] (host); t_embed [B,time_dim]; v_embed [B,TN,view_dim]; view_idx [TN] (the same views
 * of every sample).  B == 1 with slots == NULL is mvd_denoise_views.  Per sample the result equals the single-sample call to
 * fp16 operand rounding (the larger batch changes tile plans, i.e. fp32 summation orders), not bit for bit. */
int mvd_denoise_views_batch(mvd_ctx* ctx, int B, const int* slots, const float* x_noisy, const float* x_input, const float* clip,
                            const int64_t* timesteps, const float* t_embed, const float* v_embed, const int32_t* view_idx, int TN,
                            float cfg_scale, const float* noise, float sqrt_one_minus_at, float sqrt_at, float sqrt_aprev,
                            float dir_coef, float sigma, float* eps_out, float* x_prev, void* stream);

/* ---- single-kernel hooks used by the parity tests (tests/test_gpu_ops.py) ---- */
int mvd_op_conv(mvd_ctx* ctx, const float* x_nchw, int B, int Cin, int H, int W, const float* w, const float* bias,
                int Cout, int ksize, int stride, int upsample, const float* resid_nchw, float* out_nchw, int force_splitk,
                void* stream);
/* (mvd_op_conv: force_splitk == -2 with Cin <= 8 runs the UNet's inference form of its first layer, -3 with Cout <= 4 that of its
 * last layer: exact fp32 on the vector ALU) */
/* a_half != 0: A is rounded to fp16 in HBM first (the layout of operand-only activations) and, for M >= 512,
 * the dense LDS-DMA GEMM runs; resid [M][N] (or null) is added in the epilogue; force_splitk > 0 fixes the
 * split-K factor */
int mvd_op_linear(mvd_ctx* ctx, const float* a, int M, int K, const float* w, const float* bias, int N, int geglu,
                  const float* resid, int a_half, int force_splitk, float* out, void* stream);
int mvd_op_group_norm(mvd_ctx* ctx, const float* x_nchw, int B, int C, int HW, int groups, const float* gamma,
                      const float* beta, float eps, int act, float* out_nchw, void* stream);
int mvd_op_layer_norm(mvd_ctx* ctx, const float* x, int rows, int C, const float* gamma, const float* beta, float* out,
                      void* stream);
int mvd_op_attention(mvd_ctx* ctx, const float* q, const float* k, const float* v, int B, int T, int heads, int d,
                     float* out, void* stream);
int mvd_op_conv3d(mvd_ctx* ctx, const float* x_ncdhw, int B, int Cin, int D, int H, int W, const float* w,
                  const float* bias, int Cout, int stride, int transposed, const float* resid, float* out, void* stream);
/* backward-kernel hooks used by tests/test_gpu_train_ops.py (each is compared with torch.autograd of the same op):
 * self-attention backward (q, k, v, d_out, dq, dk, dv: [B,T,heads*d] fp32; the kernels themselves work on fp16), GroupNorm
 * (+ SiLU / ReLU) backward and LayerNorm backward on channels-last fp32 ([B,rows,C] / [rows,C]). */
int mvd_op_attention_bwd(mvd_ctx* ctx, const float* q, const float* k, const float* v, const float* d_out, int B, int T, int heads,
                         int d, float* dq, float* dk, float* dv, void* stream);
int mvd_op_group_norm_bwd(mvd_ctx* ctx, const float* x, const float* dy, int B, int rows, int C, int groups, const float* gamma,
                          const float* beta, float eps, int act, float* dx, float* dgamma, float* dbeta, void* stream);
int mvd_op_layer_norm_bwd(mvd_ctx* ctx, const float* x, const float* dy, int rows, int C, const float* gamma, float* dx,
                          float* dgamma, float* dbeta, void* stream);
/* Row-chain kernel test / timing hook (csrc/k_rowchain.hip): the row-local tail of a SpatialTransformer block in ONE launch,
 *   t2 = ao @ w_ao^T + b_ao + rowbias[sample] + xin   (flags & 1; otherwise t2 = xin)     ldm/modules/attention.py:196-200, 266-267
 *   t3 = t2 + FF2(GEGLU(FF1(LayerNorm(t2))))                                              ldm/modules/attention.py:37-73, 268
 *   out = t3 @ w_po^T + b_po + resid                  (flags & 2; otherwise out = fp16(t3)) ldm/modules/attention.py:333-336
 * on fp32 operands in the reference's layouts (w1 [8C][C] value rows then gate rows, w2 [C][4C]); rows % 128 == 0, T (rows per
 * sample) % 32 == 0, C in {64, 128, 256, 320}.  flags & 4 (without 2): the fp16 result is written as [hi | lo | hi] rows and
 * returned as hi + lo; flags & 8 (with 2): proj_out in extended precision.  iters > 0: the launch is repeated and *ms_out receives the mean milliseconds per launch. */
int mvd_op_st_tail(mvd_ctx* ctx, int C, int rows, int T, const float* ao, const float* xin, const float* rowbias,
                   const float* w_ao, const float* b_ao, const float* ln_g, const float* ln_b, const float* w1, const float* b1,
                   const float* w2, const float* b2, const float* w_po, const float* b_po, const float* resid, float* out,
                   int flags, int iters, float* ms_out, void* stream);
/* Row-head kernel test / timing hook (csrc/k_rowchain.hip: rowhead_kernel), the row-local front of a SpatialTransformer block in ONE
 * launch, C = 320:  t0 = n0 @ w_pi^T + b_pi (fp32 out; n0 = the GroupNorm output, rounded to fp16 as in the engine);
 * qkv = LayerNorm(t0; ln_g, ln_b) @ [w_q; w_k; w_v]^T (fp16 in the engine, returned as fp32 [rows][3C]).
 * ldm/modules/attention.py:325-332, 266, 186-190.  rows % 128 == 0.  xp != 0: proj_in in extended precision (fp16 hi + lo parts of
 * both operands, three products) -- n0 is then NOT rounded to fp16 first. */
int mvd_op_st_head(mvd_ctx* ctx, int rows, int xp, const float* n0, const float* w_pi, const float* b_pi, const float* ln_g,
                   const float* ln_b, const float* w_q, const float* w_k, const float* w_v, float* t0_out, float* qkv_out,
                   int iters, float* ms_out, void* stream);
/* time of the dominant kernel, for bench.py: runs the 3x3 conv implicit GEMM `iters` times on stream and
 * returns the mean kernel time in ms measured with HIP events on that stream */
int mvd_bench_conv(mvd_ctx* ctx, int B, int C, int H, int W, int Cout, int iters, float* ms_out, void* stream);
/* First-stage decoder.  Replaces AutoencoderKL.decode (ldm/models/autoencoder.py:330-333 =
 * post_quant_conv + Decoder.forward, ldm/modules/diffusionmodules/model.py:535-568) for a batch of latents:
 * z [B, embed_dim, h, w] fp32 NCHW on the device (already divided by the 0.18215 scale factor, as
 * decode_first_stage does, morphable_diffusion.py:468-471) -> out [B, out_ch, 8h, 8w] fp32 NCHW.  Needs the
 * first_stage_model.decoder.* and first_stage_model.post_quant_conv.* tensors uploaded before finalize. */
int mvd_vae_decode(mvd_ctx* ctx, const float* z, int B, int h, int w, float* out, void* stream);
/* AutoencoderKL.encode(x).parameters (autoencoder.py:324-328 = Encoder.forward model.py:434-459 + quant_conv):
 * x [B, 3, H, W] fp32 NCHW in [-1,1] -> moments [B, 2*embed_dim, H/8, W/8] = (mean | logvar).  Sampling from the
 * posterior (DiagonalGaussianDistribution.sample / .mode) stays host code so the RNG stream is torch's.  Needs the
 * first_stage_model.encoder.* and first_stage_model.quant_conv.* tensors. */
int mvd_vae_encode(mvd_ctx* ctx, const float* x, int B, int H, int W, float* moments, void* stream);
/* CLIP image embedding.  Replaces FrozenCLIPImageEmbedder.forward
 * (ldm/modules/encoders/modules.py:373-379 = preprocess :363-371 + clip's VisionTransformer.forward) as
 * SyncMultiviewDiffusion.prepare calls it (morphable_diffusion.py:487-488): x [B, 3, H, W] fp32 NCHW in [-1,1] on the
 * device -> out [B, embed] fp32 (the caller adds FrozenCLIPImageEmbedder.encode's unsqueeze(1)).  Needs the
 * clip_image_encoder.model.visual.* tensors (openai/CLIP key names; geometry is read off their shapes,
 * heads = width / 64 as clip's build_model does) uploaded before finalize. */
int mvd_clip_encode(mvd_ctx* ctx, const float* x, int B, int H, int W, float* out, void* stream);
/* embedding width of the uploaded CLIP vision tower (768 for ViT-L/14), 0 when none was uploaded */
int mvd_clip_embed_dim(mvd_ctx* ctx);
/* In-situ per-kernel-family timing for the roofline report (bench.py).  While enabled, launches are bracketed by HIP events
 * on their launch stream and booked under the kernel's template instance ("gemm_dma_kernel<128,0>",
 * "conv3_dma_kernel<160,16,16>", "igemm_kernel<1,128>", "splitk_reduce_kernel", "group_norm", "layernorm", "attn_kernel",
 * "depth_attn_kernel") together with their ALGORITHMIC flops and bytes (each operand and the result once).
 * mode 0: off.  mode 1: every launch of every family (a survey pass: it perturbs the step, use it outside the timed region).
 * mode 2: only `family`, a deterministic pseudo-random 1-in-`stride` sample of its launches (for the timed region).
 * A non-zero mode resets the counters.  mvd_probe_report synchronises on the recorded events and writes a JSON array
 * [{"family", "launches", "sampled", "ms", "flops", "bytes", "all_flops", "all_bytes"}, ...] (ms / flops / bytes: the bracketed
 * launches; all_*: every launch seen) into buf.  In mode 1 the report also carries the pseudo-family "(empty bracket)": event
 * pairs with NOTHING between them, recorded after every 8th launch -- ms / sampled of that row is what a bracket adds to a
 * launch's own duration (the second event's barrier packet), for the caller to subtract. */
int mvd_probe_config(mvd_ctx* ctx, int mode, const char* family, int stride);
int mvd_probe_report(mvd_ctx* ctx, char* buf, size_t cap);
/* same for a Linear layer [M,K] x [N,K]^T (fp16 operands in HBM); flags: 1 = fp32 residual add, 2 = fp16 output,
 * 4 = GEGLU epilogue, 8 = bias, 16 = per-sample bias (32 samples), 32 = cold operands (a 768 MiB memset evicts the caches
 * before every launch and each launch is timed on its own), 64 = fp32 activations */
int mvd_bench_linear(mvd_ctx* ctx, int M, int K, int N, int flags, int iters, float* ms_out, void* stream);
/* same for GroupNorm(groups) + SiLU of a channels-last fp32 [B, HW, C] tensor -> fp16 (split: the [hi | lo | hi] operand
 * of an extended-precision consumer); flags: 1 = split output */
int mvd_bench_group_norm(mvd_ctx* ctx, int B, int C, int HW, int groups, int flags, int iters, float* ms_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
