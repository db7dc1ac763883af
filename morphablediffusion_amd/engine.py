"""Thin object wrapper over the C ABI: owns one mvd_ctx on one GPU, hands torch device tensors to the
library by pointer (PyTorch is used for device memory and streams only)."""
import collections
import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import lib as L
from .spec import UNetConfig, VolumeConfig


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


IncompatibleKeys = collections.namedtuple("IncompatibleKeys", ["missing_keys", "unexpected_keys"])
MAX_SAMPLE_SLOTS = 64  # per-sample mesh / camera tables kept resident in the context (mvd_select_sample)


def _level_dims(d, s):
    """(depth, size) of the four frustum levels: stride-2 convolutions with padding 1 (network.py:320-333), i.e. (x - 1) // 2 + 1
    per level -- the same rule csrc/engine_cond.hip uses (x >> level only agrees for multiples of 8)."""
    out = []
    for _ in range(4):
        out.append((d, s))
        d, s = (d - 1) // 2 + 1, (s - 1) // 2 + 1
    return out


class Engine:
    def __init__(self, ucfg: UNetConfig, vcfg: VolumeConfig, device="cuda:0", workspace_gb: float = 16.0,
                 precision_level: int = 3, train: bool = False, vae_exact: bool = False):
        """precision_level: mvd_set_precision_level (0..6): how many of the output-side layers run with split fp16 operands
        (extended precision); 3 is the default the parity bounds are stated for (include/mvd.h: the ladder)."""
        self.precision_level = int(precision_level)
        self.vae_exact = bool(vae_exact)  # mvd_set_vae_precision: first-stage model in extended precision (~3x its cost)
        self.train_mode = bool(train)  # mvd_train_enable: master parameters / gradients kept in flat arenas (training step)
        if not torch.cuda.is_available():
            raise L.MvdError("no MI355X visible: the denoiser has no CPU path")
        self.lib = L.load()
        ucfg.validate()
        self.ucfg, self.vcfg = ucfg, vcfg
        self.device = torch.device(device)
        self._workspace_gb = workspace_gb
        self._ctx = None
        self._loaded = False
        self._create()

    def _create(self):
        ucfg, vcfg, workspace_gb = self.ucfg, self.vcfg, self._workspace_gb
        uc = L.UNetConfigC()
        uc.image_size, uc.in_channels, uc.out_channels = ucfg.image_size, ucfg.in_channels, ucfg.out_channels
        uc.model_channels, uc.num_res_blocks = ucfg.model_channels, ucfg.num_res_blocks
        for i in range(4):
            uc.channel_mult[i] = ucfg.channel_mult[i]
            uc.volume_dims[i] = ucfg.volume_dims[i]
        uc.num_heads, uc.context_dim = ucfg.num_heads, ucfg.context_dim
        uc.attention_levels = sum(int(r) for r in ucfg.attention_resolutions)
        vc = L.VolumeConfigC()
        vc.time_dim, vc.view_dim, vc.num_views = vcfg.time_dim, vcfg.view_dim, vcfg.num_views
        vc.input_image_size, vc.frustum_volume_depth = vcfg.input_image_size, vcfg.frustum_volume_depth
        vc.spatial_volume_size = vcfg.spatial_volume_size
        vc.spatial_volume_length, vc.frustum_volume_length = vcfg.spatial_volume_length, vcfg.frustum_volume_length
        if vcfg.projection not in ("perspective", "orthographic"):
            raise NotImplementedError(vcfg.projection)  # utils.py:41,66
        vc.projection = 0 if vcfg.projection == "perspective" else 1
        for i in range(4):
            vc.frustum_dims[i] = vcfg.frustum_dims[i]
        vc.voxel_size = vcfg.voxel_size
        self._ctx = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self.lib.mvd_create(C.byref(uc), C.byref(vc), self.device.index or 0,
                                        C.c_size_t(int(workspace_gb * (1 << 30))), C.byref(self._ctx)))
        L.check(self.lib.mvd_set_precision_level(self._ctx, self.precision_level))
        L.check(self.lib.mvd_train_enable(self._ctx, 1 if self.train_mode else 0))
        L.check(self.lib.mvd_set_vae_precision(self._ctx, 1 if self.vae_exact else 0))
        self._loaded = False
        self.flat_params = self.flat_grads = self.flat_m = self.flat_v = None
        self.param_table = {}
        self.num_vertices = 0
        self._slot_nv = {}

    def close(self):
        if getattr(self, "_ctx", None):
            self.lib.mvd_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = False, expected=None):
        """Key-for-key upload of a reference state_dict (generate_face.py:75-76) and packing.  Like nn.Module it may be
        called again (another checkpoint, EMA weights): the context is rebuilt, so every call must carry the complete set.
        Returns (missing_keys, unexpected_keys) w.r.t. ``expected`` (default: the hot-path manifest); ``strict`` raises on
        either, as torch does.
        Keys outside the path (``num_batches_tracked``, schedule buffers, the CLIP text tower, ...) count as unexpected only
        under strict=True, exactly as they would for a module that does not declare them."""
        if self._loaded:  # packed weights are immutable: start from a fresh context
            self.close()
            self._create()
        if expected is None:  # the hot-path manifest; a stand-alone UNet passes its own (DepthWiseAttention.load_state_dict)
            from .spec import full_manifest
            expected = full_manifest(self.ucfg, self.vcfg)
        want = set(expected)
        have = {k for k, v in sd.items() if torch.is_tensor(v)}
        missing = sorted(want - have)
        side = ("first_stage_model.", "clip_image_encoder.")
        unexpected = sorted(k for k in have - want if not k.startswith(side))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} (+{max(0, len(missing) - 5)}), "
                               f"unexpected {unexpected[:5]} (+{max(0, len(unexpected) - 5)})")
        self.has_vae_decoder = any(k.startswith("first_stage_model.decoder.") for k in sd)
        self.has_vae_encoder = any(k.startswith("first_stage_model.encoder.") for k in sd)
        self.has_clip = any(k.startswith("clip_image_encoder.model.visual.") for k in sd)
        for k, v in sd.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            t = v.detach().to(dtype=torch.float32).contiguous()
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            on_dev = 1 if t.is_cuda else 0
            L.check(self.lib.mvd_upload_weight(self._ctx, k.encode(), L.ptr(t), shape, t.dim(), on_dev))
        L.check(self.lib.mvd_finalize_weights(self._ctx))
        self._loaded = True
        if self.train_mode:
            self._adopt_arenas()
            # what a checkpoint written after training must carry besides the masters (export_state_dict): references to the
            # tensors the training step does not change (first stage, CLIP, schedule buffers, ...) -- no copies
            self._loaded_sd = dict(sd)
        return IncompatibleKeys(missing, unexpected)

    def get_tensor(self, key):
        """Current value of a resident tensor of a training context (a master parameter or a BatchNorm running statistic)."""
        ref = self._loaded_sd[key]
        out = torch.empty(tuple(ref.shape), device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_train_get_tensor(self._ctx, key.encode(), L.ptr(out), C.c_size_t(out.numel()), _stream()))
        return out

    def export_state_dict(self):
        """The loaded state_dict with every trained tensor replaced by its current value: master parameters from the arena, the
        sparse CNN's BatchNorm running statistics as the train-mode forwards left them (momentum 0.01, network.py:105),
        ``num_batches_tracked`` advanced by the number of those forwards.  Same keys, shapes and dtypes as what was loaded --
        the reference's ``torch.save({'state_dict': model.state_dict()})`` of a fine-tuned model."""
        if not self.train_mode or not self._loaded:
            raise L.MvdError("export_state_dict needs a training context with loaded weights")
        out = collections.OrderedDict()
        calls = int(self.lib.mvd_train_bn_calls(self._ctx))
        for k, v in self._loaded_sd.items():
            if not torch.is_tensor(v):
                out[k] = v
            elif k in self.param_table:
                out[k] = self.param_view(k).detach().clone().to(dtype=v.dtype)
            elif k.startswith("spatial_volume.") and k.endswith((".running_mean", ".running_var")):
                out[k] = self.get_tensor(k).to(dtype=v.dtype)
            elif k.startswith("spatial_volume.") and k.endswith(".num_batches_tracked"):
                out[k] = v.detach().clone() + calls
            else:
                out[k] = v
        return out

    # ---- stages --------------------------------------------------------------------------------
    def unet_forward(self, x, timesteps, context, source_dict, n_ctx: Optional[int] = None):
        """DepthWiseAttention.forward (attention.py:117-138). source_dict {res: [n_ctx,C,D,res,res]}."""
        dev = self.device
        Bv = x.shape[0]
        s = self.ucfg.image_size
        if context.dim() != 3 or context.shape[1] != 1:
            raise NotImplementedError("cross-attention context must be a single token [B,1,768] "
                                      "(morphable_diffusion.py:512)")
        x = _f32(x, dev)
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        ctx = _f32(context, dev)
        srcs = []
        for lvl in range(4):
            r = s >> lvl
            if r not in source_dict:
                raise KeyError(r)
            srcs.append(_f32(source_dict[r], dev))
        if n_ctx is None:
            n_ctx = srcs[0].shape[0]
        depth0 = srcs[0].shape[2]
        out = torch.empty(Bv, self.ucfg.out_channels, s, s, device=dev, dtype=torch.float32)
        L.check(self.lib.mvd_unet_forward(self._ctx, L.ptr(x), L.ptr(t), L.ptr(ctx), Bv, n_ctx, L.ptr(srcs[0]),
                                          L.ptr(srcs[1]), L.ptr(srcs[2]), L.ptr(srcs[3]), depth0, L.ptr(out), _stream()))
        return out

    def unet_block(self, path, x, timesteps=None, context=None, volume=None):
        """One block of DepthWiseAttention through the production block code: ``path`` is the reference module path below
        ``model.diffusion_model`` ("input_blocks.4.0", "output_blocks.8.2", "middle_conditions", "output_conditions.8", ...);
        ResBlocks take ``timesteps`` [B], SpatialTransformers ``context`` [B,1,768], DepthTransformers ``volume`` [B,C,D,H,W]."""
        dev = self.device
        x = _f32(x, dev)
        B, Cx, H, W = x.shape
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous() if timesteps is not None else None
        ctx = _f32(context, dev) if context is not None else None
        vol = _f32(volume, dev) if volume is not None else None
        cap = B * H * W * 4 * self.ucfg.model_channels * 8
        out = torch.empty(cap, device=dev, dtype=torch.float32)
        shape = (C.c_int * 4)()
        L.check(self.lib.mvd_unet_block(self._ctx, path.encode(), L.ptr(x), B, Cx, H, W, L.ptr(t), L.ptr(ctx), L.ptr(vol),
                                        vol.shape[2] if vol is not None else 0, L.ptr(out), cap, shape, _stream()))
        n = shape[0] * shape[1] * shape[2] * shape[3]
        return out[:n].view(shape[0], shape[1], shape[2], shape[3]).clone()

    def embed_time(self, t):
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        out = torch.empty(t.shape[0], self.vcfg.time_dim, device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_embed_time(self._ctx, L.ptr(t), t.shape[0], L.ptr(out), _stream()))
        return out

    def select_sample(self, slot: int):
        """Makes slot ``slot`` (0..MAX_SAMPLE_SLOTS-1) of per-sample mesh / camera tables the active one; set_mesh /
        set_cameras write into the active slot.  With B > 1 every sample keeps its tables across steps."""
        L.check(self.lib.mvd_select_sample(self._ctx, int(slot)))
        self._slot = int(slot)
        self.num_vertices = self._slot_nv.get(self._slot, 0)

    def set_mesh(self, vertices, coord, out_sh, bounds):
        """Per-sample mesh metadata (hoists the .tolist() sync of morphable_diffusion.py:251-252).  Host tensors are read before
        the call returns; the tables reach the device in the order of the current stream (mvd_set_mesh_async), so a training
        step that sees a new mesh per sample neither allocates nor synchronises here."""
        v = vertices.detach().cpu().float().contiguous()
        c = coord.detach().cpu().to(torch.int32).contiguous()
        o = out_sh.detach().cpu().to(torch.int32).contiguous()
        b = bounds.detach().cpu().float().contiguous()
        L.check(self.lib.mvd_set_mesh_async(self._ctx, L.ptr(v), L.ptr(c), L.ptr(o), L.ptr(b), v.shape[0], _stream()))
        self.num_vertices = v.shape[0]
        self._slot_nv[getattr(self, "_slot", 0)] = v.shape[0]

    def set_samples(self, slots, vertices, coord, out_sh, bounds, K, RT):
        """mvd_set_samples_async: mesh + cameras of several samples (lists, one entry per sample) into the slots ``slots``; the
        rule books are built on one host thread per sample, the uploads are enqueued on the current stream."""
        n = len(slots)
        v = [t.detach().cpu().float().contiguous() for t in vertices]
        c = [t.detach().cpu().to(torch.int32).contiguous() for t in coord]
        o = [t.detach().cpu().to(torch.int32).contiguous() for t in out_sh]
        b = [t.detach().cpu().float().contiguous() for t in bounds]
        k = [t.detach().cpu().float().contiguous() for t in K]
        r = [t.detach().cpu().float().contiguous() for t in RT]
        if any(t.shape[-2:] != (4, 4) for t in k):
            raise ValueError("target_K must be [N,4,4] (morphable_diffusion.py:296)")
        if len({t.shape[0] for t in k}) != 1:
            raise ValueError("every sample of a batch has the same number of target views")
        arr = lambda ts, ty: (ty * n)(*[C.cast(L.ptr(t), ty) for t in ts])
        fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.check(self.lib.mvd_set_samples_async(self._ctx, n, (C.c_int * n)(*[int(x) for x in slots]), arr(v, fp), arr(c, ip), arr(o, ip),
                                               arr(b, fp), (C.c_int * n)(*[t.shape[0] for t in v]), arr(k, fp), arr(r, fp),
                                               k[0].shape[0], _stream()))
        for sl, t in zip(slots, v):
            self._slot_nv[int(sl)] = t.shape[0]
        self.num_vertices = self._slot_nv.get(getattr(self, "_slot", 0), 0)

    def set_cameras(self, K, RT):
        K = K.detach().cpu().float().contiguous()
        RT = RT.detach().cpu().float().contiguous()
        if K.shape[-2:] != (4, 4):
            raise ValueError("target_K must be [N,4,4] (morphable_diffusion.py:296)")
        L.check(self.lib.mvd_set_cameras_async(self._ctx, L.ptr(K), L.ptr(RT), K.shape[0], _stream()))

    def vertex_features(self, x_noisy, t_embed, v_embed, view_idx, add_bias=True):
        dev = self.device
        x = _f32(x_noisy, dev)
        te, ve = _f32(t_embed, dev), _f32(v_embed, dev)
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty(self.num_vertices, 16, device=dev, dtype=torch.float32)
        L.check(self.lib.mvd_vertex_features(self._ctx, L.ptr(x), L.ptr(te), L.ptr(ve), L.ptr(vi), x.shape[0],
                                             1 if add_bias else 0, L.ptr(out), _stream()))
        return out

    def vertex_view_features(self, x_noisy, t_embed, v_embed, view_idx, out=None):
        """Per-view vertex features [n_local,Nv,16] (no view reduction): the operand of the all-gather view exchange."""
        dev = self.device
        x = _f32(x_noisy, dev)
        te, ve = _f32(t_embed, dev), _f32(v_embed, dev)
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        if out is None:
            out = torch.empty(x.shape[0], self.num_vertices, 16, device=dev, dtype=torch.float32)
        L.check(self.lib.mvd_vertex_view_features(self._ctx, L.ptr(x), L.ptr(te), L.ptr(ve), L.ptr(vi), x.shape[0],
                                                  L.ptr(out), _stream()))
        return out

    def vertex_features_stream_safe(self):
        """mvd_vertex_features_stream_safe: vertex_view_features may run on a side stream beside denoise_views of this engine."""
        return bool(self.lib.mvd_vertex_features_stream_safe(self._ctx))

    def fuse_vertex_features(self, vf_all, out=None):
        """SMPLFeatureExtractor over all views in index order: [num_views,Nv,16] -> [Nv,16]."""
        vf = _f32(vf_all, self.device)
        if out is None:
            out = torch.empty(self.num_vertices, 16, device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_fuse_vertex_features(self._ctx, L.ptr(vf), vf.shape[0], L.ptr(out), _stream()))
        return out

    # ---- the sharded step's exchange behind the C ABI (RCCL communicator owned by the library) ----------------------------
    def comm_init(self, rank=None, world=None):
        """Creates the library's RCCL communicator over the ranks of the default torch.distributed group (the 128-byte unique id
        is created on rank 0 and broadcast through that group -- any backend); world 1 needs no process group."""
        import torch.distributed as dist
        if world is None:
            world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if rank is None:  # derived independently of `world`: comm_init(world=W) alone must not leave rank 0 without the id
            rank = dist.get_rank() if (world > 1 and dist.is_available() and dist.is_initialized()) else 0
        ident = (C.c_char * 128)()
        if rank == 0:
            L.check(self.lib.mvd_comm_unique_id(C.byref(ident)))
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)
            ident = (C.c_char * 128).from_buffer_copy(box[0])
        L.check(self.lib.mvd_comm_init(self._ctx, C.byref(ident), int(rank), int(world)))
        self.comm_world = world

    def comm_destroy(self):
        L.check(self.lib.mvd_comm_destroy(self._ctx))
        self.comm_world = 0

    def exchange_view_features(self, vf_local, vf_all):
        """ncclAllGather of this rank's [n_local,Nv,16] into vf_all [N,Nv,16] on the current stream (mvd_exchange_view_features)."""
        assert vf_local.is_contiguous() and vf_all.is_contiguous() and vf_local.dtype == vf_all.dtype == torch.float32
        L.check(self.lib.mvd_exchange_view_features(self._ctx, L.ptr(vf_local), L.ptr(vf_all), vf_local.shape[0], _stream()))
        return vf_all

    def comm_all_reduce(self, buf):
        """In-place sum all-reduce of a contiguous float32 device tensor on the library's communicator (mvd_comm_all_reduce)."""
        assert buf.is_contiguous() and buf.dtype == torch.float32
        L.check(self.lib.mvd_comm_all_reduce(self._ctx, L.ptr(buf), C.c_size_t(buf.numel()), _stream()))
        return buf

    def sync_gradients(self, phase, comm_stream):
        """mvd_train_sync_gradients: phase 0 = all buckets of the last train_unet_step on ``comm_stream`` (a torch.cuda.Stream),
        phase 1 = the rest of the arena, the join with the current stream and the 1 / world scale."""
        L.check(self.lib.mvd_train_sync_gradients(self._ctx, int(phase), C.c_void_p(comm_stream.cuda_stream), _stream()))

    def stage_target_encoder(self, x_noisy, t_embed, v_embed):
        """NoisyTargetViewEncoder alone (network.py:181-207): x_noisy [n,4,s,s], t_embed [time_dim], v_embed [n,view_dim] ->
        [n,16,s,s].  Parity probe."""
        dev = self.device
        x, te, ve = _f32(x_noisy, dev), _f32(t_embed, dev), _f32(v_embed, dev)
        out = torch.empty(x.shape[0], 16, x.shape[2], x.shape[3], device=dev, dtype=torch.float32)
        L.check(self.lib.mvd_stage_target_encoder(self._ctx, L.ptr(x), L.ptr(te), L.ptr(ve), x.shape[0], L.ptr(out), _stream()))
        return out

    def stage_sparse_dense(self, fused, train=False):
        """The sparse voxel CNN alone (network.py:74-96): fused [Nv,16] -> dense [C,d,h,w] of the coarsest level.  Parity probe."""
        shp = (C.c_int32 * 4)()
        L.check(self.lib.mvd_stage_sparse_dense(self._ctx, None, 0, None, shp, _stream()))
        f = _f32(fused, self.device)
        out = torch.empty(*[int(v) for v in shp], device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_stage_sparse_dense(self._ctx, L.ptr(f), 1 if train else 0, L.ptr(out), shp, _stream()))
        return out

    def set_volume_ready_event(self, event):
        """event: torch.cuda.Event recorded after volume_from_fused on another stream (kept alive by the caller), or None."""
        h = C.c_void_p(0) if event is None else C.c_void_p(event.cuda_event)
        L.check(self.lib.mvd_set_volume_ready_event(self._ctx, h))

    def volume_from_fused(self, fused, want_output=True, train=False):
        """train: BatchNorm layers of the sparse CNN use batch statistics (the reference's module in train mode)."""
        V = self.vcfg.spatial_volume_size
        out = torch.empty(64, V, V, V, device=self.device, dtype=torch.float32) if want_output else None
        f = _f32(fused, self.device)
        fn = self.lib.mvd_volume_from_fused_train if train else self.lib.mvd_volume_from_fused
        L.check(fn(self._ctx, L.ptr(f), L.ptr(out), _stream()))
        return out

    # ---- training -----------------------------------------------------------------
    def _adopt_arenas(self):
        """The engine's master-parameter and gradient arenas move into torch-owned memory, so that parameters / gradients are
        VIEWS of two flat tensors (one collective for the DDP gradient averaging, torch optimisers work on them too)."""
        n = int(self.lib.mvd_train_arena_size(self._ctx))
        self.flat_params = torch.empty(n, device=self.device, dtype=torch.float32)
        self.flat_grads = torch.empty(n, device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_train_adopt_arena(self._ctx, 0, L.ptr(self.flat_params), C.c_int64(n)))
        L.check(self.lib.mvd_train_adopt_arena(self._ctx, 1, L.ptr(self.flat_grads), C.c_int64(n)))
        self.flat_m = self.flat_v = None
        self.param_table = {}
        name = C.create_string_buffer(512)
        off, numel, nd = C.c_int64(0), C.c_int64(0), C.c_int(0)
        shape = (C.c_int64 * 8)()
        for i in range(int(self.lib.mvd_train_param_count(self._ctx))):
            L.check(self.lib.mvd_train_param_info(self._ctx, i, name, C.c_size_t(len(name)), C.byref(off), C.byref(numel), shape,
                                                  C.byref(nd)))
            self.param_table[name.value.decode()] = (off.value, numel.value, tuple(int(shape[k]) for k in range(nd.value)))

    def param_view(self, key, grad=False):
        """View of one master parameter (or its gradient) inside the flat arena, in the reference's layout."""
        off, numel, shape = self.param_table[key]
        return (self.flat_grads if grad else self.flat_params)[off:off + numel].view(shape)

    def zero_grad(self):
        L.check(self.lib.mvd_train_zero_grad(self._ctx, _stream()))

    def train_unet_step(self, x, timesteps, context, source_dict, target, loss_scale=1.0, recompute=False, want_dsrc=False):
        """mvd_train_unet_step: forward + MSE loss + backward through every UNet block.  Returns (pred, loss[, dsrc dict]);
        parameter gradients (x loss_scale) are accumulated into ``flat_grads``."""
        if not self.train_mode:
            raise L.MvdError("the engine was not created with train=True")
        dev = self.device
        B = x.shape[0]
        s = self.ucfg.image_size
        x = _f32(x, dev)
        t = timesteps.to(device=dev, dtype=torch.int64).contiguous()
        ctx = _f32(context, dev)
        tgt = _f32(target, dev)
        srcs = [_f32(source_dict[s >> lvl], dev) for lvl in range(4)]
        depth0 = srcs[0].shape[2]
        pred = torch.empty(B, self.ucfg.out_channels, s, s, device=dev, dtype=torch.float32)
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        dsrc = [torch.empty_like(v) for v in srcs] if want_dsrc else [None] * 4
        L.check(self.lib.mvd_train_unet_step(self._ctx, L.ptr(x), L.ptr(t), L.ptr(ctx), B, L.ptr(srcs[0]), L.ptr(srcs[1]),
                                             L.ptr(srcs[2]), L.ptr(srcs[3]), depth0, L.ptr(tgt), C.c_float(loss_scale),
                                             1 if recompute else 0, L.ptr(pred), L.ptr(loss), L.ptr(dsrc[0]), L.ptr(dsrc[1]),
                                             L.ptr(dsrc[2]), L.ptr(dsrc[3]), _stream()))
        if want_dsrc:
            return pred, loss[0], {s >> lvl: dsrc[lvl] for lvl in range(4)}
        return pred, loss[0]

    def train_cond_backward(self, cond_index, x, context, d_out):
        """mvd_train_cond_backward (parity hook): one DepthTransformer's backward from exact inputs -> (dx, dcontext)."""
        dev = self.device
        x, context, d_out = _f32(x, dev), _f32(context, dev), _f32(d_out, dev)
        B, _, H, W = x.shape
        depth0 = context.shape[2] * (self.ucfg.image_size // H)
        dx, dc = torch.empty_like(x), torch.empty_like(context)
        L.check(self.lib.mvd_train_cond_backward(self._ctx, int(cond_index), L.ptr(x), L.ptr(context), L.ptr(d_out), B, H, W, depth0,
                                                 L.ptr(dx), L.ptr(dc), _stream()))
        return dx, dc

    def train_conditioner_backward(self, x_noisy, timestep, v_embed, target_index, dsrc, debug=False):
        """mvd_train_conditioner_backward for the ACTIVE sample: x_noisy [N,4,s,s], v_embed [N,4], dsrc {res: [1,C,D,res,res]}
        (loss-scaled).  Accumulates the gradients of spatial_volume.* / time_embed.*; debug=True also returns the intermediate
        gradients (d volume, d fused, d encoder features, d step embedding)."""
        dev = self.device
        x, ve = _f32(x_noisy, dev), _f32(v_embed, dev)
        s = self.vcfg.input_image_size // 8
        ds = [_f32(dsrc[s >> lvl], dev) for lvl in range(4)]
        dbg = [None] * 4
        if debug:
            V = self.vcfg.spatial_volume_size
            dbg = [torch.empty(64, V, V, V, device=dev), torch.empty(self.num_vertices, 16, device=dev),
                   torch.empty(x.shape[0], 16, x.shape[2], x.shape[3], device=dev), torch.empty(self.vcfg.time_dim, device=dev)]
        L.check(self.lib.mvd_train_conditioner_backward(self._ctx, L.ptr(x), C.c_int64(int(timestep)), L.ptr(ve), x.shape[0],
                                                        int(target_index), L.ptr(ds[0]), L.ptr(ds[1]), L.ptr(ds[2]), L.ptr(ds[3]),
                                                        L.ptr(dbg[0]), L.ptr(dbg[1]), L.ptr(dbg[2]), L.ptr(dbg[3]), _stream()))
        return dbg if debug else None

    def train_conditioner_backward_batch(self, slots, x_noisy, timesteps, v_embed, target_index, dsrc):
        """mvd_train_conditioner_backward_batch: the conditioner's backward for the samples resident in ``slots`` -- x_noisy
        [B,N,4,s,s], timesteps / target_index host ints [B], v_embed [B,N,4], dsrc {res: [B,C,D,res,res]} (loss-scaled).  The frustum
        network runs once with the samples as its batch; everything mesh-specific per sample."""
        dev = self.device
        x, ve = _f32(x_noisy, dev), _f32(v_embed, dev)
        B = x.shape[0]
        s = self.vcfg.input_image_size // 8
        ds = [_f32(dsrc[s >> lvl], dev) for lvl in range(4)]
        assert len(slots) == B == len(timesteps) == len(target_index) and all(d.shape[0] == B for d in ds)
        L.check(self.lib.mvd_train_conditioner_backward_batch(
            self._ctx, B, (C.c_int * B)(*[int(v) for v in slots]), L.ptr(x), (C.c_int64 * B)(*[int(v) for v in timesteps]), L.ptr(ve),
            x.shape[1], (C.c_int * B)(*[int(v) for v in target_index]), L.ptr(ds[0]), L.ptr(ds[1]), L.ptr(ds[2]), L.ptr(ds[3]),
            None, None, None, None, _stream()))

    def get_grad(self, key: str, shape):
        out = torch.empty(tuple(shape), device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_train_get_grad(self._ctx, key.encode(), L.ptr(out), C.c_size_t(out.numel()), _stream()))
        return out

    def grad_buckets(self):
        """Gradient buckets of the last train_unet_step (mvd_train_grad_bucket*): a list, in the order the gradients become final
        during the backward pass, of lists of (offset, length) ranges of ``flat_grads``; together model.diffusion_model.* once."""
        out = []
        for k in range(int(self.lib.mvd_train_grad_bucket_count(self._ctx))):
            n = C.c_int(0)
            L.check(self.lib.mvd_train_grad_bucket(self._ctx, k, 0, None, None, C.byref(n)))
            offs, lens = (C.c_int64 * max(1, n.value))(), (C.c_int64 * max(1, n.value))()
            L.check(self.lib.mvd_train_grad_bucket(self._ctx, k, n.value, offs, lens, C.byref(n)))
            out.append([(int(offs[i]), int(lens[i])) for i in range(n.value)])
        return out

    def grad_bucket_wait(self, k, stream):
        """Makes ``stream`` (a torch.cuda.Stream: the communication stream) wait until bucket k's gradients are final."""
        L.check(self.lib.mvd_train_grad_bucket_wait(self._ctx, int(k), C.c_void_p(stream.cuda_stream)))

    def set_bucket_snapshot(self, arena):
        """Test hook (mvd_train_set_bucket_snapshot): ``arena`` = a float32 device tensor shaped like ``flat_grads``, or None."""
        L.check(self.lib.mvd_train_set_bucket_snapshot(self._ctx, L.ptr(arena)))
        self._bucket_snapshot = arena  # keep it alive while the engine writes to it

    def adamw_step(self, lr, lr_aux, step, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, inv_scale=1.0, finetune_unet=True,
                   check=True):
        """torch.optim.AdamW on the arena (two learning-rate groups, morphable_diffusion.py:627-646) + in-place re-pack of the
        fp16 weights.  Returns True when the update was skipped because a gradient was inf / nan (check=False: not read back)."""
        self.ensure_moments()
        skipped = C.c_int(0)
        L.check(self.lib.mvd_train_adamw_step(self._ctx, C.c_float(lr), C.c_float(lr_aux), C.c_float(betas[0]),
                                              C.c_float(betas[1]), C.c_float(eps), C.c_float(weight_decay), int(step),
                                              C.c_float(inv_scale), 1 if finetune_unet else 0,
                                              C.byref(skipped) if check else None, _stream()))
        self.repack()
        return bool(skipped.value)

    def ensure_moments(self):
        """The two Adam moment arenas (zero until the first update), adopted by the library like the master / gradient arenas."""
        if self.flat_m is None:
            self.flat_m = torch.zeros_like(self.flat_params)
            self.flat_v = torch.zeros_like(self.flat_params)
            n = self.flat_params.numel()
            L.check(self.lib.mvd_train_adopt_arena(self._ctx, 2, L.ptr(self.flat_m), C.c_int64(n)))
            L.check(self.lib.mvd_train_adopt_arena(self._ctx, 3, L.ptr(self.flat_v), C.c_int64(n)))

    def repack(self):
        """Re-derive every packed fp16 weight from the master parameters (after they changed), in place, in the order of the
        current stream (mvd_train_repack_async: behind the optimiser update enqueued on it, ahead of the next forward)."""
        if os.environ.get("MVD_REPACK_SYNC") == "1":  # A/B switch: the device-synchronising form
            L.check(self.lib.mvd_train_repack(self._ctx))
        else:
            L.check(self.lib.mvd_train_repack_async(self._ctx, _stream()))

    def set_volume(self, volume):
        v = _f32(volume, self.device)
        L.check(self.lib.mvd_set_volume(self._ctx, L.ptr(v), _stream()))

    def mse_loss(self, a, b):
        a, b = _f32(a, self.device), _f32(b, self.device)
        out = torch.empty(1, device=self.device, dtype=torch.float32)
        L.check(self.lib.mvd_mse_loss(self._ctx, L.ptr(a), L.ptr(b), C.c_size_t(a.numel()), L.ptr(out), _stream()))
        return out[0]

    def frustum_volumes_batch(self, slots, volumes, t_embed, v_rows, view_idx):
        """mvd_frustum_volumes_batch: one target view per sample -- volumes [B,64,V,V,V], t_embed [B,time_dim], v_rows [B,view_dim],
        view_idx [B]; the samples' cameras are resident in ``slots``.  Returns {res: [B,C,D,res,res]}."""
        dev = self.device
        vol, te, vr = _f32(volumes, dev), _f32(t_embed, dev), _f32(v_rows, dev)
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        B = vol.shape[0]
        outs, ptrs = {}, []
        for lvl, (Dl, Sl) in enumerate(_level_dims(self.vcfg.frustum_volume_depth, self.vcfg.frustum_volume_size)):
            o = torch.empty(B, self.vcfg.frustum_dims[lvl], Dl, Sl, Sl, device=dev, dtype=torch.float32)
            outs[Sl] = o
            ptrs.append(L.ptr(o))
        L.check(self.lib.mvd_frustum_volumes_batch(self._ctx, B, (C.c_int * B)(*[int(v) for v in slots]), L.ptr(vol), L.ptr(te), L.ptr(vr),
                                                   L.ptr(vi), *ptrs, _stream()))
        return outs

    def frustum_volumes(self, t_embed, v_embed, view_idx):
        dev = self.device
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        TN = vi.shape[0]
        outs = {}
        ptrs = []
        for lvl, (Dl, Sl) in enumerate(_level_dims(self.vcfg.frustum_volume_depth, self.vcfg.frustum_volume_size)):
            o = torch.empty(TN, self.vcfg.frustum_dims[lvl], Dl, Sl, Sl, device=dev, dtype=torch.float32)
            outs[Sl] = o
            ptrs.append(L.ptr(o))
        te, ve = _f32(t_embed, dev), _f32(v_embed, dev)
        L.check(self.lib.mvd_frustum_volumes(self._ctx, L.ptr(te), L.ptr(ve), L.ptr(vi), TN, *ptrs, _stream()))
        return outs

    def denoise_views(self, x_noisy, x_input, clip, timestep, t_embed, v_embed, view_idx, cfg_scale, noise, coef,
                      want_eps=False, out=None, eps_out=None):
        """coef = (sqrt_one_minus_at, sqrt_at, sqrt_aprev, dir_coef, sigma) python floats.  ``out`` / ``eps_out``: contiguous
        float32 device tensors shaped like x_noisy that receive x_prev / the guided eps directly (no copy afterwards)."""
        dev = self.device
        x = _f32(x_noisy, dev)
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        TN = vi.shape[0]

        def direct(t):
            return t is not None and t.is_contiguous() and t.dtype == torch.float32 and t.device == x.device and t.shape == x.shape

        x_prev = out if direct(out) else torch.empty_like(x)
        eps = (eps_out if direct(eps_out) else torch.empty_like(x)) if want_eps else None
        nz = None if noise is None else _f32(noise, dev)
        xi, cl, te, ve = _f32(x_input, dev), _f32(clip, dev), _f32(t_embed, dev), _f32(v_embed, dev)
        L.check(self.lib.mvd_denoise_views(
            self._ctx, L.ptr(x), L.ptr(xi), L.ptr(cl), C.c_int64(int(timestep)),
            L.ptr(te), L.ptr(ve), L.ptr(vi), TN, C.c_float(cfg_scale), L.ptr(nz),
            C.c_float(coef[0]), C.c_float(coef[1]), C.c_float(coef[2]), C.c_float(coef[3]), C.c_float(coef[4]),
            L.ptr(eps), L.ptr(x_prev), _stream()))
        if out is not None and x_prev is not out:
            out.copy_(x_prev)
        if want_eps and eps_out is not None and eps is not eps_out:
            eps_out.copy_(eps)
        return (x_prev, eps) if want_eps else x_prev

    def denoise_views_batch(self, slots, x_noisy, x_input, clip, timesteps, t_embed, v_embed, view_idx, cfg_scale, noise, coef,
                            want_eps=False):
        """B samples in one UNet pass (mvd_denoise_views_batch): x_noisy / noise [B,TN,4,h,w], x_input [B,4,h,w], clip [B,768],
        timesteps: B python ints, t_embed [B,td], v_embed [B,TN,vd]; slots[b] holds sample b's tables, cameras and volume."""
        dev = self.device
        x = _f32(x_noisy, dev)
        B, TN = x.shape[:2]
        vi = view_idx.to(device=dev, dtype=torch.int32).contiguous()
        assert vi.shape[0] == TN and len(slots) == B and len(timesteps) == B
        x_prev = torch.empty_like(x)
        eps = torch.empty_like(x) if want_eps else None
        nz = None if noise is None else _f32(noise, dev)
        xi, cl, te, ve = _f32(x_input, dev), _f32(clip, dev), _f32(t_embed, dev), _f32(v_embed, dev)
        sl = (C.c_int * B)(*[int(v) for v in slots])
        ts = (C.c_int64 * B)(*[int(v) for v in timesteps])
        L.check(self.lib.mvd_denoise_views_batch(
            self._ctx, B, sl, L.ptr(x), L.ptr(xi), L.ptr(cl), ts, L.ptr(te), L.ptr(ve), L.ptr(vi), TN, C.c_float(cfg_scale),
            L.ptr(nz), C.c_float(coef[0]), C.c_float(coef[1]), C.c_float(coef[2]), C.c_float(coef[3]), C.c_float(coef[4]),
            L.ptr(eps), L.ptr(x_prev), _stream()))
        return (x_prev, eps) if want_eps else x_prev

    # ---- single-kernel hooks (parity tests) -------------------------------------------------------
    def op_conv(self, x, w, bias=None, stride=1, upsample=0, resid=None, force_splitk=0):
        dev = self.device
        x, w = _f32(x, dev), _f32(w, dev)
        B, Cin, H, W = x.shape
        Cout, k = w.shape[0], w.shape[2]
        Ho, Wo = ((H << upsample) - 1) // stride + 1, ((W << upsample) - 1) // stride + 1
        out = torch.empty(B, Cout, Ho, Wo, device=dev)
        b = None if bias is None else _f32(bias, dev)
        r = None if resid is None else _f32(resid, dev)
        L.check(self.lib.mvd_op_conv(self._ctx, L.ptr(x), B, Cin, H, W, L.ptr(w), L.ptr(b), Cout, k, stride, upsample,
                                     L.ptr(r), L.ptr(out), force_splitk, _stream()))
        return out

    def op_conv3d(self, x, w, bias=None, stride=1, transposed=False, resid=None):
        dev = self.device
        x, w = _f32(x, dev), _f32(w, dev)
        B, Cin, D, H, W = x.shape
        Cout = w.shape[1] if transposed else w.shape[0]
        if transposed:
            od = (2 * D, 2 * H, 2 * W)
        else:
            od = ((D - 1) // stride + 1, (H - 1) // stride + 1, (W - 1) // stride + 1)
        out = torch.empty(B, Cout, *od, device=dev)
        b = None if bias is None else _f32(bias, dev)
        r = None if resid is None else _f32(resid, dev)
        L.check(self.lib.mvd_op_conv3d(self._ctx, L.ptr(x), B, Cin, D, H, W, L.ptr(w), L.ptr(b), Cout, stride,
                                       1 if transposed else 0, L.ptr(r), L.ptr(out), _stream()))
        return out

    def op_linear(self, a, w, bias=None, geglu=False, resid=None, a_half=False, force_splitk=0):
        dev = self.device
        a, w = _f32(a, dev), _f32(w, dev)
        M, K = a.shape
        N = w.shape[0]
        out = torch.empty(M, N // 2 if geglu else N, device=dev)
        b = None if bias is None else _f32(bias, dev)
        r = None if resid is None else _f32(resid, dev)
        L.check(self.lib.mvd_op_linear(self._ctx, L.ptr(a), M, K, L.ptr(w), L.ptr(b), N, 1 if geglu else 0, L.ptr(r),
                                       1 if a_half else 0, int(force_splitk), L.ptr(out), _stream()))
        return out

    def op_group_norm(self, x, groups, gamma, beta, eps, act=0):
        dev = self.device
        x = _f32(x, dev)
        B, Cc = x.shape[:2]
        HW = x[0, 0].numel()
        out = torch.empty_like(x)
        g, b = _f32(gamma, dev), _f32(beta, dev)  # keep alive across the call
        L.check(self.lib.mvd_op_group_norm(self._ctx, L.ptr(x), B, Cc, HW, groups, L.ptr(g), L.ptr(b), C.c_float(eps),
                                           act, L.ptr(out), _stream()))
        return out

    def op_layer_norm(self, x, gamma, beta):
        dev = self.device
        x = _f32(x, dev)
        out = torch.empty_like(x)
        g, b = _f32(gamma, dev), _f32(beta, dev)  # keep alive across the call
        L.check(self.lib.mvd_op_layer_norm(self._ctx, L.ptr(x), x.shape[0], x.shape[1], L.ptr(g), L.ptr(b), L.ptr(out),
                                           _stream()))
        return out

    def op_st_tail(self, xin, ln_g, ln_b, w1, b1, w2, b2, ao=None, w_ao=None, b_ao=None, rowbias=None, T=None, w_po=None, b_po=None,
                   resid=None, split=False, iters=0, xp_out=False):
        """Row-chain kernel (csrc/k_rowchain.hip) through the C ABI: [to_out + t0] -> LayerNorm3 -> FF -> + t2 [-> proj_out + x_in].
        Returns the fp32 result (and the mean milliseconds per launch when iters > 0)."""
        dev = self.device
        xin = _f32(xin, dev)
        rows, Cc = xin.shape
        keep = [_f32(t, dev) if t is not None else None for t in (ao, rowbias, w_ao, b_ao, ln_g, ln_b, w1, b1, w2, b2, w_po, b_po, resid)]
        ao_, rb_, wao_, bao_, g_, b_, w1_, b1_, w2_, b2_, wpo_, bpo_, res_ = keep
        flags = (1 if ao is not None else 0) | (2 if w_po is not None else 0) | (4 if split else 0) | (8 if xp_out else 0)
        out = torch.empty_like(xin)
        ms = C.c_float(0)
        L.check(self.lib.mvd_op_st_tail(self._ctx, Cc, rows, int(T or rows), L.ptr(ao_), L.ptr(xin), L.ptr(rb_), L.ptr(wao_), L.ptr(bao_),
                                        L.ptr(g_), L.ptr(b_), L.ptr(w1_), L.ptr(b1_), L.ptr(w2_), L.ptr(b2_), L.ptr(wpo_), L.ptr(bpo_),
                                        L.ptr(res_), L.ptr(out), flags, int(iters), C.byref(ms), _stream()))
        return (out, ms.value) if iters > 0 else out

    def op_st_head(self, n0, w_pi, b_pi, ln_g, ln_b, w_q, w_k, w_v, iters=0, xp=False):
        """Row-head kernel (csrc/k_rowchain.hip) through the C ABI: proj_in -> t0, LayerNorm1, q | k | v.  Returns (t0, qkv[, ms])."""
        dev = self.device
        n0 = _f32(n0, dev)
        rows, Cc = n0.shape
        keep = [_f32(t, dev) for t in (w_pi, b_pi, ln_g, ln_b, w_q, w_k, w_v)]
        t0 = torch.empty(rows, Cc, device=dev)
        qkv = torch.empty(rows, 3 * Cc, device=dev)
        ms = C.c_float(0)
        L.check(self.lib.mvd_op_st_head(self._ctx, rows, 1 if xp else 0, L.ptr(n0), *[L.ptr(t) for t in keep], L.ptr(t0), L.ptr(qkv), int(iters),
                                        C.byref(ms), _stream()))
        return (t0, qkv, ms.value) if iters > 0 else (t0, qkv)

    def op_attention(self, q, k, v, heads):
        dev = self.device
        q, k, v = _f32(q, dev), _f32(k, dev), _f32(v, dev)
        B, T, Cc = q.shape
        out = torch.empty_like(q)
        L.check(self.lib.mvd_op_attention(self._ctx, L.ptr(q), L.ptr(k), L.ptr(v), B, T, heads, Cc // heads, L.ptr(out),
                                          _stream()))
        return out

    def op_attention_bwd(self, q, k, v, d_out, heads):
        dev = self.device
        q, k, v, d_out = _f32(q, dev), _f32(k, dev), _f32(v, dev), _f32(d_out, dev)
        B, T, Cc = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        L.check(self.lib.mvd_op_attention_bwd(self._ctx, L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(d_out), B, T, heads, Cc // heads,
                                              L.ptr(dq), L.ptr(dk), L.ptr(dv), _stream()))
        return dq, dk, dv

    def op_group_norm_bwd(self, x, dy, groups, gamma, beta, eps, act=0):
        """x, dy [B, rows, C] channels-last -> (dx, dgamma, dbeta)"""
        dev = self.device
        x, dy, g, b = _f32(x, dev), _f32(dy, dev), _f32(gamma, dev), _f32(beta, dev)
        B, rows, Cc = x.shape
        dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(g)
        L.check(self.lib.mvd_op_group_norm_bwd(self._ctx, L.ptr(x), L.ptr(dy), B, rows, Cc, groups, L.ptr(g), L.ptr(b), C.c_float(eps),
                                               act, L.ptr(dx), L.ptr(dg), L.ptr(db), _stream()))
        return dx, dg, db

    def op_layer_norm_bwd(self, x, dy, gamma):
        dev = self.device
        x, dy, g = _f32(x, dev), _f32(dy, dev), _f32(gamma, dev)
        dx, dg, db = torch.empty_like(x), torch.empty_like(g), torch.empty_like(g)
        L.check(self.lib.mvd_op_layer_norm_bwd(self._ctx, L.ptr(x), L.ptr(dy), x.shape[0], x.shape[1], L.ptr(g), L.ptr(dx), L.ptr(dg),
                                               L.ptr(db), _stream()))
        return dx, dg, db

    def bench_conv(self, B, Cc, H, W, Cout, iters=20):
        ms = C.c_float(0)
        L.check(self.lib.mvd_bench_conv(self._ctx, B, Cc, H, W, Cout, iters, C.byref(ms), _stream()))
        return ms.value

    def bench_linear(self, M, K, N, resid=False, out_half=False, geglu=False, iters=20, bias=False, rowbias=False, cold=False,
                     a_f32=False):
        ms = C.c_float(0)
        flags = ((1 if resid else 0) | (2 if out_half else 0) | (4 if geglu else 0) | (8 if bias else 0) | (16 if rowbias else 0) |
                 (32 if cold else 0) | (64 if a_f32 else 0))
        L.check(self.lib.mvd_bench_linear(self._ctx, M, K, N, flags, iters, C.byref(ms), _stream()))
        return ms.value

    def bench_group_norm(self, B, Cc, HW, groups=32, split=False, iters=20, apply_only=False):
        ms = C.c_float(0)
        L.check(self.lib.mvd_bench_group_norm(self._ctx, B, Cc, HW, groups, (1 if split else 0) | (2 if apply_only else 0), iters, C.byref(ms),
                                              _stream()))
        return ms.value

    def probe_config(self, mode, family=None, stride=1):
        """mvd_probe_config: 0 off, 1 every launch of every kernel family, 2 a 1-in-stride sample of ``family``."""
        L.check(self.lib.mvd_probe_config(self._ctx, int(mode), None if family is None else family.encode(), int(stride)))

    def probe_report(self):
        """Per-family table since the last probe_config: list of dicts (family, launches, sampled, ms, flops, bytes, ...)."""
        import json
        buf = C.create_string_buffer(1 << 16)
        L.check(self.lib.mvd_probe_report(self._ctx, buf, C.c_size_t(len(buf))))
        return json.loads(buf.value.decode())

    def vae_decode(self, z):
        """AutoencoderKL.decode (autoencoder.py:330-333) for a batch of latents z [B,4,h,w] (already divided by the
        first-stage scale factor) -> [B,3,8h,8w]; needs the first_stage_model.decoder.* weights in load_state_dict."""
        if not getattr(self, "has_vae_decoder", False):
            raise L.MvdError("first-stage decoder weights were not part of the uploaded state_dict")
        z = _f32(z, self.device)
        B, _, h, w = z.shape
        out = torch.empty(B, 3, 8 * h, 8 * w, device=self.device)
        # chunks of at most 16 x 256^2 output pixels: the kernels address an operand with 32-bit byte offsets (4 GiB), and the
        # widest decoder activation is 128 channels x 4 bytes per pixel
        per = max(1, (16 * 256 * 256) // (64 * h * w))
        for i in range(0, B, per):
            zc = z[i:i + per].contiguous()
            L.check(self.lib.mvd_vae_decode(self._ctx, L.ptr(zc), zc.shape[0], h, w, L.ptr(out[i:i + per]), _stream()))
        return out

    def clip_encode(self, x):
        """FrozenCLIPImageEmbedder.encode (ldm/modules/encoders/modules.py:363-382): x [B,3,H,W] in [-1,1] ->
        [B,1,embed]; needs the clip_image_encoder.model.visual.* weights in load_state_dict."""
        if not getattr(self, "has_clip", False):
            raise L.MvdError("CLIP vision-tower weights were not part of the uploaded state_dict")
        x = _f32(x, self.device)
        B, ch, H, W = x.shape
        if ch != 3:
            raise ValueError("clip_encode expects [B,3,H,W]")
        out = torch.empty(B, int(self.lib.mvd_clip_embed_dim(self._ctx)), device=self.device)
        L.check(self.lib.mvd_clip_encode(self._ctx, L.ptr(x), B, H, W, L.ptr(out), _stream()))
        return out.unsqueeze(1)

    def vae_encode_moments(self, x):
        """AutoencoderKL.encode(x).parameters (autoencoder.py:324-328): x [B,3,H,W] in [-1,1] -> [B,8,H/8,W/8]
        (mean | logvar); needs the first_stage_model.encoder.* weights in load_state_dict."""
        if not getattr(self, "has_vae_encoder", False):
            raise L.MvdError("first-stage encoder weights were not part of the uploaded state_dict")
        x = _f32(x, self.device)
        B, _, H, W = x.shape
        out = torch.empty(B, 8, H // 8, W // 8, device=self.device)
        L.check(self.lib.mvd_vae_encode(self._ctx, L.ptr(x), B, H, W, L.ptr(out), _stream()))
        return out
