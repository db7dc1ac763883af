"""Host-side mirror of the reference's plug-in surface for the denoising hot path.

Same class names, constructor kwargs, method names, argument meaning, batch-dict schema and state_dict keys
as the reference (ldm/models/diffusion/morphable_diffusion.py, ldm/models/diffusion/attention.py), so a
caller written against the reference (generate_face.py:227-243) runs unchanged; all arithmetic of the
denoising step executes in libmvd_hip.so.  The frozen CLIP image encoder and the first-stage VAE run in the engine
too when the checkpoint's ``clip_image_encoder.model.visual.*`` / ``first_stage_model.*`` tensors are loaded; otherwise
modules injected by the caller (``clip_image_encoder``, ``first_stage_model``) are used.
"""
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .engine import Engine
from .schedule import DDIMSchedule
from .spec import UNetConfig, VolumeConfig


def _unet_cfg(kw) -> UNetConfig:
    known = UNetConfig.__dataclass_fields__.keys()
    extra = {k: v for k, v in kw.items() if k not in known}
    for k in extra:
        if k not in ("dropout", "conv_resample", "dims", "num_classes", "use_fp16", "num_head_channels",
                     "num_heads_upsample", "use_scale_shift_norm", "resblock_updown", "use_new_attention_order",
                     "n_embed", "disable_self_attentions", "num_attention_blocks"):
            raise TypeError(f"unexpected UNet argument {k!r}")
    args = {k: (tuple(v) if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig" else v)
            for k, v in kw.items() if k in known}
    return UNetConfig(**args)


def instantiate_from_config(config):
    """ldm/util.py:217-232: ``target`` strings that name the reference's classes resolve to this module."""
    if "target" not in config:
        raise KeyError("Expected key `target` to instantiate.")
    name = config["target"].rsplit(".", 1)[1]
    table = {"DepthWiseAttention": DepthWiseAttention, "SyncMultiviewDiffusion": SyncMultiviewDiffusion}
    if name not in table:
        raise NotImplementedError(config["target"])
    return table[name](**config.get("params", dict()))


class DepthWiseAttention(nn.Module):
    """Drop-in for ldm.models.diffusion.attention.DepthWiseAttention (YAML unet_config.target,
    configs/facescape.yaml:26-42).  forward(x, timesteps, context, source_dict) -> [Bv,4,h,w]."""

    def __init__(self, volume_dims=(5, 16, 32, 64), *args, precision_level=3, train_mode=False, **kwargs):
        super().__init__()
        if args:
            raise TypeError("pass UNet arguments by keyword, as the reference config does")
        self.precision_level = precision_level  # not a reference kwarg: mvd_set_precision_level of a stand-alone engine
        self.train_mode = train_mode            # not a reference kwarg: a stand-alone engine keeps master weights / gradients
        self.cfg = _unet_cfg(dict(kwargs, volume_dims=tuple(volume_dims)))
        self.cfg.validate()
        self._engine: Optional[Engine] = None
        self._owns_engine = False
        self._trainable = {}

    def bind(self, engine: Engine):
        self._engine = engine

    def load_state_dict(self, state_dict, strict=True):
        """Standalone use: keys as in the reference UNet's own state_dict (no ``model.diffusion_model.`` prefix)."""
        if self._engine is None:
            self._engine = Engine(self.cfg, VolumeConfig(), precision_level=self.precision_level, train=self.train_mode)
            self._owns_engine = True
        from .spec import unet_manifest
        sd = {"model.diffusion_model." + k: v for k, v in state_dict.items()}
        inc = self._engine.load_state_dict(sd, strict=strict, expected=unet_manifest(self.cfg))
        self._keep_trainable(state_dict)
        n = len("model.diffusion_model.")
        return type(inc)([k[n:] for k in inc.missing_keys], [k[n:] for k in inc.unexpected_keys])

    def forward(self, x, timesteps=None, context=None, source_dict=None, **kwargs):
        if self._engine is None:
            raise RuntimeError("DepthWiseAttention has no weights: call load_state_dict first")
        return self._engine.unet_forward(x, timesteps, context, source_dict)

    def get_trainable_parameters(self):
        """attention.py:140-142: the parameters of middle_conditions and output_conditions, in the reference's registration
        order.  With the engine in training mode they are VIEWS of its flat master-parameter arena (nn.Parameter, ``.grad`` a
        view of the gradient arena: the training step accumulates into it, an optimiser step on them changes the engine's
        masters -- call ``engine.repack()`` afterwards, ArenaAdamW does).  Otherwise detached fp32 copies of the loaded tensors."""
        return list(self._trainable.values())

    def named_parameters_all(self):
        """(key relative to the UNet, nn.Parameter view) of EVERY UNet parameter, the reference's ``self.model.parameters()``
        (finetune_unet=True, morphable_diffusion.py:633-634).  Training mode only."""
        eng = self._engine
        pre = "model.diffusion_model."
        if eng is None or not eng.train_mode:
            raise RuntimeError("the engine was not created in training mode")
        if not hasattr(self, "_all_params"):
            self._all_params = {k[len(pre):]: _arena_param(eng, k) for k in sorted(eng.param_table) if k.startswith(pre)}
        return list(self._all_params.items())

    def _keep_trainable(self, sd, prefix=""):
        from .spec import unet_manifest
        self._trainable = {}
        self.__dict__.pop("_all_params", None)
        eng = self._engine
        for k in unet_manifest(self.cfg, prefix=""):
            if k.startswith(("middle_conditions.", "output_conditions.")) and prefix + k in sd:
                if eng is not None and eng.train_mode and ("model.diffusion_model." + k) in eng.param_table:
                    self._trainable[k] = _arena_param(eng, "model.diffusion_model." + k)
                else:
                    self._trainable[k] = nn.Parameter(sd[prefix + k].detach().float().clone())
        # deliberately NOT registered on the module: state_dict() stays what the reference's is (the engine owns the weights)


def _arena_param(eng, key):
    p_ = nn.Parameter(eng.param_view(key), requires_grad=True)
    p_.grad = eng.param_view(key, grad=True)
    return p_


class UNetWrapper(nn.Module):
    """morphable_diffusion.py:67-149 (inference paths)."""

    def __init__(self, diff_model_config, drop_conditions=False, drop_scheme="default", use_zero_123=True):
        super().__init__()
        if drop_scheme != "default":
            raise NotImplementedError  # morphable_diffusion.py:92
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.drop_conditions = drop_conditions
        self.drop_scheme = drop_scheme
        self.use_zero_123 = use_zero_123

    def get_trainable_parameters(self):
        return self.diffusion_model.get_trainable_parameters()

    def drop(self, cond, mask):
        return mask.view(cond.shape[0], *[1] * (cond.dim() - 1)) * cond

    def get_drop_scheme(self, B, device, random=None):
        """morphable_diffusion.py:84-93.  ``random``: the uniform draw [B] (default: torch.rand, as the reference)."""
        if self.drop_scheme != "default":
            raise NotImplementedError
        if random is None:
            random = torch.rand(B, dtype=torch.float32, device=device)
        random = random.to(device)
        return ((random > 0.15) & (random <= 0.2), (random > 0.1) & (random <= 0.15), (random > 0.05) & (random <= 0.1),
                random <= 0.05)

    def forward(self, x, t, clip_embed, volume_feats, x_concat, is_train=False, drop_random=None):
        """morphable_diffusion.py:95-130.  is_train with drop_conditions: the condition dropout of the training step (the
        forward pass runs in the engine; see SyncMultiviewDiffusion.training_step for what exists of the backward pass)."""
        if self.drop_conditions and is_train:
            B = x.shape[0]
            drop_clip, drop_volume, drop_concat, drop_all = self.get_drop_scheme(B, x.device, drop_random)
            clip_embed = self.drop(clip_embed, 1.0 - (drop_clip | drop_all).float())
            vm = 1.0 - (drop_volume | drop_all).float()
            for k, v in volume_feats.items():  # in place on the dict, as the reference does (:114-115)
                volume_feats[k] = self.drop(v, vm)
            x_concat = self.drop(x_concat, 1.0 - (drop_concat | drop_all).float())
        xc = x_concat * 1.0
        if self.use_zero_123:
            xc[:, :4] = xc[:, :4] / 0.18215
        return self.diffusion_model(torch.cat([x, xc], 1), t, clip_embed, source_dict=volume_feats)

    def train_step(self, x, t, clip_embed, volume_feats, x_concat, target, drop_random=None, loss_scale=1.0, recompute=True):
        """forward(is_train=True) (morphable_diffusion.py:95-130) + MSE + backward in one engine call.  Returns
        (pred, loss, dsrc): dsrc[res] = dL/d(volume_feats[res]) * loss_scale, after the dropout mask."""
        vm = None
        if self.drop_conditions:
            B = x.shape[0]
            drop_clip, drop_volume, drop_concat, drop_all = self.get_drop_scheme(B, x.device, drop_random)
            clip_embed = self.drop(clip_embed, 1.0 - (drop_clip | drop_all).float())
            vm = 1.0 - (drop_volume | drop_all).float()
            for k, v in volume_feats.items():
                volume_feats[k] = self.drop(v, vm)
            x_concat = self.drop(x_concat, 1.0 - (drop_concat | drop_all).float())
        xc = x_concat * 1.0
        if self.use_zero_123:
            xc[:, :4] = xc[:, :4] / 0.18215
        eng = self.diffusion_model._engine
        pred, loss, dsrc = eng.train_unet_step(torch.cat([x, xc], 1), t, clip_embed, volume_feats, target, loss_scale=loss_scale,
                                               recompute=recompute, want_dsrc=True)
        if vm is not None:  # the dropout is a multiplication by the mask: so is its adjoint
            dsrc = {k: self.drop(v, vm) for k, v in dsrc.items()}
        return pred, loss, dsrc

    def predict_with_unconditional_scale(self, x, t, clip_embed, volume_feats, x_concat, unconditional_scale):
        x_ = torch.cat([x] * 2, 0)
        t_ = torch.cat([t] * 2, 0)
        clip_ = torch.cat([clip_embed, torch.zeros_like(clip_embed)], 0)
        xc = torch.cat([x_concat, torch.zeros_like(x_concat)], 0)
        if self.use_zero_123:
            xc[:, :4] = xc[:, :4] / 0.18215
        eng = self.diffusion_model._engine
        # the unconditional half has all-zero volumes (morphable_diffusion.py:137-139): pass only the cond half
        s, s_uc = eng.unet_forward(torch.cat([x_, xc], 1), t_, clip_, volume_feats, n_ctx=x.shape[0]).chunk(2)
        return s_uc + unconditional_scale * (s - s_uc)


class SpatialVolumeNet(nn.Module):
    """morphable_diffusion.py:151-320, use_spatial_volume=False (both shipped configs)."""

    def __init__(self, time_dim, view_dim, view_num, input_image_size=256, frustum_volume_depth=48,
                 spatial_volume_size=32, spatial_volume_length=0.5, frustum_volume_length=0.86603,
                 projection="perspective", use_spatial_volume=False):
        super().__init__()
        if use_spatial_volume:
            raise NotImplementedError("use_spatial_volume=True (SpatialTime3DNet) is not used by any shipped config")
        self.cfg = VolumeConfig(time_dim=time_dim, view_dim=view_dim, num_views=view_num,
                                input_image_size=input_image_size, frustum_volume_depth=frustum_volume_depth,
                                spatial_volume_size=spatial_volume_size, spatial_volume_length=spatial_volume_length,
                                frustum_volume_length=frustum_volume_length, projection=projection)
        self.frustum_volume_size = input_image_size // 8
        self.frustum_volume_depth = frustum_volume_depth
        self.spatial_volume_size = spatial_volume_size
        self._engine: Optional[Engine] = None
        self._slots = {}  # slot -> (key, tensors): per-sample tables resident in the engine
        self._host = (None, None, None)  # (key, tensors, host copies) of the batch whose tables were staged last

    def bind(self, engine: Engine):
        self._engine = engine

    _SAMPLE_KEYS = ("vertices", "coord", "out_sh", "bounds", "target_K", "target_RT")

    def invalidate(self):
        """Forget which meshes / cameras are resident: the next step re-reads the batch (SyncDDIMSampler.sample does this
        on entry, so every sampling run uploads at least once -- the reference re-reads the batch at every step)."""
        self._slots = {}
        self._host = (None, None, None)

    def _host_tables(self, batch):
        """Host copies of the table inputs of ``batch``: ONE device-to-host transfer per tensor and batch (not per sample: every
        transfer drains the launch queue), none at all when the loader left them on the CPU."""
        ts = tuple(batch[k] for k in self._SAMPLE_KEYS)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device), t.dtype) for t in ts)
        if self._host[0] != key:
            self._host = (key, ts, {k: t.detach().cpu() for k, t in zip(self._SAMPLE_KEYS, ts)})
        return self._host[2]

    def _set_sample(self, batch, bi):
        """Makes sample ``bi`` of ``batch`` the engine's active mesh + cameras.  The tables are rebuilt only when the batch content
        changes: the key covers every tensor that feeds them -- storage address, shape and torch's in-place version counter --
        and the cache holds references to those tensors, so a freed tensor's address cannot come back under the same key.  A
        batch whose tables are not resident (a training step's new batch) is uploaded as a whole, once."""
        from .engine import MAX_SAMPLE_SLOTS
        slot = bi % MAX_SAMPLE_SLOTS
        held = self._slots.get(slot)
        if held is None or held[0] != self._sample_key(batch, bi):
            if batch["vertices"].shape[0] <= MAX_SAMPLE_SLOTS:
                self._upload_batch(batch)
            else:  # more samples than slots: samples share slots and are uploaded one by one, on use
                self._slots.pop(slot, None)
                h = self._host_tables(batch)
                self._engine.select_sample(slot)
                self._engine.set_mesh(h["vertices"][bi], h["coord"][bi], h["out_sh"][bi], h["bounds"][bi])
                self._engine.set_cameras(h["target_K"][bi], h["target_RT"][bi])
                self._slots[slot] = (self._sample_key(batch, bi), tuple(batch[k] for k in self._SAMPLE_KEYS))
        self._engine.select_sample(slot)

    def _sample_key(self, batch, bi):
        ts = tuple(batch[k] for k in self._SAMPLE_KEYS)
        return (bi,) + tuple((t.data_ptr(), t._version, tuple(t.shape), str(t.device), t.dtype) for t in ts)

    def _upload_batch(self, batch):
        """Tables of every sample of ``batch`` (at most MAX_SAMPLE_SLOTS) that is not resident yet, in ONE call: the rule books are
        built on one host thread per sample and uploaded in stream order (mvd_set_samples_async)."""
        ts = tuple(batch[k] for k in self._SAMPLE_KEYS)
        todo = [bi for bi in range(batch["vertices"].shape[0])
                if (self._slots.get(bi) or (None,))[0] != self._sample_key(batch, bi)]
        if not todo:
            return
        h = self._host_tables(batch)
        for bi in todo:
            self._slots.pop(bi, None)  # a failed upload must not leave a stale key behind
        self._engine.set_samples(todo, [h["vertices"][bi] for bi in todo], [h["coord"][bi] for bi in todo],
                                 [h["out_sh"][bi] for bi in todo], [h["bounds"][bi] for bi in todo],
                                 [h["target_K"][bi] for bi in todo], [h["target_RT"][bi] for bi in todo])
        for bi in todo:
            self._slots[bi] = (self._sample_key(batch, bi), ts)

    def construct_spatial_volume(self, x, t_embed, v_embed, batch):
        """train mode (nn.Module.train(), as Lightning sets it for training_step): the sparse CNN's BatchNorm layers use
        batch statistics, exactly like the reference's module in train mode."""
        B, N = x.shape[:2]
        vols = []
        for bi in range(B):
            self._set_sample(batch, bi)
            fused = self._engine.vertex_features(x[bi], t_embed[bi], v_embed[bi], torch.arange(N))
            vols.append(self._engine.volume_from_fused(fused, train=self.training))
        return torch.stack(vols)

    def construct_view_frustum_volume(self, spatial_volume, t_embed, v_embed, target_indices, batch):
        """B == 1 uses the volume held by the engine from the last construct_spatial_volume of the same sample; B > 1 (the
        training step: one target view per sample) re-uploads each sample's ``spatial_volume[bi]`` first."""
        B, TN = target_indices.shape
        from .engine import MAX_SAMPLE_SLOTS
        if 1 < B <= MAX_SAMPLE_SLOTS and TN == 1:  # the training step: the frustum network once, the samples as its batch
            for bi in range(B):
                self._set_sample(batch, bi)  # (tables resident, slot = sample index)
            idx = target_indices[:, 0].to(v_embed.device)
            v_rows = v_embed[torch.arange(B, device=v_embed.device), idx]
            return self._engine.frustum_volumes_batch(list(range(B)), spatial_volume, t_embed, v_rows, idx), None
        outs = []
        for bi in range(B):
            self._set_sample(batch, bi)
            if B > 1:
                self._engine.set_volume(spatial_volume[bi])
            idx = target_indices[bi]
            outs.append(self._engine.frustum_volumes(t_embed[bi], v_embed[bi][idx.to(v_embed.device)], idx))
        if B == 1:
            return outs[0], None
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}, None


class SyncMultiviewDiffusion(nn.Module):
    """morphable_diffusion.py:322-646, inference surface (sample / prepare / embed_time / ...)."""

    def __init__(self, unet_config, scheduler_config=None, finetune_unet=False, finetune_projection=True,
                 projection="perspective", use_spatial_volume=False, view_num=16, image_size=256, cfg_scale=3.0,
                 output_num=8, batch_view_num=4, drop_conditions=False, drop_scheme="default",
                 clip_image_encoder_path=None, sample_type="ddim", sample_steps=50, target_elevation=30,
                 first_stage_model=None, clip_image_encoder=None, device="cuda:0", workspace_gb=16.0, precision_level=3,
                 train_mode=False, loss_scale=65536.0, recompute=True, first_stage_precision="exact"):
        """train_mode / loss_scale / recompute are not reference kwargs: train_mode keeps fp32 master parameters, gradients and
        Adam moments in the engine (training_step runs the backward pass); loss_scale multiplies dL/dpred so that the fp16 MFMA
        operands of the backward pass stay in range (un-done by the optimiser); recompute = per-block activation checkpointing
        (the reference's use_checkpoint: True), False keeps every activation (fits the 288 GB of an MI355X, faster).
        first_stage_precision: "exact" (extended precision: <= 1e-3 relative; the default since round 4, so that the shipped
        path meets 1e-3 end to end -- decoding 16 views takes ~3x the fast mode's 16 ms, against a 0.7 s sampling loop) or "fast"
        (fp16 operands: decoded images within 0.8 of an 8-bit step of the reference's)."""
        if first_stage_precision not in ("fast", "exact"):
            raise ValueError("first_stage_precision must be 'fast' or 'exact'")
        super().__init__()
        self.finetune_unet = finetune_unet
        self.scheduler_config = scheduler_config
        self.learning_rate = 5e-5  # train_morphable_diffusion.py:313-321 sets model.learning_rate before fit
        self.train_mode = train_mode
        self.train_conditioner = True  # training_step also back-propagates into spatial_volume.* / time_embed.*
        self.loss_scale = float(loss_scale)
        self.recompute = bool(recompute)
        self.global_step = 0
        self.overlap_grad_sync = True  # DDP: bucketed all-reduces started by training_step (False: one flat all-reduce in sync_gradients)
        self._grad_sync = None
        self._grad_comm = None
        self.global_rank = 0
        self.image_dir = "."
        self.view_num = view_num
        self.viewpoint_dim = 4
        self.output_num = output_num
        self.image_size = image_size
        self.batch_view_num = batch_view_num
        self.cfg_scale = cfg_scale
        self.target_elevation = target_elevation
        self.time_embed_dim = 256
        self.first_stage_scale_factor = 0.18215
        self.first_stage_model = first_stage_model
        self.clip_image_encoder = clip_image_encoder
        self.num_timesteps = 1000
        self.model = UNetWrapper(unet_config, drop_conditions=drop_conditions, drop_scheme=drop_scheme)
        self.spatial_volume = SpatialVolumeNet(self.time_embed_dim, self.viewpoint_dim, view_num,
                                               input_image_size=image_size, projection=projection,
                                               use_spatial_volume=use_spatial_volume)
        self.engine = Engine(self.model.diffusion_model.cfg, self.spatial_volume.cfg, device=device,
                             workspace_gb=workspace_gb, precision_level=precision_level, train=train_mode,
                             vae_exact=first_stage_precision == "exact")
        self.model.diffusion_model.bind(self.engine)
        self.spatial_volume.bind(self.engine)
        self._device = torch.device(device)
        if sample_type != "ddim":
            raise NotImplementedError  # morphable_diffusion.py:359
        self.sampler = SyncDDIMSampler(self, sample_steps, "uniform", 1.0, latent_size=image_size // 8)

    @property
    def device(self):
        return self._device

    def load_state_dict(self, state_dict, strict=False):
        """generate_face.py:76 calls this with strict=False on ``ckpt['state_dict']``; returns torch's
        (missing_keys, unexpected_keys) pair w.r.t. the keys the denoising path consumes."""
        self.spatial_volume.invalidate()
        inc = self.engine.load_state_dict(state_dict, strict=strict)
        self.model.diffusion_model._keep_trainable(state_dict, "model.diffusion_model.")
        return inc

    def state_dict(self, *args, **kwargs):
        """Training mode: the checkpoint of the fine-tuned model under the reference's keys (Engine.export_state_dict) -- what
        Lightning's ModelCheckpoint saves and generate_face.py / the reference itself loads.  Otherwise nn.Module's."""
        if getattr(self.engine, "train_mode", False) and getattr(self.engine, "_loaded", False):
            return self.engine.export_state_dict()
        return super().state_dict(*args, **kwargs)

    def get_viewpoint_embedding(self, batch):
        d_e = torch.deg2rad(batch["target_elevation"]) - torch.deg2rad(batch["input_elevation"])
        d_a = torch.deg2rad(batch["target_azimuth"]) - torch.deg2rad(batch["input_azimuth"])
        return torch.stack([d_e, torch.sin(d_a), torch.cos(d_a), torch.zeros_like(d_a)], -1)

    def embed_time(self, t):
        return self.engine.embed_time(t)

    def encode_first_stage(self, x, sample=True):
        """morphable_diffusion.py:460-466.  With first_stage_model.encoder.* loaded the encoder runs in the HIP engine and
        only the posterior's sample()/mode() (torch RNG) stays on the host; otherwise the injected module is used."""
        if getattr(self.engine, "has_vae_encoder", False):
            posterior = DiagonalGaussianDistribution(self.engine.vae_encode_moments(x))
            z = posterior.sample() if sample else posterior.mode()
            return z.detach() * self.first_stage_scale_factor
        if self.first_stage_model is None:
            raise RuntimeError("no first_stage_model injected and no first_stage_model.encoder weights loaded")
        with torch.no_grad():
            posterior = self.first_stage_model.encode(x)
            z = posterior.sample() if sample else posterior.mode()
            return z.detach() * self.first_stage_scale_factor

    def decode_first_stage(self, z):
        """morphable_diffusion.py:468-471.  When the checkpoint's first_stage_model.decoder.* tensors were loaded the
        decoder runs in the HIP engine (any batch size: pass all views at once); otherwise in the injected module."""
        if getattr(self.engine, "has_vae_decoder", False):
            return self.engine.vae_decode(z / self.first_stage_scale_factor)
        if self.first_stage_model is None:
            raise RuntimeError("no first_stage_model injected and no first_stage_model.decoder weights loaded")
        with torch.no_grad():
            return self.first_stage_model.decode(z / self.first_stage_scale_factor)

    def prepare(self, batch, encode_targets=False):
        """morphable_diffusion.py:473-489.  ``encode_targets=True`` (the training step): the N target images are VAE-encoded
        view by view with posterior.sample(), exactly as the reference does (:475-479), and returned as x [B,N,4,h,w].
        At inference the reference encodes them too and then discards the result (:568); that dead work is skipped, but its
        side effect on the global RNG stream is not: each of those N encodes draws ``posterior.sample()`` noise
        (distributions.py:36) BEFORE the input image is encoded and before x_T is drawn, so the same draws are consumed here
        -- a run seeded with torch.manual_seed sees the same stream position as the reference."""
        x = None
        if "target_image" in batch and batch["target_image"] is not None:
            B, N = batch["target_image"].shape[:2]
            if encode_targets:
                image_target = batch["target_image"].permute(0, 1, 4, 2, 3)  # b,n,h,w,3 -> b,n,3,h,w
                x = torch.stack([self.encode_first_stage(image_target[:, ni], True) for ni in range(N)], 1)
            else:
                h, w = batch["target_image"].shape[2] // 8, batch["target_image"].shape[3] // 8
                for _ in range(N):
                    torch.randn([B, 4, h, w])
        image_input = batch["input_image"].permute(0, 3, 1, 2)
        x_input = self.encode_first_stage(image_input)
        input_info = {"image": image_input, "elevation": batch["input_elevation"][:, 0], "x": x_input}
        if getattr(self.engine, "has_clip", False):  # clip_image_encoder.model.visual.* were in the state_dict
            clip_embed = self.engine.clip_encode(image_input)
        elif self.clip_image_encoder is None:
            raise RuntimeError("no clip_image_encoder injected and no clip_image_encoder.model.visual weights loaded")
        else:
            with torch.no_grad():
                clip_embed = self.clip_image_encoder.encode(image_input)
        return x, clip_embed, input_info

    def add_noise(self, x_start, t, noise=None):
        """morphable_diffusion.py:551-565 (schedule buffers :428-450).  ``noise``: the N(0,1) draw (default: randn_like)."""
        B = x_start.shape[0]
        if noise is None:
            noise = torch.randn_like(x_start)
        ac = self.sampler.schedule.alphas_cumprod.to(x_start.device)
        shape = (B,) + (1,) * (x_start.dim() - 1)
        x_noisy = ac.sqrt()[t].view(shape) * x_start + (1.0 - ac).sqrt()[t].view(shape) * noise
        return x_noisy, noise

    def training_step(self, batch, prepared=None, time_steps=None, noise=None, target_index=None, drop_random=None,
                      backward=None):
        """SyncMultiviewDiffusion.training_step (morphable_diffusion.py:520-549) in the HIP engine: random time steps, prepare
        (VAE-encode the N target views + the input view, CLIP), add_noise, one random target view per sample, the 32^3 volume
        from ALL noisy views (BatchNorm in train mode), one frustum volume per sample, the UNet with condition dropout, MSE
        against the injected noise -- and, with the engine in training mode (``backward`` defaults to that), loss.backward():
        dL/dpred back through every UNet block; ``.grad`` of every UNet parameter (views of the engine's gradient arena,
        multiplied by ``self.loss_scale``) is ACCUMULATED like torch does until zero_grad.  ``self.last_dsrc`` holds dL/d(frustum
        volumes); from it the conditioner's backward (``self.train_conditioner``, default on) fills the gradients of
        ``spatial_volume.*`` and ``time_embed.*`` sample by sample.
        ``prepared`` = (x, clip_embed, input_info) replaces self.prepare(batch); the random draws may be passed in (parity
        tests), otherwise they are made on the host in the reference's order (randint for the time steps BEFORE prepare's
        posterior samples, then randn_like, randint, rand) so that torch.manual_seed reproduces the reference's CPU stream.
        Returns the loss (a device scalar, no autograd graph); the prediction is kept in ``self.last_noise_predict``."""
        dev = self.device
        if backward is None:
            backward = self.train_mode
        B = batch["target_image"].shape[0] if prepared is None else prepared[0].shape[0]
        if time_steps is None:  # drawn first, as the reference does (:521-522)
            time_steps = torch.randint(0, self.num_timesteps, (B,)).long()
        x, clip_embed, input_info = self.prepare(batch, encode_targets=True) if prepared is None else prepared
        B, N = x.shape[:2]
        if noise is None:
            noise = torch.randn(x.shape)
        if target_index is None:
            target_index = torch.randint(0, N, (B, 1)).long()
        if drop_random is None and self.model.drop_conditions:
            drop_random = torch.rand(B, dtype=torch.float32)
        time_steps, target_index = time_steps.to(dev), target_index.to(dev)
        x_noisy, noise = self.add_noise(x.to(dev), time_steps, noise.to(dev))
        was_training = self.training
        self.train()  # BatchNorm batch statistics in the sparse CNN, as in the reference's training_step
        try:
            v_embed = self.get_viewpoint_embedding(batch).to(dev)
            t_embed = self.embed_time(time_steps)
            sv = self.spatial_volume.construct_spatial_volume(x_noisy, t_embed, v_embed, batch)
            clip_, vf, xc = self.get_target_view_feats(input_info["x"].to(dev), sv, clip_embed.to(dev), t_embed, v_embed,
                                                       target_index, batch)
        finally:
            self.train(was_training)
        ar = torch.arange(B, device=dev)[:, None]
        target = noise[ar, target_index][:, 0].contiguous()
        x_t = x_noisy[ar, target_index][:, 0]
        if not backward:
            pred = self.model(x_t, time_steps, clip_, vf, xc, is_train=True, drop_random=drop_random)
            self.last_noise_predict = pred
            return self.engine.mse_loss(target, pred)
        pred, loss, dsrc = self.model.train_step(x_t, time_steps, clip_, vf, xc, target, drop_random=drop_random,
                                                 loss_scale=self.loss_scale, recompute=self.recompute)
        self.last_noise_predict, self.last_dsrc = pred, dsrc
        self._start_grad_sync()  # DDP: the UNet's buckets are reduced while the rest of the backward runs
        if self.train_conditioner:  # spatial_volume.* / time_embed.*: per sample, from dL/d(its frustum volumes)
            hs = [int(v) for v in time_steps.tolist()]
            ti = [int(v) for v in target_index[:, 0].tolist()]
            from .engine import MAX_SAMPLE_SLOTS
            if B <= MAX_SAMPLE_SLOTS:  # every sample's tables are resident (slot = sample index): one call, the frustum
                for bi in range(B):    # network batched over the samples
                    self.spatial_volume._set_sample(batch, bi)
                self.engine.train_conditioner_backward_batch(list(range(B)), x_noisy, hs, v_embed, ti, dsrc)
            else:
                for bi in range(B):
                    self.spatial_volume._set_sample(batch, bi)
                    self.engine.train_conditioner_backward(x_noisy[bi], hs[bi], v_embed[bi], ti[bi],
                                                           {k: v[bi:bi + 1] for k, v in dsrc.items()})
        return loss

    # ---- optimiser surface (morphable_diffusion.py:627-646, train_morphable_diffusion.py:302-321) --------------------
    def configure_optimizers(self):
        """AdamW with the reference's parameter groups and LambdaLR on its scheduler_config: the UNet (all of it when
        finetune_unet, else get_trainable_parameters()) at lr, time_embed and spatial_volume at 10 lr.  The optimiser is a
        torch.optim.Optimizer whose step() runs ONE fused HIP kernel per group range on the engine's arenas and re-packs the
        fp16 weights."""
        from torch.optim.lr_scheduler import LambdaLR
        lr = self.learning_rate
        print(f"setting learning rate to {lr:.4f} ...")
        if not self.train_mode:
            raise RuntimeError("configure_optimizers needs the model created with train_mode=True")
        dm = self.model.diffusion_model
        unet = [p_ for _, p_ in dm.named_parameters_all()] if self.finetune_unet else self.model.get_trainable_parameters()
        eng = self.engine
        te = [_arena_param(eng, k) for k in sorted(eng.param_table) if k.startswith("time_embed.")]
        sv = [_arena_param(eng, k) for k in sorted(eng.param_table) if k.startswith("spatial_volume.")]
        paras = [{"params": unet, "lr": lr}, {"params": te, "lr": lr * 10.0}, {"params": sv, "lr": lr * 10.0}]
        opt = ArenaAdamW(self, paras, lr=lr)
        if self.scheduler_config is None:
            return [opt], []
        sched = LambdaLinearScheduler(**self.scheduler_config.get("params", {}))
        print("Setting up LambdaLR scheduler...")
        return [opt], [{"scheduler": LambdaLR(opt, lr_lambda=sched.schedule), "interval": "step", "frequency": 1}]

    def sync_gradients(self):
        """DDP's gradient averaging (train_morphable_diffusion.py:302-303: Lightning wraps the module in
        DistributedDataParallel) -- RCCL over xGMI when the process group's backend is nccl.  With ``overlap_grad_sync`` (default)
        training_step has already started the bucketed all-reduces on the communication stream, behind the events the backward
        pass recorded (UNet buckets while the backward and the conditioner's backward still run); this call starts the rest,
        joins the stream and applies the 1 / world_size.  Otherwise ONE all-reduce on the flat gradient arena.  No-op without an
        initialised process group."""
        sync, self._grad_sync = self._grad_sync, None  # one averaging per backward pass, as with the flat all-reduce
        if self.overlap_grad_sync is None:
            # inside no_sync(): DistributedDataParallel suppresses EVERY reduction there -- a flat all-reduce here would average
            # (and scale by 1 / world) a half-accumulated arena, and the earlier micro-batches again at the final averaging
            return False
        if sync is not None:
            return sync.finish()
        return sync_flat_gradients(self.engine.flat_grads)

    def no_sync(self):
        """DistributedDataParallel.no_sync(): inside the context training_step starts no gradient all-reduce (gradient
        accumulation over several batches: only the last one, outside the context, is followed by the averaging)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = self.overlap_grad_sync
            self.overlap_grad_sync = None  # neither bucketed nor (by mistake) a flat one through a stale object
            try:
                yield
            finally:
                self.overlap_grad_sync = old
        return ctx()

    def _start_grad_sync(self):
        """Called by training_step right after the UNet's forward + backward is enqueued."""
        self._grad_sync = None
        if not self.overlap_grad_sync or not _dist_world() > 1:
            return
        eng = self.engine
        dev = eng.flat_grads.device
        if dev.type == "cuda":
            if getattr(self, "_grad_comm", None) is None:
                self._grad_comm = torch.cuda.Stream(device=dev)
            lib_comm = ensure_library_comm(eng, dev)
            self._grad_sync = BucketedGradSync(eng.flat_grads, eng.grad_buckets(), comm=self._grad_comm,
                                               wait=lambda k: eng.grad_bucket_wait(k, self._grad_comm),
                                               engine=eng if lib_comm else None)
        else:  # CPU stand-ins of the engine (tests): no streams, the buckets are reduced in order
            self._grad_sync = BucketedGradSync(eng.flat_grads, eng.grad_buckets())
        self._grad_sync.start()

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """morphable_diffusion.py:601-617: rank 0 samples the first ``output_num`` items of the first validation batch and
        writes the image grid <image_dir>/images/val/<step>.jpg."""
        if batch_idx == 0 and self.global_rank == 0:
            self.eval()
            batch_ = {}
            for k, v in batch.items():
                batch_[k] = {k_: v_[:self.output_num] for k_, v_ in v.items()} if isinstance(v, dict) else v[:self.output_num]
            x_sample = self.sample(self.sampler, batch_, self.cfg_scale, self.batch_view_num)
            from pathlib import Path
            from .batch import save_image_grid
            out = Path(self.image_dir) / "images" / "val"
            out.mkdir(exist_ok=True, parents=True)
            save_image_grid(x_sample, batch_, str(out / f"{self.global_step}.jpg"))
            return x_sample

    def log_image(self, x_sample, batch, step, output_dir):
        """morphable_diffusion.py:589-599: one row per sample -- the input view followed by the N generated views -- as
        <output_dir>/<step>.jpg."""
        from pathlib import Path
        from .batch import save_image_grid
        save_image_grid(x_sample, batch, str(Path(output_dir) / f"{step}.jpg"))

    @torch.no_grad()
    def test_step(self, batch, batch_idx):
        """morphable_diffusion.py:619-625: sample the batch, write <outdir>/<batch_idx>.jpg (``outdir`` is set by the caller, as
        the reference's trainer script does)."""
        from pathlib import Path
        self.eval()
        x_sample = self.sample(self.sampler, batch, self.cfg_scale, self.batch_view_num)
        out = Path(getattr(self, "outdir", "."))
        out.mkdir(exist_ok=True, parents=True)
        self.log_image(x_sample, batch, batch_idx, output_dir=out)
        return x_sample

    def get_target_view_feats(self, x_input, spatial_volume, clip_embed, t_embed, v_embed, target_index, batch):
        B, _, H, W = x_input.shape
        TN = target_index.shape[1]
        vf, _ = self.spatial_volume.construct_view_frustum_volume(spatial_volume, t_embed, v_embed, target_index, batch)
        clip_ = clip_embed.unsqueeze(1).repeat(1, TN, 1, 1).view(B * TN, 1, 768)
        x_ = x_input.unsqueeze(1).repeat(1, TN, 1, 1, 1).view(B * TN, 4, H, W)
        return clip_, vf, x_

    def sample(self, sampler, batch, cfg_scale, batch_view_num, return_inter_results=False, inter_interval=50,
               inter_view_interval=2):
        _, clip_embed, input_info = self.prepare(batch)
        x_sample, inter = sampler.sample(input_info, clip_embed, unconditional_scale=cfg_scale,
                                         log_every_t=inter_interval, batch_view_num=batch_view_num, batch=batch)
        B, N = x_sample.shape[:2]
        if getattr(self.engine, "has_vae_decoder", False):  # all views in one batched decode
            img = self.decode_first_stage(x_sample.reshape(B * N, *x_sample.shape[2:]))
            x_sample = img.reshape(B, N, *img.shape[1:])
        else:
            x_sample = torch.stack([self.decode_first_stage(x_sample[:, ni]) for ni in range(N)], 1)
        if return_inter_results:
            inter = torch.stack(inter["x_inter"], 2)
            T = inter.shape[2]
            res = [torch.stack([self.decode_first_stage(inter[:, ni, ti]) for ti in range(T)], 1)
                   for ni in range(0, N, inter_view_interval)]
            return x_sample, torch.stack(res, 1)
        return x_sample


def _dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def ensure_library_comm(eng, dev):
    """Under a launcher with one device per rank (process group on the nccl = RCCL backend) the collectives of the step -- the
    view exchange of the sharded sampler, the gradient reducer of the training step -- run on the LIBRARY's communicator
    (mvd_comm_init) instead of through Python's c10d: created here on first use, collectively (every rank gets here at the same
    point of its first step).  Falls back to torch.distributed -- on ALL ranks, agreed by a MIN all-reduce of the outcome --
    when librccl cannot be opened or the communicator cannot be created; MVD_NO_LIB_COMM=1 keeps c10d.  Returns True when the
    library communicator spans the process group."""
    import os
    import warnings
    import torch.distributed as dist
    world = _dist_world()
    if world <= 1 or getattr(dev, "type", str(dev).split(":")[0]) != "cuda":
        return False
    if getattr(eng, "comm_world", 0) == world:
        return True
    if getattr(eng, "_lib_comm_tried", False):
        return False
    eng._lib_comm_tried = True
    if os.environ.get("MVD_NO_LIB_COMM") or dist.get_backend() != "nccl":  # gloo stand-ins share one device: no RCCL there
        return False
    ok = 1
    try:
        eng.comm_init(dist.get_rank(), world)
    except Exception as exc:  # MvdError: librccl.so missing, ncclCommInitRank failed
        ok = 0
        warnings.warn(f"mvd_comm_init failed ({exc}); the step's collectives stay on torch.distributed")
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if ok:
            eng.comm_destroy()
        eng.comm_world = 0
        return False
    return True


class BucketedGradSync:
    """DDP's reducer on the flat gradient arena: ``buckets`` = lists of (offset, length) ranges in the order their gradients
    become final; ``start()`` enqueues one all-reduce per range on the communication stream ``comm``, each bucket behind
    ``wait(k)`` (which makes ``comm`` wait for the event of bucket k); ``finish()`` reduces every range of the arena no bucket
    covered (the conditioner's parameters, written after the UNet's backward), makes the caller's stream wait for ``comm`` and
    scales by 1 / world_size.  The arithmetic per element is the flat all-reduce's: sum over ranks, then the scale."""

    def __init__(self, flat, buckets, comm=None, wait=None, engine=None):
        self.flat, self.buckets, self.comm, self.wait = flat, buckets, comm, wait
        # engine: its library communicator spans the process group -> the whole reducer is two C calls (mvd_train_sync_gradients:
        # ncclAllReduce per range on `comm`, no Python per bucket); None: one dist.all_reduce per range (c10d)
        self.engine = engine if comm is not None else None
        self.done = False

    def _ctx(self):
        import contextlib
        return torch.cuda.stream(self.comm) if self.comm is not None else contextlib.nullcontext()

    def start(self):
        import torch.distributed as dist
        if self.engine is not None:
            self.engine.sync_gradients(0, self.comm)
            return
        with self._ctx():
            for k, ranges in enumerate(self.buckets):
                if self.wait is not None:
                    self.wait(k)
                for off, n in ranges:
                    dist.all_reduce(self.flat[off:off + n])

    def uncovered(self):
        """Ranges of the arena outside every bucket, ascending."""
        spans = sorted((o, o + n) for r in self.buckets for o, n in r)
        out, pos = [], 0
        for a, b in spans:
            if a < pos:
                raise RuntimeError("gradient buckets overlap")
            if a > pos:
                out.append((pos, a - pos))
            pos = b
        if pos < self.flat.numel():
            out.append((pos, self.flat.numel() - pos))
        return out

    def finish(self):
        import torch.distributed as dist
        if self.done:
            return True
        if self.engine is not None:
            self.engine.sync_gradients(1, self.comm)  # rest of the arena, join, 1 / world -- all in the library
            self.done = True
            return True
        if self.comm is not None:
            self.comm.wait_stream(torch.cuda.current_stream(self.flat.device))  # the conditioner's backward wrote the rest
        with self._ctx():
            for off, n in self.uncovered():
                dist.all_reduce(self.flat[off:off + n])
        if self.comm is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm)
        self.flat.mul_(1.0 / dist.get_world_size())
        self.done = True
        return True


def sync_flat_gradients(flat_grads):
    """One all-reduce (mean) over the flat gradient buffer; returns True when a collective ran."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return False
    dist.all_reduce(flat_grads)
    flat_grads.mul_(1.0 / dist.get_world_size())
    return True


class ArenaAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (the reference's optimiser, morphable_diffusion.py:642) on the engine's flat arenas:
    param_groups[0] = the UNet group, [1] / [2] = time_embed / spatial_volume (one learning rate: the reference gives both
    10 lr).  step() = mvd_train_adamw_step (fused HIP kernel per contiguous group range, un-does the loss scale, skips the
    update when a gradient overflowed) + in-place re-pack of the fp16 weights.  LR schedulers act on param_groups as usual."""

    def __init__(self, model, params, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.steps_done = 0
        self.steps_skipped = 0
        self.growth_interval = 2000  # torch.cuda.amp.GradScaler's defaults: x2 after 2000 clean steps, x0.5 on overflow
        self._clean = 0

    @property
    def dynamic_scale(self):
        """loss_scale == 1 (the bfloat16 build: fp32 exponent range) means NO loss scaling: no overflow check, no read-back of the
        "skipped" flag (a stream synchronisation per step), no scale growth -- torch.optim.AdamW's own behaviour.  Derived from
        the model's CURRENT scale at every step: load_state_dict restores the scale, and a resumed run must check (or not) as
        the run that wrote the checkpoint did."""
        return float(self.model.loss_scale) != 1.0

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g0 = self.param_groups[0]
        aux = self.param_groups[1]["lr"] if len(self.param_groups) > 1 else g0["lr"]
        m = self.model
        skipped = m.engine.adamw_step(g0["lr"], aux, self.steps_done + 1, betas=g0["betas"], eps=g0["eps"],
                                      weight_decay=g0["weight_decay"], inv_scale=1.0 / m.loss_scale,
                                      finetune_unet=m.finetune_unet, check=self.dynamic_scale)
        if skipped:  # torch.cuda.amp.GradScaler's rule: back off and try again
            self.steps_skipped += 1
            self._clean = 0
            m.loss_scale = max(1.0, m.loss_scale * 0.5)
        else:
            self.steps_done += 1
            self._clean += 1
            m.global_step = getattr(m, "global_step", 0) + 1  # what Lightning's loop advances per optimiser step
            if self.dynamic_scale and self._clean >= self.growth_interval:
                self._clean = 0
                m.loss_scale = min(m.loss_scale * 2.0, 2.0 ** 24)
        return loss

    def zero_grad(self, set_to_none=False):
        self.model.engine.zero_grad()  # the .grad views stay attached to the (now zero) arena

    # The Adam moments live in the engine's flat arenas, not in torch's per-parameter ``state``: without these two methods a
    # Lightning / torch checkpoint would carry an EMPTY optimiser state and a resumed run would restart with zero moments,
    # bias-correction step 1 and the initial loss scale (torch.optim.AdamW round-trips all of that, morphable_diffusion.py:642).
    def state_dict(self):
        sd = super().state_dict()  # param_groups (learning rates as the scheduler left them); ``state`` is empty
        eng = self.model.engine
        eng.ensure_moments()
        sd["arena"] = {"exp_avg": eng.flat_m.detach().clone(), "exp_avg_sq": eng.flat_v.detach().clone(),
                       "step": self.steps_done, "steps_skipped": self.steps_skipped, "clean_steps": self._clean,
                       "loss_scale": float(self.model.loss_scale), "numel": int(eng.flat_params.numel())}
        return sd

    def load_state_dict(self, state_dict):
        arena = state_dict.get("arena")
        rest = {k: v for k, v in state_dict.items() if k != "arena"}
        eng = self.model.engine
        eng.ensure_moments()
        if arena is None:
            # A checkpoint written by torch.optim.AdamW itself (the reference's optimiser, or this class before it carried the
            # arenas): per-parameter exp_avg / exp_avg_sq in ``state``.  They are adopted when the saved parameters line up with
            # this optimiser's (same count and sizes, group by group); otherwise the moments restart from zero, loudly.
            import warnings
            state = rest.get("state", {})
            mine = [p_ for g_ in self.param_groups for p_ in g_["params"]]
            saved = [i for g_ in rest.get("param_groups", []) for i in g_["params"]]
            ok = len(state) > 0 and len(saved) == len(mine) and all(
                i in state and state[i]["exp_avg"].numel() == p_.numel() for i, p_ in zip(saved, mine))
            super().load_state_dict({"state": {}, "param_groups": rest["param_groups"]} if "param_groups" in rest else rest)
            eng.flat_m.zero_()
            eng.flat_v.zero_()
            self.steps_done = self.steps_skipped = self._clean = 0
            if ok:
                base = eng.flat_params.data_ptr()
                for i, p_ in zip(saved, mine):
                    off = (p_.data_ptr() - base) // 4
                    eng.flat_m[off:off + p_.numel()].copy_(state[i]["exp_avg"].reshape(-1).to(eng.flat_m.device))
                    eng.flat_v[off:off + p_.numel()].copy_(state[i]["exp_avg_sq"].reshape(-1).to(eng.flat_v.device))
                self.steps_done = int(max(float(state[i]["step"]) for i in saved))
                warnings.warn("ArenaAdamW: adopted the per-parameter moments of a torch.optim.AdamW checkpoint "
                              f"(step {self.steps_done}); the loss scale keeps its current value")
            else:
                warnings.warn("ArenaAdamW: the checkpoint carries no 'arena' entry and its per-parameter state does not line up "
                              "with this model's parameters: Adam moments and the bias-correction step restart from zero")
            return
        super().load_state_dict(rest)
        if int(arena["numel"]) != eng.flat_params.numel():
            raise ValueError(f"ArenaAdamW: checkpoint arena has {arena['numel']} elements, the engine {eng.flat_params.numel()}")
        eng.flat_m.copy_(arena["exp_avg"].to(eng.flat_m.device))  # in place: the library holds these pointers
        eng.flat_v.copy_(arena["exp_avg_sq"].to(eng.flat_v.device))
        self.steps_done = int(arena["step"])
        self.steps_skipped = int(arena.get("steps_skipped", 0))
        self._clean = int(arena.get("clean_steps", 0))
        self.model.loss_scale = float(arena["loss_scale"])  # dynamic_scale follows it (property)


class LambdaLinearScheduler:
    """ldm/lr_scheduler.py:59-98 (LambdaWarmUpCosineScheduler2.find_in_interval + LambdaLinearScheduler.schedule): the
    learning-rate multiplier of configs/facescape.yaml:17-24 -- linear warm-up f_start -> f_max, then linear decay to f_min
    over the cycle."""

    def __init__(self, warm_up_steps, f_min, f_max, f_start, cycle_lengths, verbosity_interval=0):
        assert len(warm_up_steps) == len(f_min) == len(f_max) == len(f_start) == len(cycle_lengths)
        self.lr_warm_up_steps, self.f_start, self.f_min, self.f_max = warm_up_steps, f_start, f_min, f_max
        self.cycle_lengths = cycle_lengths
        self.cum_cycles = np.cumsum([0] + list(cycle_lengths))
        self.last_f = 0.0

    def find_in_interval(self, n):
        interval = 0
        for cl in self.cum_cycles[1:]:
            if n <= cl:
                return interval
            interval += 1

    def schedule(self, n, **kwargs):
        cycle = self.find_in_interval(n)
        n = n - self.cum_cycles[cycle]
        if n < self.lr_warm_up_steps[cycle]:
            f = (self.f_max[cycle] - self.f_start[cycle]) / self.lr_warm_up_steps[cycle] * n + self.f_start[cycle]
        else:
            f = self.f_min[cycle] + (self.f_max[cycle] - self.f_min[cycle]) * (self.cycle_lengths[cycle] - n) / self.cycle_lengths[cycle]
        self.last_f = f
        return f

    __call__ = schedule


class DiagonalGaussianDistribution:
    """ldm/modules/distributions/distributions.py:24-37 (the part encode_first_stage uses): host code on the moments
    the engine's encoder returns, so the random draw is torch's own."""

    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self):
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean


class SyncDDIMSampler:
    """morphable_diffusion.py:648-776.  With torch.distributed initialised and ``shard_views=True`` the N views
    are partitioned over the ranks (contiguous slices); the only per-step exchange is ONE collective on the per-vertex
    features:

    * ``exchange="all_gather"`` (default): all-gather of the per-view features [N,Nv,16] (5.1 MB at N=16, FLAME), then every
      rank sums the views in index order -- the sharded step is BIT-IDENTICAL to the single-GPU step;
    * ``exchange="all_reduce"``: all-reduce of each rank's share of the view mean [Nv,16] (321 KB); the fp32 summation order
      then depends on the rank count (differences ~1e-7 relative).

    The exchange, the view fusion, the sparse voxel CNN and the lattice gather are enqueued on a separate communication
    stream: the UNet's full-resolution input blocks need none of it, so ``mvd_denoise_views`` starts on the caller's stream
    at once and only its frustum stage waits for the volume (``mvd_set_volume_ready_event``).  ``overlap=False`` (or
    MVD_NO_COMM_OVERLAP=1) keeps everything on the caller's stream."""

    def __init__(self, model, ddim_num_steps, ddim_discretize="uniform", ddim_eta=1.0, latent_size=32,
                 shard_views=False, exchange="all_gather", overlap=True):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.latent_size = latent_size
        self.schedule = DDIMSchedule(ddim_num_steps, ddim_eta, self.ddpm_num_timesteps)
        if ddim_discretize != "uniform":
            raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discretize}"')
        self.ddim_timesteps = self.schedule.ddim_timesteps
        self.ddim_alphas, self.ddim_alphas_prev = self.schedule.ddim_alphas, self.schedule.ddim_alphas_prev
        self.ddim_sigmas = self.schedule.ddim_sigmas
        self.ddim_sqrt_one_minus_alphas = self.schedule.ddim_sqrt_one_minus_alphas
        self.eta = ddim_eta
        self.shard_views = shard_views
        if exchange not in ("all_gather", "all_reduce"):
            raise ValueError(f"unknown exchange {exchange!r}")
        self.exchange = exchange
        import os
        self.overlap = overlap and not os.environ.get("MVD_NO_COMM_OVERLAP")
        # B > 1 (the eval driver): "batched" = all samples share each UNet pass (batch 2 * B * views with guidance), like the
        # reference's own batching; "loop" = samples one by one (each then equals its single-sample run bit for bit)
        self.sample_batching = os.environ.get("MVD_SAMPLE_BATCHING", "batched")
        self.simulate_world = 0  # timing aid (bench.py --simulate-gpus): run ONE rank's share without a process group
        self._comm = None  # (stream, event after the vertex features, event after the volume), created on first use
        self._bufs = {}    # persistent exchange buffers: never handed back to the allocator while the side stream uses them

    # -- distributed helpers -------------------------------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if self.simulate_world:
            return 0, self.simulate_world
        if self.shard_views and dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def use_library_exchange(self):
        """Moves the per-step all-gather behind the C ABI: the library creates its own RCCL communicator over the ranks of the
        default process group (Engine.comm_init) and `mvd_exchange_view_features` enqueues the ncclAllGather on the
        communication stream.  Collective: every rank calls it once, after init_process_group."""
        rank, world = self._world()
        if world > 1 and not self.simulate_world:
            self.model.engine.comm_init(rank, world)

    def view_range(self, N):
        rank, world = self._world()
        if N % world:
            raise ValueError(f"{N} views do not shard evenly over {world} ranks")
        per = N // world
        return rank * per, (rank + 1) * per

    def denoise_apply(self, x_target_noisy, input_info, clip_embed, time_steps, index, unconditional_scale,
                      batch_view_num=1, is_step0=False, batch=None, noise=None, return_eps=False, host_steps=None):
        """One multi-view denoising step.  x_target_noisy [B,N_local,4,H,W] holds this rank's views
        (all N when not sharded).  ``noise``: optional explicit N(0,1) draw [B,N_local,4,H,W]; default is a
        fresh torch.randn_like as in the reference (:695-697).  ``return_eps``: also return the guided noise prediction
        e_t (:736), the tensor denoise_apply_impl consumes.  ``host_steps``: the values of ``time_steps`` as python ints
        (the sampling loop knows them), so that no device->host read happens on the step path."""
        import torch.distributed as dist
        m = self.model
        eng = m.engine
        x_input = input_info["x"]
        B, NL, C, H, W = x_target_noisy.shape
        N = m.view_num
        rank, world = self._world()
        lo = rank * NL if world > 1 else 0
        if world == 1 and NL != N:
            raise ValueError("x_target_noisy must hold all views when the sampler is not sharded")
        v_embed = self._v_embed(batch, x_target_noisy.device)
        t_embed = m.embed_time(time_steps)
        coef = self.schedule.coefficients(index)
        if noise is None and not is_step0:
            noise = torch.randn_like(x_target_noisy)
        out = torch.empty_like(x_target_noisy, memory_format=torch.contiguous_format)
        eps_out = torch.empty_like(out) if return_eps else None
        # view indices of this rank, resident on the device (a host tensor here would be a pageable host-to-device copy per
        # engine call: a host synchronisation on the step path, and not capturable in a hipGraph)
        key = (lo, NL, str(x_target_noisy.device))
        if getattr(self, "_idx_key", None) != key:
            self._idx_key = key
            self._idx_dev = torch.arange(lo, lo + NL, dtype=torch.int32, device=x_target_noisy.device)
        local_idx = self._idx_dev
        if host_steps is None:  # one read for the whole call (none at all when the caller passes host_steps)
            host_steps = [int(v) for v in time_steps.tolist()]
        from .engine import MAX_SAMPLE_SLOTS
        if B > 1 and self.sample_batching == "batched" and B <= MAX_SAMPLE_SLOTS:
            try:
                return self._denoise_apply_batched(x_target_noisy, x_input, clip_embed, host_steps, t_embed, v_embed, local_idx, lo, NL,
                                                   rank, world, N, unconditional_scale, batch_view_num, is_step0, batch, noise, coef,
                                                   out, eps_out, return_eps)
            except Exception as e:  # MvdError from the library
                if "workspace" not in str(e):
                    raise
                # One UNet pass of 2 * B * views plus B slot volumes needs more workspace than the per-sample loop an eval batch
                # may have been sized for: fall back to that loop (each sample then equals its single-sample run) and stay there
                import warnings
                warnings.warn(f"batched sampling of {B} samples does not fit the engine's workspace ({e}); "
                              "falling back to the per-sample loop (MVD_SAMPLE_BATCHING=loop)")
                self.sample_batching = "loop"
                m.spatial_volume.invalidate()
        for bi in range(B):
            m.spatial_volume._set_sample(batch, bi)
            self._build_volume(x_target_noisy[bi], t_embed[bi], v_embed[bi, lo:lo + NL], local_idx, rank, world, N)
            for ni in range(0, NL, batch_view_num):
                sl = slice(ni, min(NL, ni + batch_view_num))
                idx = local_idx[sl]
                # x_prev / eps land in the caller-visible tensors straight from the C call (contiguous slices of `out`), and the
                # view embeddings are a slice, not a gather: no torch kernel on the step path
                o_v, e_v = out[bi, sl], (eps_out[bi, sl] if return_eps else None)
                r = eng.denoise_views(
                    x_target_noisy[bi, sl], x_input[bi], clip_embed[bi].reshape(-1), host_steps[bi], t_embed[bi],
                    v_embed[bi, lo + sl.start:lo + sl.stop], idx, float(unconditional_scale),
                    None if is_step0 else noise[bi, sl], coef, want_eps=return_eps, out=o_v, eps_out=e_v)
                xp, ep = r if return_eps else (r, None)
                if xp.data_ptr() != o_v.data_ptr():  # an engine that returned its own tensors (CPU stand-ins)
                    o_v.copy_(xp)
                if return_eps and ep.data_ptr() != e_v.data_ptr():
                    e_v.copy_(ep)
        return (out, eps_out) if return_eps else out

    def _denoise_apply_batched(self, x_target_noisy, x_input, clip_embed, host_steps, t_embed, v_embed, local_idx, lo, NL, rank, world,
                               N, unconditional_scale, batch_view_num, is_step0, batch, noise, coef, out, eps_out, return_eps):
        from .engine import MAX_SAMPLE_SLOTS
        m, eng = self.model, self.model.engine
        B = x_target_noisy.shape[0]
        # All samples share each UNet pass (batch 2 * B * views with guidance), as the reference's own batching does
        # (eval/generate_all_facescape.py:106-108,128-129); every sample's volume is built first, into its slot.
        for bi in range(B):
            m.spatial_volume._set_sample(batch, bi)
            self._build_volume(x_target_noisy[bi], t_embed[bi], v_embed[bi, lo:lo + NL], local_idx, rank, world, N,
                               tag=f".{bi}" if bi else "")
        slots = [bi % MAX_SAMPLE_SLOTS for bi in range(B)]
        for ni in range(0, NL, batch_view_num):
            sl = slice(ni, min(NL, ni + batch_view_num))
            idx = local_idx[sl]
            r = eng.denoise_views_batch(
                slots, x_target_noisy[:, sl], x_input, clip_embed.reshape(B, -1), host_steps, t_embed,
                v_embed[:, lo + sl.start:lo + sl.stop], idx, float(unconditional_scale),
                None if is_step0 else noise[:, sl], coef, want_eps=return_eps)
            if return_eps:
                out[:, sl], eps_out[:, sl] = r
            else:
                out[:, sl] = r
        return (out, eps_out) if return_eps else out

    def _v_embed(self, batch, device):
        """get_viewpoint_embedding(batch) on `device`, cached on the CONTENT identity of the four angle tensors (storage address,
        shape, in-place version counter; references held so an address cannot be recycled under a live key): the embedding is
        constant over the 50 steps of a trajectory, and its ~12 torch kernels were on every step."""
        ts = tuple(batch[k] for k in ("target_elevation", "input_elevation", "target_azimuth", "input_azimuth"))
        key = tuple((t.data_ptr(), tuple(t.shape), t._version, str(t.device)) for t in ts) + (str(device),)
        c = getattr(self, "_ve_cache", None)
        if c is None or c[0] != key:
            c = (key, self.model.get_viewpoint_embedding(batch).to(device).contiguous(), ts)
            self._ve_cache = c
        return c[1]

    def _buf(self, name, shape, device):
        t = self._bufs.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != device:
            t = torch.empty(shape, device=device, dtype=torch.float32)
            self._bufs[name] = t
        return t

    def _build_volume(self, x_local, t_embed, v_embed_local, local_idx, rank, world, N, tag=""):
        """2-D encoder + vertex gather for this rank's views (caller's stream), then exchange -> view fusion -> sparse voxel
        CNN -> lattice gather on the communication stream; leaves the 32^3 volume in the engine (in the ACTIVE sample slot),
        guarded by an event.  ``tag``: suffix of the persistent exchange buffers -- consecutive builds of different samples
        (batched denoise_apply) must not share them, the communication stream still reads the previous sample's."""
        import torch.distributed as dist
        eng = self.model.engine
        dev = x_local.device
        real = world > 1 and not self.simulate_world
        if real:  # under a launcher the designed path is the library's communicator (falls back to c10d, logged)
            ensure_library_comm(eng, dev)
        side = self.overlap and dev.type == "cuda"
        # nn.Module semantics, as in the reference: in train mode the sparse CNN's BatchNorm layers use batch statistics
        # (construct_spatial_volume, morphable_diffusion.py:253-254); callers that sample call .eval() (generate_face.py:77)
        bn_train = bool(getattr(self.model.spatial_volume, "training", False))
        if side and self._comm is None:
            self._comm = (torch.cuda.Stream(device=dev), torch.cuda.Event(), torch.cuda.Event())
        NL = x_local.shape[0]
        if self.exchange == "all_gather":
            Nv = eng.num_vertices
            # The whole head of the step -- 2-D encoder + vertex gather too -- goes to the communication stream when the engine's
            # vertex-feature stage touches no shared workspace (mvd_vertex_features_stream_safe): the UNet's input blocks need none
            # of it, so the caller's stream starts the UNet at once (round 6: ~0.1 ms of small kernels per step left the
            # critical path).  MVD_HEAD_ON_MAIN=1: the round-5 placement.
            import os
            head_side = (side and dev.type == "cuda" and not os.environ.get("MVD_HEAD_ON_MAIN")
                         and getattr(eng, "vertex_features_stream_safe", lambda: False)())
            if dev.type == "cuda":
                vf_all = self._buf("vf_all" + tag, (N, Nv, 16), dev)
                lo = rank * NL if world > 1 else 0
                vf_loc = vf_all[lo:lo + NL] if not real else self._buf("vf_loc" + tag, (NL, Nv, 16), dev)
                if not head_side:
                    eng.vertex_view_features(x_local, t_embed, v_embed_local, local_idx, out=vf_loc)
            else:  # CPU stand-ins of the engine (tests)
                vf_loc = eng.vertex_view_features(x_local, t_embed, v_embed_local, local_idx)
                vf_all = vf_loc if world == 1 else torch.empty((N,) + tuple(vf_loc.shape[1:]), dtype=vf_loc.dtype)
            fused_buf = self._buf("fused" + tag, (Nv, 16), dev) if dev.type == "cuda" else None

            def tail():
                if head_side:  # on the communication stream, behind ev_in (x_local and t_embed are final there)
                    for t_ in (x_local, t_embed, v_embed_local):  # allocated on the caller's stream, read on this one
                        t_.record_stream(torch.cuda.current_stream(dev))
                    eng.vertex_view_features(x_local, t_embed, v_embed_local, local_idx, out=vf_loc)
                if real and getattr(eng, "comm_world", 0) == world:
                    # the library's own communicator (mvd_comm_init): ncclAllGather enqueued by the C ABI on this stream, no
                    # torch.distributed on the step path (SyncDDIMSampler.use_library_exchange)
                    eng.exchange_view_features(vf_loc, vf_all)
                elif real:
                    dist.all_gather_into_tensor(vf_all, vf_loc)  # RCCL over xGMI; rank r's slice lands at views [r*NL, ...)
                elif world > 1:  # --simulate-gpus: stand in for the other ranks' slices
                    for r in range(1, world):
                        vf_all[r * NL:(r + 1) * NL].copy_(vf_all[:NL])
                eng.volume_from_fused(eng.fuse_vertex_features(vf_all, out=fused_buf), want_output=False, train=bn_train)
        else:
            fused = eng.vertex_features(x_local, t_embed, v_embed_local, local_idx, add_bias=(rank == 0))
            if dev.type == "cuda":  # persistent: the communication stream reads it after this call returns
                fused = self._buf("fused" + tag, tuple(fused.shape), dev).copy_(fused)

            def tail():
                if real:
                    dist.all_reduce(fused)  # Nv*16 fp32, latency-bound
                eng.volume_from_fused(fused, want_output=False, train=bn_train)
        if not side:
            if dev.type == "cuda":
                eng.set_volume_ready_event(None)
            tail()
            return
        comm, ev_in, ev_vol = self._comm
        ev_in.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(comm):
            comm.wait_event(ev_in)  # after the vertex features AND after every earlier reader of the volume on that stream
            tail()
            ev_vol.record(comm)
        eng.set_volume_ready_event(ev_vol)

    def sample(self, input_info, clip_embed, unconditional_scale=1.0, log_every_t=50, batch_view_num=1, batch=None,
               generator=None, return_eps=False):
        """Returns (x [B,N,4,h,w], {'x_inter': [...]}) like the reference (:742-776); with view sharding every rank
        returns the gathered full tensor.  RNG: x_T is drawn first, then one N(0,1) tensor per step except the last
        (index 0), in the order the reference consumes its stream.  ``generator``: a torch.Generator; a CPU generator
        makes the draws on the host (bit-identical to the reference run on CPU with the same seed), a device generator or
        None draws on the device.  ``return_eps``: adds 'eps' (one [B,N,4,h,w] per step) to the returned dict."""
        print(f"unconditional scale {unconditional_scale:.1f}")
        C, H, W = 4, self.latent_size, self.latent_size
        B = clip_embed.shape[0]
        N = self.model.view_num
        device = self.model.device
        rank, world = self._world()
        lo, hi = self.view_range(N)
        self.model.spatial_volume.invalidate()  # every run re-reads the batch at least once
        on_host = generator is not None and generator.device.type == "cpu"

        def draw():
            # full-size draws on every rank (same generator state) then slice: matches the single-GPU RNG stream
            if on_host:
                full = torch.randn([B, N, C, H, W], generator=generator)
                return full[:, lo:hi].contiguous().to(device)
            return torch.randn([B, N, C, H, W], device=device, generator=generator)[:, lo:hi].contiguous()

        x = draw()
        intermediates = {"x_inter": []}
        if return_eps:
            intermediates["eps"] = []
        time_range = np.flip(self.ddim_timesteps)
        total_steps = self.ddim_timesteps.shape[0]
        with torch.no_grad():
            for i, step in enumerate(time_range):
                index = total_steps - i - 1
                time_steps = self._time_steps(B, int(step), device)
                noise = draw() if index != 0 else None
                x = self.denoise_apply(x, input_info, clip_embed, time_steps, index, unconditional_scale,
                                       batch_view_num=batch_view_num, is_step0=index == 0, batch=batch, noise=noise,
                                       return_eps=return_eps, host_steps=[int(step)] * B)
                if return_eps:
                    x, e = x
                    intermediates["eps"].append(self._gather(e, world))
                if index % log_every_t == 0 or index == total_steps - 1:
                    intermediates["x_inter"].append(self._gather(x, world))
        return self._gather(x, world), intermediates

    def _time_steps(self, B, step, device):
        """[B] int64 tensor of one DDIM time step, kept per step value (50 small tensors per sampler): no fill kernel per step."""
        c = self.__dict__.setdefault("_ts_cache", {})
        key = (B, step, str(device))
        if key not in c:
            c[key] = torch.full((B,), step, device=device, dtype=torch.long)
        return c[key]

    def _gather(self, x, world):
        if world == 1:
            return x
        import torch.distributed as dist
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous())
        return torch.cat(parts, 1)
