"""Generate tests/golden/*.npz by running the REFERENCE's own Python (imported from /root/reference, CPU,
build container only) on the seeded inputs of tests/golden_inputs.py.  The fixtures are data only: inputs
are regenerated from seeds by the tests, expected outputs are stored (full when small, strided sample +
checksums when large).  Re-run:  python tools/make_goldens.py [--skip-full]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import ref_import  # noqa: E402
from morphablediffusion_amd import synthetic  # noqa: E402
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest, unet_manifest  # noqa: E402
from tests import golden_inputs as gi  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, packs, extra=None):
    flat = gi.flatten_packs(packs)
    if extra:
        flat.update(extra)
    np.savez_compressed(os.path.join(OUT, name), **flat)
    print("wrote", name, f"{os.path.getsize(os.path.join(OUT, name)) / 1024:.0f} KiB")


def cfg_kwargs(cfg: UNetConfig):
    return dict(volume_dims=list(cfg.volume_dims), image_size=cfg.image_size, in_channels=cfg.in_channels,
                out_channels=cfg.out_channels, model_channels=cfg.model_channels,
                attention_resolutions=list(cfg.attention_resolutions), num_res_blocks=cfg.num_res_blocks,
                channel_mult=list(cfg.channel_mult), num_heads=cfg.num_heads, use_spatial_transformer=True,
                transformer_depth=1, context_dim=cfg.context_dim, use_checkpoint=False, legacy=False)


def load_unet(ns, cfg, style="init"):
    m = ns.attention.DepthWiseAttention(**cfg_kwargs(cfg)).eval()
    W = gi.unet_weights(cfg, style)
    sd = {k[len("model.diffusion_model."):]: v for k, v in W.items()}
    ref_keys = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(ref_keys) == set(sd), (set(ref_keys) ^ set(sd))
    for k in sd:
        assert ref_keys[k] == tuple(sd[k].shape), (k, ref_keys[k], sd[k].shape)
    m.load_state_dict(sd, strict=True)
    return m


def gold_basic(ns):
    packs = {}
    for dim in (256, 320):
        t = torch.tensor([1, 481, 981], dtype=torch.long)
        packs[f"temb{dim}"] = gi.pack(ns.dutil.timestep_embedding(t, dim))
    save("basic.npz", packs)


def gold_unet_small(ns):
    cfg = gi.SMALL_UNET
    m = load_unet(ns, cfg)
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2)
    packs = {}
    with torch.no_grad():
        packs["unet_out"] = gi.pack(m(x, t, ctx, source_dict=sd))
        temb = ns.dutil.timestep_embedding(t, cfg.model_channels)
        emb = m.time_embed(temb)
        packs["emb"] = gi.pack(emb)
        g = torch.Generator().manual_seed(5)
        h64 = torch.randn(2, 64, 32, 32, generator=g)
        h128 = torch.randn(2, 128, 16, 16, generator=g)
        h256 = torch.randn(2, 256, 8, 8, generator=g)
        packs["res_same"] = gi.pack(m.input_blocks[1][0](h64, emb))  # 64 -> 64
        packs["res_skip"] = gi.pack(m.input_blocks[4][0](h64[:, :, ::2, ::2].contiguous(), emb))  # 64 -> 128 @16
        packs["st32"] = gi.pack(m.input_blocks[1][1](h64, ctx))
        packs["st16"] = gi.pack(m.input_blocks[4][1](h128, ctx))
        packs["st8"] = gi.pack(m.input_blocks[7][1](h256, ctx))
        packs["down"] = gi.pack(m.input_blocks[3][0](h64))
        packs["up"] = gi.pack(m.output_blocks[8][2](h128))  # Upsample 128ch 16->32
        packs["cond8"] = gi.pack(m.output_conditions[8](h64, context=sd[32]))
        packs["cond3"] = gi.pack(m.output_conditions[3](h128, context=sd[16]))
        packs["cond_mid"] = gi.pack(m.middle_conditions(h256[:, :, ::2, ::2].contiguous(), context=sd[4]))
    save("unet_small.npz", packs)


def gold_unet_full(ns):
    cfg = gi.FULL_UNET
    t0 = time.time()
    m = load_unet(ns, cfg)
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2)
    with torch.no_grad():
        out = m(x, t, ctx, source_dict=sd)
    print("full unet", time.time() - t0, "s")
    save("unet_full.npz", {"unet_out": gi.pack(out)})


def gold_unet_trained(ns):
    """UNet eps on the second weight set (trained-checkpoint-like statistics), reduced and full width."""
    for name, cfg in (("unet_small_trained.npz", gi.SMALL_UNET), ("unet_full_trained.npz", gi.FULL_UNET)):
        m = load_unet(ns, cfg, "trained")
        x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2, seed=13)
        with torch.no_grad():
            out = m(x, t, ctx, source_dict=sd)
        print(name, "std", float(out.std()), "absmax", float(out.abs().max()))
        assert torch.isfinite(out).all() and out.std() > 1e-3
        save(name, {"unet_out": gi.pack(out)})


def build_full_model(ns, ucfg, vcfg, N, style="init"):
    md = ns.md
    model = md.SyncMultiviewDiffusion(
        unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": cfg_kwargs(ucfg)},
        scheduler_config=None, finetune_unet=False, projection=vcfg.projection, use_spatial_volume=False,
        view_num=N, image_size=vcfg.input_image_size, cfg_scale=2.0, output_num=8, batch_view_num=4, drop_conditions=False,
        clip_image_encoder_path="", sample_type="ddim", sample_steps=50, target_elevation=0).eval()
    model.spatial_volume.smpl_feature_extractor.num_views = N  # gotcha G3: hard-wired 16 in the reference
    if vcfg.input_image_size != 256:  # gotcha G4: the image size is not forwarded to SpatialVolumeNet
        model.spatial_volume.input_image_size = vcfg.input_image_size
        model.spatial_volume.frustum_volume_size = vcfg.input_image_size // 8
    W = gi.full_weights(ucfg, vcfg, style)
    ref_sd = model.state_dict()
    hot = {k: tuple(v.shape) for k, v in ref_sd.items()
           if k.startswith(("model.diffusion_model.", "spatial_volume.", "time_embed."))
           and not k.endswith("num_batches_tracked")}
    assert set(hot) == set(W), sorted(set(hot) ^ set(W))[:10]
    for k in W:
        assert hot[k] == tuple(W[k].shape), (k, hot[k], W[k].shape)
    missing, unexpected = model.load_state_dict(W, strict=False)
    assert not unexpected, unexpected
    return model, hot


def record_eps(sampler, sink):
    """Wraps the reference's denoise_apply_impl (morphable_diffusion.py:675-698) so that the noise prediction it receives
    -- eps after classifier-free guidance, BEFORE the DDIM update dilutes it -- is stored next to x_prev."""
    impl = sampler.denoise_apply_impl

    def wrapped(x_target_noisy, index, noise_pred, is_step0=False):
        sink.append(noise_pred.detach().clone())
        return impl(x_target_noisy, index, noise_pred, is_step0)

    sampler.denoise_apply_impl = wrapped


def gold_step(ns, name, ucfg, N, projection, index, with_noise, nverts, bvn, stages=False, image_size=256,
              radii=(0.22, 0.28, 0.25), style="init", frustum_views=0):
    vcfg = VolumeConfig(num_views=N, projection=projection, input_image_size=image_size)
    model, hot = build_full_model(ns, ucfg, vcfg, N, style)
    batch = synthetic.make_batch(N, projection, nverts, mesh_seed=1, image_size=image_size, radii=radii)
    x_T, x_in, clip = synthetic.make_latents(N, image_size // 8, seed=6033)
    sampler = model.sampler
    step = int(sampler.ddim_timesteps[index])
    ts = torch.full((1,), step, dtype=torch.long)
    packs = {}
    extra = {"index": index, "step": step, "N": N, "nverts_in": nverts, "bvn": bvn,
             "with_noise": int(with_noise), "noise_seed": 99, "image_size": image_size, "radii": np.array(radii)}
    eps_sink = []
    record_eps(sampler, eps_sink)
    with torch.no_grad():
        t0 = time.time()
        torch.manual_seed(99)
        out = sampler.denoise_apply(x_T, {"x": x_in, "elevation": batch["input_elevation"][:, 0]}, clip, ts, index,
                                    2.0, batch_view_num=bvn, is_step0=not with_noise, batch=batch)
        print(name, "denoise_apply", time.time() - t0, "s")
        packs["x_prev"] = gi.pack(out)
        packs["eps"] = gi.pack(eps_sink[0])
        if frustum_views and not stages:  # per-view frustum volumes through the real conditioner (full width)
            v_embed = model.get_viewpoint_embedding(batch)
            t_embed = model.embed_time(ts)
            sv_ = model.spatial_volume.construct_spatial_volume(x_T, t_embed, v_embed, batch)
            packs["spatial_volume"] = gi.pack(sv_)
            idx = torch.tensor([[0, N - 1][:frustum_views]])
            fd, _ = model.spatial_volume.construct_view_frustum_volume(sv_, t_embed, v_embed, idx, batch)
            for k, v in fd.items():
                packs[f"frustum_{k}"] = gi.pack(v)
            extra["frustum_idx"] = idx[0].numpy()
        if stages:
            v_embed = model.get_viewpoint_embedding(batch)
            t_embed = model.embed_time(ts)
            packs["t_embed"] = gi.pack(t_embed)
            packs["v_embed"] = gi.pack(v_embed)
            sv = model.spatial_volume
            f0 = sv.target_encoder(x_T[:, 0], t_embed, v_embed[:, 0])
            packs["enc_view0"] = gi.pack(f0)
            V = 32
            lin = torch.linspace(-0.5, 0.5, V)
            verts = torch.stack(torch.meshgrid(lin, lin, lin), -1).reshape(1, V ** 3, 3)[:, :, (2, 1, 0)]
            verts = verts.view(1, V, V, V, 3).permute(0, 4, 1, 2, 3)
            for vi in (0, N - 1):
                c = ns.utils.get_warp_coordinates(verts, 32, 256, batch["target_K"][:, vi], batch["target_RT"][:, vi],
                                                  projection=projection)
                packs[f"warp_view{vi}"] = gi.pack(c)
            spatial_volume = sv.construct_spatial_volume(x_T, t_embed, v_embed, batch)
            packs["spatial_volume"] = gi.pack(spatial_volume)
            # intermediate: per-view vertex features and fused features, via the same calls the reference makes
            feats = []
            for ni in range(N):
                x_ = sv.target_encoder(x_T[:, ni], t_embed, v_embed[:, ni])
                cs = ns.utils.get_warp_coordinates(verts, 32, 256, batch["target_K"][:, ni], batch["target_RT"][:, ni],
                                                   projection=projection).view(1, V, V * V, 2)
                u = torch.nn.functional.grid_sample(x_, cs, mode="bilinear", padding_mode="zeros", align_corners=True)
                feats.append(u.view(1, -1, V, V, V))
            feats = torch.stack(feats, 1).view(1, -1, V, V, V)
            Nv = batch["vertices"].shape[1]
            grid = (batch["vertices"].unsqueeze(2).unsqueeze(2) / 0.5).unsqueeze(1).repeat(1, N, 1, 1, 1, 1).reshape(N, Nv, 1, 1, 3)
            vf = torch.nn.functional.grid_sample(feats.reshape(N, -1, V, V, V), grid, mode="bilinear", padding_mode="zeros",
                                                 align_corners=True)[:, :, :, 0, 0].reshape(1, N, -1, Nv)
            packs["vertex_feats"] = gi.pack(vf)
            fused = sv.smpl_feature_extractor(vf).permute(0, 2, 1)
            packs["fused"] = gi.pack(fused)
            from spconv.pytorch.core import SparseConvTensor
            coord = torch.cat([torch.zeros(Nv, 1, dtype=torch.int32), batch["coord"][0]], 1).int()
            dense = sv.xyzc_net(SparseConvTensor(fused[0], coord, batch["out_sh"][0].tolist(), 1))
            packs["sparse_dense"] = gi.pack(dense)
            idx = torch.arange(0, min(2, N))[None]
            fd, _ = sv.construct_view_frustum_volume(spatial_volume, t_embed, v_embed, idx, batch)
            for k, v in fd.items():
                packs[f"frustum_{k}"] = gi.pack(v)
            # frustum geometry
            poses = batch["target_RT"][0, :2]
            Ks = batch["target_K"][0, :2]
            camp = -(poses[:, :, :3].transpose(1, 2) @ poses[:, :, 3:])[:, :, 0]
            dist = camp.norm(dim=-1).reshape(2, 1)
            near = torch.ones(2, 1, 32, 32) * dist[..., None, None] - 0.86603
            far = torch.ones(2, 1, 32, 32) * dist[..., None, None] + 0.86603
            xyz, _ = ns.utils.create_target_volume(48, 32, 256, poses, Ks, near, far, projection)
            packs["frustum_xyz"] = gi.pack(xyz)
            extra["nverts"] = Nv
    save(name, packs, extra)
    return hot


def gold_vae(ns):
    """First-stage decoder: the reference's AutoencoderKL.decode on seeded weights and latents;
    reduced width (ch=32, two latents) and the real width (ch=128, one latent)."""
    import importlib
    from morphablediffusion_amd.spec import VaeConfig, vae_decoder_manifest, vae_encoder_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    ae = importlib.import_module("ldm.models.autoencoder")
    for name, cfg, B in (("vae_small.npz", VaeConfig(ch=32), 2), ("vae_full.npz", VaeConfig(), 1)):
        dd = dict(double_z=True, z_channels=cfg.z_channels, resolution=256, in_channels=3, out_ch=cfg.out_ch, ch=cfg.ch,
                  ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks, attn_resolutions=[], dropout=0.0)
        model = ae.AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=cfg.embed_dim).eval()
        man = dict(vae_decoder_manifest(cfg))
        man.update(vae_encoder_manifest(cfg))
        W = seeded_state_dict(man, gi.WEIGHT_SEED)
        sd = model.state_dict()
        dec = {k: tuple(v.shape) for k, v in sd.items()
               if k.startswith(("decoder.", "post_quant_conv.", "encoder.", "quant_conv."))}
        assert {("first_stage_model." + k): v for k, v in dec.items()} == {k: tuple(v) for k, v in man.items()}
        missing, unexpected = model.load_state_dict({k[len("first_stage_model."):]: v for k, v in W.items()}, strict=False)
        assert not unexpected, unexpected
        g = torch.Generator().manual_seed(31)
        z = torch.randn(B, cfg.embed_dim, 32, 32, generator=g) * 4.0  # latents / 0.18215 have a std of about 4-5
        with torch.no_grad():
            t0 = time.time()
            out = model.decode(z)
            print(name, "decode", time.time() - t0, "s", tuple(out.shape))
        x = torch.rand(B, 3, 256, 256, generator=g) * 2.0 - 1.0  # images in [-1, 1]
        with torch.no_grad():
            mom = model.encode(x).parameters
        save(name, {"out": gi.pack(out), "moments": gi.pack(mom)}, {"B": B, "ch": cfg.ch})


def gold_clip(ns=None):
    """CLIP image embedding.  openai/CLIP and kornia are not importable here, so the goldens come
    from transformers' CLIPVisionModelWithProjection -- an independent implementation of the same published model --
    loaded with the seeded weights under the openai key names (mapping below); the input is the oracle's preprocess of a
    seeded 256^2 image (F.interpolate bicubic, what kornia's resize calls)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from morphablediffusion_amd.spec import ClipConfig, clip_manifest, CLIP_PREFIX as P
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import clip_oracle as CO
    for name, cfg, B in (("clip_small.npz", ClipConfig(width=128, layers=2, heads=2, embed=64), 2),
                         ("clip_full.npz", ClipConfig(), 1)):
        W = seeded_state_dict(clip_manifest(cfg), gi.WEIGHT_SEED)
        hc = CLIPVisionConfig(hidden_size=cfg.width, intermediate_size=4 * cfg.width, projection_dim=cfg.embed,
                              num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, image_size=cfg.image,
                              patch_size=cfg.patch, hidden_act="quick_gelu", layer_norm_eps=1e-5, attention_dropout=0.0)
        model = CLIPVisionModelWithProjection(hc).eval()
        w = cfg.width
        sd = {"vision_model.embeddings.class_embedding": W[P + "class_embedding"],
              "vision_model.embeddings.patch_embedding.weight": W[P + "conv1.weight"],
              "vision_model.embeddings.position_embedding.weight": W[P + "positional_embedding"],
              "visual_projection.weight": W[P + "proj"].t().contiguous()}
        for a, b in (("pre_layrnorm", "ln_pre"), ("post_layernorm", "ln_post")):
            for x in ("weight", "bias"):
                sd[f"vision_model.{a}.{x}"] = W[f"{P}{b}.{x}"]
        for i in range(cfg.layers):
            h, o = f"vision_model.encoder.layers.{i}.", f"{P}transformer.resblocks.{i}."
            for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                sd[h + f"self_attn.{n}.weight"] = W[o + "attn.in_proj_weight"][j * w:(j + 1) * w]
                sd[h + f"self_attn.{n}.bias"] = W[o + "attn.in_proj_bias"][j * w:(j + 1) * w]
            for a, b in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                         ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
                for x in ("weight", "bias"):
                    sd[h + a + "." + x] = W[o + b + "." + x]
        missing, unexpected = model.load_state_dict(sd, strict=False)
        missing = [k for k in missing if "position_ids" not in k]
        assert not missing and not unexpected, (missing, unexpected)
        g = torch.Generator().manual_seed(47)
        x = torch.rand(B, 3, 256, 256, generator=g) * 2.0 - 1.0
        with torch.no_grad():
            t0 = time.time()
            pix = CO.preprocess(x, cfg.image)
            out = model(pixel_values=pix).image_embeds
            print(name, time.time() - t0, "s", tuple(out.shape), float(out.abs().max()))
        save(name, {"embed": gi.pack(out.unsqueeze(1)), "pixels": gi.pack(pix)}, {"B": B, "width": cfg.width})


def gold_traj(ns):
    """Multi-step trajectory (a1: SyncDDIMSampler.sample, morphable_diffusion.py:742-776): the reference's own loop on a
    4- and a 5-step DDIM schedule at reduced width -- loop order, ``index``, time-step table, and the consumption of the
    global RNG stream (x_T first, then one randn_like per step except the last).  Every step's x and eps are stored."""
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    model, _ = build_full_model(ns, ucfg, vcfg, N)
    batch = synthetic.make_batch(N, "perspective", 600, mesh_seed=1)
    _, x_in, clip = synthetic.make_latents(N, 32, seed=6033)
    packs, extra = {}, {"N": N, "nverts_in": 600, "seed": 321, "bvn": 2}
    for steps in (4, 5):
        sampler = ns.md.SyncDDIMSampler(model, steps, ddim_discretize="uniform", ddim_eta=1.0, latent_size=32)
        eps_sink = []
        record_eps(sampler, eps_sink)
        torch.manual_seed(321)
        x, inter = sampler.sample({"x": x_in, "elevation": batch["input_elevation"][:, 0]}, clip, unconditional_scale=2.0,
                                  log_every_t=1, batch_view_num=2, batch=batch)
        assert len(inter["x_inter"]) == steps and len(eps_sink) == steps
        extra[f"timesteps{steps}"] = sampler.ddim_timesteps.astype(np.int64)
        for i in range(steps):
            packs[f"s{steps}_x{i}"] = gi.pack(inter["x_inter"][i].detach())
            packs[f"s{steps}_eps{i}"] = gi.pack(eps_sink[i])
        packs[f"s{steps}_final"] = gi.pack(x.detach())
    save("traj_small.npz", packs, extra)


def gold_trained(ns, full=True):
    """Second weight set (weights.py style "trained"): UNet eps and one whole step, reduced and full width."""
    gold_unet_trained(ns)
    gold_step(ns, "step_small_trained.npz", gi.SMALL_UNET, 4, "perspective", 33, True, 600, 2, style="trained")
    if full:
        gold_step(ns, "step_full_trained.npz", gi.FULL_UNET, 16, "perspective", 20, True, 5023, 16, style="trained")


FULL_TRAIN_TENSORS = (  # the sample of tools/make_goldens.py --only-train-full: every block kind, every level, both ends
    "input_blocks.0.0.weight", "input_blocks.1.0.in_layers.2.weight", "input_blocks.1.0.emb_layers.1.weight",
    "input_blocks.1.1.transformer_blocks.0.attn1.to_q.weight", "input_blocks.2.1.transformer_blocks.0.ff.net.0.proj.weight",
    "input_blocks.3.0.op.weight", "input_blocks.4.0.skip_connection.weight", "input_blocks.5.1.proj_in.weight",
    "input_blocks.7.1.transformer_blocks.0.attn1.to_out.0.weight", "input_blocks.8.0.out_layers.3.weight",
    "input_blocks.10.0.in_layers.2.weight", "middle_block.1.transformer_blocks.0.ff.net.2.weight", "middle_block.2.out_layers.0.weight",
    "middle_conditions.proj_in.0.weight", "middle_conditions.depth_attn.to_k.weight", "output_conditions.2.proj_out.3.weight",
    "output_conditions.5.depth_attn.to_q.weight", "output_conditions.8.proj_context.0.weight", "output_conditions.8.proj_out.5.weight",
    "output_blocks.2.1.conv.weight", "output_blocks.5.1.transformer_blocks.0.norm1.weight", "output_blocks.8.0.in_layers.2.weight",
    "output_blocks.11.0.out_layers.3.bias", "output_blocks.11.1.proj_out.weight", "out.2.weight", "time_embed.2.weight")


def gold_train(ns, ucfg=None, B=4, out="train_small.npz", only=None, drops=(0.03, 0.12, 0.07, 0.6), with_conditioner=True):
    """(ucfg / B / out / only: the full-width variant stores loss, prediction and the gradients of the tensors named in `only`.)
    f2: the reference's training_step (morphable_diffusion.py:520-549) at reduced width, run with
    the reference's own methods in the reference's own order -- time steps, add_noise (:551-565), random target view,
    construct_spatial_volume on all noisy views, one frustum volume per sample, UNetWrapper.forward(is_train=True) with
    condition dropout (:95-130), MSE -- then loss.backward().  ``prepare`` (VAE / CLIP) is replaced by seeded latents.
    The random draws are stored so that the HIP path can be fed the same ones; the dropout's uniform draw is INJECTED
    (torch.rand patched for that one call) so that all four branches of get_drop_scheme (:84-93) occur in a batch of 4.
    Stored: loss, noise_predict, dL/dpred, the gradient of EVERY UNet parameter (sample + norm; the 170 DepthTransformer
    tensors with larger samples) and the gradient w.r.t. the four frustum volumes."""
    N = 4
    ucfg, vcfg = ucfg or gi.SMALL_UNET, VolumeConfig(num_views=N)
    model, _ = build_full_model(ns, ucfg, vcfg, N)
    model.model.drop_conditions = True
    model.train()
    b0 = synthetic.make_batch(N, "perspective", 500, mesh_seed=1)
    batch = {k: v.repeat(B, *([1] * (v.dim() - 1))).clone() for k, v in b0.items()}
    # different cameras per sample: rotate the rig index
    for bi in range(B):
        batch["target_K"][bi] = b0["target_K"][0].roll(bi, 0)
        batch["target_RT"][bi] = b0["target_RT"][0].roll(bi, 0)
    g = torch.Generator().manual_seed(77)
    x0 = torch.randn(B, N, 4, 32, 32, generator=g) * 0.8           # target latents (what prepare() returns as x)
    x_in = torch.randn(B, 4, 32, 32, generator=g) * 0.18215
    clip = torch.randn(B, 1, 768, generator=g)
    drop_random = torch.tensor(list(drops)[:B])                    # drop all | drop volume | drop concat | keep everything
    torch.manual_seed(4242)
    time_steps = torch.randint(0, model.num_timesteps, (B,)).long()
    x_noisy, noise = model.add_noise(x0, time_steps)
    target_index = torch.randint(0, N, (B, 1)).long()
    v_embed = model.get_viewpoint_embedding(batch)
    t_embed = model.embed_time(time_steps)
    sv = model.spatial_volume.construct_spatial_volume(x_noisy, t_embed, v_embed, batch)
    clip_, vf, xc = model.get_target_view_feats(x_in, sv, clip, t_embed, v_embed, target_index, batch)
    x_noisy_ = x_noisy[torch.arange(B)[:, None], target_index][:, 0]
    # the frustum volumes as LEAVES: the stored d loss / d volume is then what flows out of the UNet alone (in the full graph
    # volume 16 also feeds volume 32 through the frustum net's up path, so its .grad would mix in the conditioner's own
    # backward, which this golden does not cover); UNet parameter gradients are unaffected
    vf = {k_: v_.detach().requires_grad_(True) for k_, v_ in vf.items()}
    vf_pre = dict(vf)  # UNetWrapper.forward replaces the dict entries by their dropped versions (:114-115)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: drop_random.clone()
    try:
        pred = model.model(x_noisy_, time_steps, clip_, vf, xc, is_train=True)
    finally:
        torch.rand = real_rand
    pred.retain_grad()
    noise_target = noise[torch.arange(B)[:, None], target_index][:, 0]
    loss = torch.nn.functional.mse_loss(noise_target, pred, reduction="none").mean()
    loss.backward()
    packs = {"noise_predict": gi.pack(pred), "loss": gi.pack(loss.reshape(1)), "dpred": gi.pack(pred.grad),
             "x_noisy": gi.pack(x_noisy)}
    # every parameter of the UNet (finetune_unet: True trains all of them, configs/facescape.yaml:10; the 170 tensors of
    # get_trainable_parameters(), attention.py:140-142, are the middle_conditions / output_conditions entries): a strided
    # sample + the L2 norm of each gradient
    names, norms = [], []
    for n_, p_ in model.model.diffusion_model.named_parameters():
        if only is not None and n_ not in only:
            continue
        if p_.grad is None:  # attn2.to_q / to_k / norm2 see a single context token: autograd may leave them untouched
            g_ = torch.zeros_like(p_)
        else:
            g_ = p_.grad
        cond = n_.startswith(("middle_conditions.", "output_conditions."))
        packs["grad." + n_] = gi.pack(g_, limit=1 << 14 if cond else 2048, target=2048 if cond else 512)
        names.append(n_)
        norms.append(float(g_.double().norm()))
    # gradient w.r.t. the frustum volumes BEFORE the condition dropout: where the conditioner's backward starts
    for k_, v_ in vf_pre.items():
        packs[f"dsrc.{k_}"] = gi.pack(v_.grad, limit=2048, target=4096)
    if not with_conditioner:
        print("train golden:", out, "loss", float(loss), "pred std", float(pred.std()), "UNet grads stored:", len(names))
        save(out, packs, {"B": B, "N": N, "nverts_in": 500, "time_steps": time_steps.numpy(), "target_index": target_index.numpy(),
                          "drop_random": drop_random.numpy(), "seed_latents": 77, "seed_draws": 4242, "grad_names": np.array(names),
                          "grad_norms": np.array(norms)})
        return
    # second pass with the FULL graph (frustum volumes attached to the conditioner): the gradients of spatial_volume.* and of
    # the step-embedding MLP time_embed.* (the reference's second and third optimiser groups, morphable_diffusion.py:639-640)
    model.zero_grad()
    t_embed2 = model.embed_time(time_steps)
    sv2 = model.spatial_volume.construct_spatial_volume(x_noisy, t_embed2, v_embed, batch)
    clip2, vf2, xc2 = model.get_target_view_feats(x_in, sv2, clip, t_embed2, v_embed, target_index, batch)
    torch.rand = lambda *a, **k: drop_random.clone()
    try:
        pred2 = model.model(x_noisy_, time_steps, clip2, vf2, xc2, is_train=True)
    finally:
        torch.rand = real_rand
    assert torch.equal(pred2, pred)
    torch.nn.functional.mse_loss(noise_target, pred2, reduction="none").mean().backward()
    cnames, cnorms = [], []
    for pre_, mod_ in (("time_embed.", model.time_embed), ("spatial_volume.", model.spatial_volume)):
        for n_, p_ in mod_.named_parameters():
            g_ = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            packs["gradc." + pre_ + n_] = gi.pack(g_, limit=2048, target=512)
            cnames.append(pre_ + n_)
            cnorms.append(float(g_.double().norm()))
    extra_c = {"cond_names": np.array(cnames), "cond_norms": np.array(cnorms)}
    print("train golden: loss", float(loss), "pred std", float(pred.std()), "UNet grads:", len(names), "zero-norm:",
          [n for n, v in zip(names, norms) if v == 0.0][:8], "| conditioner grads:", len(cnames), "zero-norm:",
          [n for n, v in zip(cnames, cnorms) if v == 0.0][:6])
    save(out, packs, {"B": B, "N": N, "nverts_in": 500, "time_steps": time_steps.numpy(),
                                    "target_index": target_index.numpy(), "drop_random": drop_random.numpy(),
                                    "seed_latents": 77, "seed_draws": 4242, "grad_names": np.array(names),
                                    "grad_norms": np.array(norms), **extra_c})


def gold_cameras():
    """generate_camera_trajectory of the reference's generate_face.py (:25-45), extracted from its source by name with ast (the
    module itself cannot be imported here: torchvision / pytorch3d / carvekit) and executed; the camera matrices of the
    script's main loop (:161-173) are then built from it with scipy, as the script does."""
    import ast
    from scipy.spatial.transform import Rotation as Rot
    src = open("/root/reference/generate_face.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "generate_camera_trajectory"]
    assert len(fn) == 1
    scope = {"np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), "generate_face.py", "exec"), scope)
    out = {}
    for n in (16, 8):
        pos, rot = scope["generate_camera_trajectory"](n)
        RTs = []
        for p_, r_ in zip(pos, rot):  # generate_face.py:166-173
            R = Rot.from_euler("xyz", np.array(r_), True).as_matrix()
            RT = np.zeros((3, 4))
            RT[:3, :3] = R
            RT[:3, 3] = (-R @ np.array(p_).reshape(3, 1)).reshape(3,)
            RTs.append(RT)
        out[f"positions{n}"], out[f"rotations{n}"], out[f"RT{n}"] = np.array(pos), np.array(rot), np.array(RTs)
    np.savez_compressed(os.path.join(OUT, "cameras.npz"), **out)
    print("wrote cameras.npz")


def gold_variants(ns):
    """The other BASELINE.json configs as parity cases, at reduced UNet width:
    config 1 (N=8, 256^2), config 0 (one view, 64^2 latent, FLAME-sized mesh, first DDIM step without noise) and
    config 4 (SMPL-X-sized mesh, N=32, 512^2 -> 64^2 latent, orthographic cameras)."""
    import dataclasses
    small64 = dataclasses.replace(gi.SMALL_UNET, image_size=64)
    gold_step(ns, "step_small_n8.npz", gi.SMALL_UNET, 8, "perspective", 10, True, 600, 4)
    gold_step(ns, "step_small_lat64_n1.npz", small64, 1, "perspective", 0, False, 5023, 1, image_size=512)
    gold_step(ns, "step_small_smplx_n32.npz", small64, 32, "orthographic", 30, True, 10475, 4, image_size=512,
              radii=(0.18, 0.45, 0.12))


def gold_variants_full(ns, which=("n8", "lat64_n1", "smplx_n32")):
    """The same three BASELINE.json configurations at the FULL UNet width (916.9 M parameters) -- the width their parity was
    only property-tested at before: configs[1] (N=8, 256^2, 8 views per pass), configs[0] (one view, 64^2 latent, first DDIM
    step) and configs[4] (SMPL-X-sized mesh, N=32, 512^2 -> 64^2 latents, orthographic, 4 views per pass).  Minutes of fp32 eager
    PyTorch each on a few cores; the fixtures hold strided samples + checksums (tests/golden_inputs.pack)."""
    import dataclasses
    full64 = dataclasses.replace(gi.FULL_UNET, image_size=64)
    if "n8" in which:
        gold_step(ns, "step_full_n8.npz", gi.FULL_UNET, 8, "perspective", 10, True, 5023, 8)
    if "lat64_n1" in which:
        gold_step(ns, "step_full_lat64_n1.npz", full64, 1, "perspective", 0, False, 5023, 1, image_size=512)
    if "smplx_n32" in which:
        gold_step(ns, "step_full_smplx_n32.npz", full64, 32, "orthographic", 30, True, 10475, 4, image_size=512,
                  radii=(0.18, 0.45, 0.12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--only-variants-full", default=None,
                    help="only the FULL-WIDTH BASELINE config 0/1/4 variants; comma list of n8,lat64_n1,smplx_n32 (or 'all')")
    ap.add_argument("--only-variants", action="store_true", help="only the BASELINE config 0/1/4 variants")
    ap.add_argument("--only-vae", action="store_true", help="only the first-stage decoder goldens")
    ap.add_argument("--only-train", action="store_true", help="only the training-step golden (loss + gradients)")
    ap.add_argument("--only-train-full", action="store_true", help="only the FULL-WIDTH training-step golden (loss, prediction, a sample of 26 gradient tensors)")
    ap.add_argument("--only-traj", action="store_true", help="only the multi-step trajectory golden")
    ap.add_argument("--only-trained", action="store_true", help="only the goldens on the trained-like weight set")
    ap.add_argument("--only-cameras", action="store_true", help="only the camera-trajectory golden (ast-extracted from generate_face.py)")
    ap.add_argument("--only-clip", action="store_true", help="only the CLIP image-embedding goldens (needs transformers, not the reference)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if args.only_clip:
        gold_clip()
        return
    if args.only_cameras:
        gold_cameras()
        return
    ns = ref_import.import_reference_full()
    if args.only_variants:
        gold_variants(ns)
        return
    if args.only_variants_full:
        w = args.only_variants_full
        gold_variants_full(ns, ("n8", "lat64_n1", "smplx_n32") if w == "all" else tuple(w.split(",")))
        return
    if args.only_vae:
        gold_vae(ns)
        return
    if args.only_traj:
        gold_traj(ns)
        return
    if args.only_train:
        gold_train(ns)
        return
    if args.only_train_full:
        gold_train(ns, ucfg=gi.FULL_UNET, B=2, out="train_full.npz", only=set(FULL_TRAIN_TENSORS), drops=(0.6, 0.12),
                   with_conditioner=False)
        return
    if args.only_trained:
        gold_trained(ns, not args.skip_full)
        return
    gold_basic(ns)
    gold_unet_small(ns)
    hot = gold_step(ns, "step_small_persp.npz", gi.SMALL_UNET, 4, "perspective", 25, True, 600, 2, stages=True)
    gold_step(ns, "step_small_ortho.npz", gi.SMALL_UNET, 4, "orthographic", 0, False, 600, 4, stages=True)
    if not args.skip_full:
        gold_unet_full(ns)
        hot = gold_step(ns, "step_full.npz", gi.FULL_UNET, 16, "perspective", 49, True, 5023, 8, frustum_views=2)
    gold_traj(ns)
    gold_train(ns)
    gold_cameras()
    gold_variants(ns)
    if not args.skip_full:
        gold_variants_full(ns)
    gold_trained(ns, not args.skip_full)
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump({k: list(v) for k, v in sorted(hot.items())}, f)
    # DDIM tables from the reference sampler
    model, _ = build_full_model(ns, gi.SMALL_UNET, VolumeConfig(num_views=4), 4)
    s = model.sampler
    np.savez_compressed(os.path.join(OUT, "ddim.npz"), timesteps=s.ddim_timesteps.astype(np.int64),
                        alphas=s.ddim_alphas.numpy(), alphas_prev=s.ddim_alphas_prev.numpy(),
                        sigmas=s.ddim_sigmas.numpy(), sqrt_one_minus_alphas=s.ddim_sqrt_one_minus_alphas.numpy())
    print("done")


if __name__ == "__main__":
    main()
