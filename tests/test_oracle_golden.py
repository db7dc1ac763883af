"""CPU: pin the oracle (oracle/mvd_oracle.py) against the golden vectors produced by the reference's own
Python (tools/make_goldens.py).  fp32 vs fp32, so tolerances are fp32-roundoff class."""
import json
import os

import numpy as np
import pytest
import torch

from morphablediffusion_amd import synthetic
from morphablediffusion_amd.spec import VolumeConfig, build_unet_plan, full_manifest
from oracle import mvd_oracle as O
from tests import golden_inputs as gi

G = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(G, name))


def check(t, g, key, rtol=2e-4):
    got, want, sums = gi.unpack_compare(t, g, key)
    assert got.shape == want.shape, (key, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item() / scale
    assert err < rtol, f"{key}: normalised max err {err:.3e}"
    if sums is not None:
        s, ws, a, wa = sums
        assert abs(a - wa) <= 1e-4 * wa + 1e-6, f"{key}: abssum {a} vs {wa}"
        assert abs(s - ws) <= 1e-4 * wa + 1e-6, f"{key}: sum {s} vs {ws}"


def test_manifest_matches_reference_state_dict():
    with open(os.path.join(G, "manifest.json")) as f:
        ref = {k: tuple(v) for k, v in json.load(f).items()}
    assert ref == {k: tuple(v) for k, v in full_manifest(gi.FULL_UNET, VolumeConfig()).items()}


def test_timestep_embedding():
    g = load("basic.npz")
    t = torch.tensor([1, 481, 981])
    for dim in (256, 320):
        check(O.timestep_embedding(t, dim), g, f"temb{dim}", 1e-6)


def test_ddim_tables():
    g = load("ddim.npz")
    tab = O.ddim_tables(50, 1.0)
    assert np.array_equal(tab["timesteps"].numpy(), g["timesteps"])
    for k in ("alphas", "alphas_prev", "sigmas", "sqrt_one_minus_alphas"):
        assert np.array_equal(tab[k].numpy(), g[k]), k


@pytest.fixture(scope="module")
def small():
    cfg = gi.SMALL_UNET
    return cfg, build_unet_plan(cfg), gi.unet_weights(cfg), gi.unet_inputs(cfg, Bv=2), load("unet_small.npz")


def test_unet_small_forward(small):
    cfg, plan, W, (x, t, ctx, sd), g = small
    check(O.unet_forward(W, plan, x, t, ctx, sd), g, "unet_out")


def test_unet_small_blocks(small):
    cfg, plan, W, (x, t, ctx, sd), g = small
    P = "model.diffusion_model."
    emb = O.timestep_embedding(t, cfg.model_channels)
    emb = torch.nn.functional.linear(emb, W[P + "time_embed.0.weight"], W[P + "time_embed.0.bias"])
    emb = torch.nn.functional.linear(O.silu(emb), W[P + "time_embed.2.weight"], W[P + "time_embed.2.bias"])
    check(emb, g, "emb")
    gen = torch.Generator().manual_seed(5)
    h64 = torch.randn(2, 64, 32, 32, generator=gen)
    h128 = torch.randn(2, 128, 16, 16, generator=gen)
    h256 = torch.randn(2, 256, 8, 8, generator=gen)
    check(O.res_block(W, P + "input_blocks.1.0", h64, emb), g, "res_same")
    check(O.res_block(W, P + "input_blocks.4.0", h64[:, :, ::2, ::2].contiguous(), emb), g, "res_skip")
    check(O.spatial_transformer(W, P + "input_blocks.1.1", h64, ctx, 8), g, "st32")
    check(O.spatial_transformer(W, P + "input_blocks.4.1", h128, ctx, 8), g, "st16")
    check(O.spatial_transformer(W, P + "input_blocks.7.1", h256, ctx, 8), g, "st8")
    F = torch.nn.functional
    check(F.conv2d(h64, W[P + "input_blocks.3.0.op.weight"], W[P + "input_blocks.3.0.op.bias"], stride=2, padding=1), g, "down")
    up = h128.repeat_interleave(2, 2).repeat_interleave(2, 3)
    check(F.conv2d(up, W[P + "output_blocks.8.2.conv.weight"], W[P + "output_blocks.8.2.conv.bias"], padding=1), g, "up")
    check(O.depth_transformer(W, P + "output_conditions.8", h64, sd[32]), g, "cond8")
    check(O.depth_transformer(W, P + "output_conditions.3", h128, sd[16]), g, "cond3")
    check(O.depth_transformer(W, P + "middle_conditions", h256[:, :, ::2, ::2].contiguous(), sd[4]), g, "cond_mid")


@pytest.mark.parametrize("name,projection", [("step_small_persp.npz", "perspective"), ("step_small_ortho.npz", "orthographic")])
def test_step_and_stages_small(name, projection):
    g = load(name)
    N, index, bvn = int(g["N"]), int(g["index"]), int(g["bvn"])
    ucfg = gi.SMALL_UNET
    vcfg = VolumeConfig(num_views=N, projection=projection)
    plan = build_unet_plan(ucfg)
    W = gi.full_weights(ucfg, vcfg)
    batch = synthetic.make_batch(N, projection, int(g["nverts_in"]), mesh_seed=1)
    assert batch["vertices"].shape[1] == int(g["nverts"])
    x_T, x_in, clip = synthetic.make_latents(N, 32, seed=6033)
    tab = O.ddim_tables(50, 1.0)
    ts = torch.full((1,), int(g["step"]), dtype=torch.long)
    assert int(tab["timesteps"][index]) == int(g["step"])
    v_embed = O.viewpoint_embedding(batch)
    t_embed = O.embed_time(W, ts)
    check(t_embed, g, "t_embed")
    check(v_embed, g, "v_embed", 1e-6)
    check(O.target_encoder(W, x_T[:, 0], t_embed, v_embed[:, 0]), g, "enc_view0")
    pts = O.lattice(32, 0.5)
    for vi in (0, N - 1):
        uv = O.warp_coordinates(pts, 32, 256, batch["target_K"][:, vi], batch["target_RT"][:, vi], projection)
        check(uv.reshape(1, 32, 32, 32, 2), g, f"warp_view{vi}", 1e-5)
    vf = O.vertex_features(W, vcfg, x_T, t_embed, v_embed, batch)
    check(vf, g, "vertex_feats")
    fused = O.fuse_views(W, vf)
    check(fused, g, "fused")
    dense = O.sparse_conv_net(W, fused[0], batch["coord"][0], batch["out_sh"][0])
    check(dense, g, "sparse_dense")
    sv = O.construct_spatial_volume(W, vcfg, x_T, t_embed, v_embed, batch)
    check(sv, g, "spatial_volume")
    xyz = O.frustum_points(vcfg, batch["target_RT"][0, :2], batch["target_K"][0, :2])
    check(xyz, g, "frustum_xyz", 2e-5)
    fd = O.construct_view_frustum_volume(W, vcfg, sv, t_embed, v_embed, torch.arange(0, 2)[None], batch)
    for k, v in fd.items():
        check(v, g, f"frustum_{k}")
    noise = None
    if int(g["with_noise"]):
        torch.manual_seed(int(g["noise_seed"]))
        noise = torch.randn(x_T.shape)
    out, eps = O.denoise_apply(W, plan, vcfg, tab, x_T, x_in, clip, ts, index, 2.0, batch, batch_view_num=bvn, noise=noise,
                               return_eps=True)
    check(out, g, "x_prev")
    check(eps, g, "eps")


@pytest.mark.parametrize("name,projection", [("step_small_n8.npz", "perspective"), ("step_small_lat64_n1.npz", "perspective"),
                                             ("step_full_lat64_n1.npz", "perspective")])
def test_step_config_variants(name, projection):
    """BASELINE configs 1 and 0 at reduced width (N=8 at 256^2; one view at a 64^2 latent, first DDIM step), and config 0 at the
    FULL UNet width (the reference's own CPU-runnable case).  The full-width N=8 and the N=32 / SMPL-X-sized variants are
    checked on the GPU only: the oracle needs minutes for them."""
    import dataclasses
    g = load(name)
    N, index, bvn, size = int(g["N"]), int(g["index"]), int(g["bvn"]), int(g["image_size"])
    ucfg = dataclasses.replace(gi.FULL_UNET if name.startswith("step_full") else gi.SMALL_UNET, image_size=size // 8)
    vcfg = VolumeConfig(num_views=N, projection=projection, input_image_size=size)
    W = gi.full_weights(ucfg, vcfg)
    batch = synthetic.make_batch(N, projection, int(g["nverts_in"]), mesh_seed=1, image_size=size,
                                 radii=tuple(float(r) for r in g["radii"]))
    x_T, x_in, clip = synthetic.make_latents(N, size // 8, seed=6033)
    tab = O.ddim_tables(50, 1.0)
    ts = torch.full((1,), int(g["step"]), dtype=torch.long)
    noise = None
    if int(g["with_noise"]):
        torch.manual_seed(int(g["noise_seed"]))
        noise = torch.randn(x_T.shape)
    out, eps = O.denoise_apply(W, build_unet_plan(ucfg), vcfg, tab, x_T, x_in, clip, ts, index, 2.0, batch,
                               batch_view_num=bvn, noise=noise, return_eps=True)
    check(out, g, "x_prev")
    check(eps, g, "eps")


def test_trajectory_small():
    """a1: SyncDDIMSampler.sample (morphable_diffusion.py:742-776) -- the oracle's loop against the reference's own loop on
    4- and 5-step schedules: time-step table, index order, RNG consumption (x_T, then one draw per step but the last),
    every intermediate x and every step's eps."""
    g = load("traj_small.npz")
    N = int(g["N"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    W = gi.full_weights(ucfg, vcfg)
    batch = synthetic.make_batch(N, "perspective", int(g["nverts_in"]), mesh_seed=1)
    _, x_in, clip = synthetic.make_latents(N, 32, seed=6033)
    for steps in (4, 5):
        assert np.array_equal(O.ddim_tables(steps, 1.0)["timesteps"].numpy(), g[f"timesteps{steps}"])
        gen = torch.Generator().manual_seed(int(g["seed"]))
        x, inter, eps = O.sample(W, build_unet_plan(ucfg), vcfg, x_in, clip, 2.0, batch, num_ddim=steps,
                                 batch_view_num=int(g["bvn"]), log_every_t=1, generator=gen)
        assert len(inter) == steps and len(eps) == steps
        for i in range(steps):
            check(eps[i], g, f"s{steps}_eps{i}", 5e-4)  # later steps see the accumulated fp32 differences
            check(inter[i], g, f"s{steps}_x{i}", 5e-4)
        check(x, g, f"s{steps}_final", 5e-4)


def test_unet_small_trained_weights():
    """Second weight set (trained-checkpoint-like statistics, weights.py style "trained")."""
    cfg = gi.SMALL_UNET
    g = load("unet_small_trained.npz")
    W = gi.unet_weights(cfg, "trained")
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2, seed=13)
    check(O.unet_forward(W, build_unet_plan(cfg), x, t, ctx, sd), g, "unet_out")


def test_step_small_trained_weights():
    g = load("step_small_trained.npz")
    N, index, bvn = int(g["N"]), int(g["index"]), int(g["bvn"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    W = gi.full_weights(ucfg, vcfg, "trained")
    batch = synthetic.make_batch(N, "perspective", int(g["nverts_in"]), mesh_seed=1)
    x_T, x_in, clip = synthetic.make_latents(N, 32, seed=6033)
    tab = O.ddim_tables(50, 1.0)
    ts = torch.full((1,), int(g["step"]), dtype=torch.long)
    torch.manual_seed(int(g["noise_seed"]))
    noise = torch.randn(x_T.shape)
    out, eps = O.denoise_apply(W, build_unet_plan(ucfg), vcfg, tab, x_T, x_in, clip, ts, index, 2.0, batch,
                               batch_view_num=bvn, noise=noise, return_eps=True)
    check(out, g, "x_prev")
    check(eps, g, "eps")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "unet_full.npz")), reason="full-width golden not generated")
def test_unet_full_forward():
    cfg = gi.FULL_UNET
    g = load("unet_full.npz")
    W = gi.unet_weights(cfg)
    x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2)
    check(O.unet_forward(W, build_unet_plan(cfg), x, t, ctx, sd), g, "unet_out")


@pytest.mark.skipif(not os.path.exists(os.path.join(G, "step_full.npz")), reason="full-width golden not generated")
def test_step_full_width_n16():
    """One full denoise_apply at the headline shape (N=16, 5023-vertex mesh, full-width UNet, CFG 2.0)."""
    g = load("step_full.npz")
    N, index, bvn = int(g["N"]), int(g["index"]), int(g["bvn"])
    ucfg, vcfg = gi.FULL_UNET, VolumeConfig(num_views=N)
    W = gi.full_weights(ucfg, vcfg)
    batch = synthetic.make_batch(N, "perspective", int(g["nverts_in"]), mesh_seed=1)
    x_T, x_in, clip = synthetic.make_latents(N, 32, seed=6033)
    tab = O.ddim_tables(50, 1.0)
    ts = torch.full((1,), int(g["step"]), dtype=torch.long)
    torch.manual_seed(int(g["noise_seed"]))
    noise = torch.randn(x_T.shape)
    out, eps = O.denoise_apply(W, build_unet_plan(ucfg), vcfg, tab, x_T, x_in, clip, ts, index, 2.0, batch,
                               batch_view_num=bvn, noise=noise, return_eps=True)
    check(out, g, "x_prev")
    check(eps, g, "eps")


@pytest.mark.parametrize("name,ch", [("vae_small.npz", 32), ("vae_full.npz", 128)])
def test_vae_decoder(name, ch):
    """First-stage decoder against the reference's AutoencoderKL.decode."""
    from morphablediffusion_amd.spec import VaeConfig, vae_decoder_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import vae_oracle as V
    g = load(name)
    cfg = VaeConfig(ch=ch)
    W = seeded_state_dict(vae_decoder_manifest(cfg), gi.WEIGHT_SEED)
    gen = torch.Generator().manual_seed(31)
    z = torch.randn(int(g["B"]), cfg.embed_dim, 32, 32, generator=gen) * 4.0
    check(V.decode(W, cfg, z), g, "out")


@pytest.mark.parametrize("name,ch", [("vae_small.npz", 32), ("vae_full.npz", 128)])
def test_vae_encoder(name, ch):
    """First-stage encoder against the reference's AutoencoderKL.encode(x).parameters."""
    from morphablediffusion_amd.spec import VaeConfig, vae_encoder_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import vae_oracle as V
    g = load(name)
    cfg = VaeConfig(ch=ch)
    W = seeded_state_dict(vae_encoder_manifest(cfg), gi.WEIGHT_SEED)
    gen = torch.Generator().manual_seed(31)
    B = int(g["B"])
    torch.randn(B, cfg.embed_dim, 32, 32, generator=gen)  # the latent draw of the decoder golden comes first
    x = torch.rand(B, 3, 256, 256, generator=gen) * 2.0 - 1.0
    check(V.encode_moments(W, cfg, x), g, "moments")


@pytest.mark.parametrize("name,kw", [("clip_small.npz", dict(width=128, layers=2, heads=2, embed=64)), ("clip_full.npz", {})])
def test_clip_image_embedding(name, kw):
    """CLIP image embedding: the oracle against transformers' CLIPVisionModelWithProjection on
    the same seeded weights (openai/CLIP itself is not importable here; see oracle/clip_oracle.py)."""
    from morphablediffusion_amd.spec import ClipConfig, clip_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    from oracle import clip_oracle as CO
    g = load(name)
    cfg = ClipConfig(**kw)
    W = seeded_state_dict(clip_manifest(cfg), gi.WEIGHT_SEED)
    gen = torch.Generator().manual_seed(47)
    x = torch.rand(int(g["B"]), 3, 256, 256, generator=gen) * 2.0 - 1.0
    check(CO.preprocess(x, cfg.image), g, "pixels")
    check(CO.encode(W, cfg, x), g, "embed")


def _train_inputs(g):
    B, N = int(g["B"]), int(g["N"])
    b0 = synthetic.make_batch(N, "perspective", int(g["nverts_in"]), mesh_seed=1)
    batch = {k: v.repeat(B, *([1] * (v.dim() - 1))).clone() for k, v in b0.items()}
    for bi in range(B):
        batch["target_K"][bi] = b0["target_K"][0].roll(bi, 0)
        batch["target_RT"][bi] = b0["target_RT"][0].roll(bi, 0)
    gen = torch.Generator().manual_seed(int(g["seed_latents"]))
    x0 = torch.randn(B, N, 4, 32, 32, generator=gen) * 0.8
    x_in = torch.randn(B, 4, 32, 32, generator=gen) * 0.18215
    clip = torch.randn(B, 1, 768, generator=gen)
    # the reference's stream: randint (time steps), randn_like (noise), randint (target view) on one seeded CPU generator
    torch.manual_seed(int(g["seed_draws"]))
    ts = torch.randint(0, 1000, (B,)).long()
    noise = torch.randn_like(x0)
    ti = torch.randint(0, N, (B, 1)).long()
    assert np.array_equal(ts.numpy(), g["time_steps"]) and np.array_equal(ti.numpy(), g["target_index"])
    return batch, x0, x_in, clip, ts, noise, ti, torch.from_numpy(np.asarray(g["drop_random"]))


def test_training_step_loss_and_gradients():
    """f2: the oracle's training_step (forward, loss, and -- through autograd on the functional restatement -- the gradient of
    EVERY UNet parameter) against the reference's own training_step + loss.backward()."""
    g = load("train_small.npz")
    N = int(g["N"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    W = gi.full_weights(ucfg, vcfg)
    batch, x0, x_in, clip, ts, noise, ti, dr = _train_inputs(g)
    P = "model.diffusion_model."
    names = [str(n) for n in g["grad_names"]]
    assert len(names) == 856 and sum(n.startswith(("middle_conditions.", "output_conditions.")) for n in names) == 170
    for n in names:
        W[P + n].requires_grad_(True)
    cnames = [str(n) for n in g["cond_names"]]
    assert len(cnames) == 149
    for n in cnames:  # spatial_volume.* and the step-embedding MLP time_embed.*
        W[n].requires_grad_(True)
    loss, pred = O.training_step(W, build_unet_plan(ucfg), vcfg, x0, x_in, clip, batch, ts, noise, ti, dr)
    pred.retain_grad()
    loss.backward()
    check(loss.detach().reshape(1), g, "loss", 1e-5)
    check(pred.detach(), g, "noise_predict")
    check(pred.grad, g, "dpred", 1e-5)
    for n, want_norm in zip(names, g["grad_norms"]):
        gr = W[P + n].grad
        if want_norm == 0.0:  # attn2.to_q / to_k / norm2: one context token, softmax == 1, exactly zero gradient
            assert gr is None or float(gr.abs().max()) == 0.0, n
            continue
        check(gr, g, "grad." + n, 2e-3)
        assert abs(float(gr.double().norm()) - want_norm) <= 2e-3 * want_norm, n
    for n, want_norm in zip(cnames, g["cond_norms"]):
        check(W[n].grad, g, "gradc." + n, 2e-3)
        assert abs(float(W[n].grad.double().norm()) - want_norm) <= 2e-3 * want_norm, n
    m_clip, m_vol, m_cat = O.drop_masks(dr)
    assert m_clip.tolist() == [0, 1, 1, 1] and m_vol.tolist() == [0, 0, 1, 1] and m_cat.tolist() == [0, 1, 0, 1]


# ---- a11: the sparse voxel CNN on three independent derivations -----------------------------------------------------------
def _hashmap_sparse_conv(feats, coords, shape, weight, stride):
    """spconv's own formulation (get_indice_pairs + gather - GEMM - scatter-add), written from its documented algorithm and
    independent of both the oracle's index-grid gather and the dense-masked emulation the goldens come from: every ACTIVE
    INPUT proposes, for each kernel offset, the output site it contributes to; a hash map (a dict) assigns output rows;
    per offset the (input row, output row) pairs are gathered, multiplied by that offset's [Cin, Cout] matrix and
    scatter-added.  stride 1 = SubMConv3d (output sites are exactly the input sites: pairs whose output voxel is not active
    are dropped), stride 2 = SparseConv3d(k3, s2, p1) (every proposed site inside the output extent becomes active)."""
    inp = {tuple(c): i for i, c in enumerate(coords.tolist())}
    oshape = list(shape) if stride == 1 else [(s - 1) // 2 + 1 for s in shape]
    if stride == 1:
        out_index = dict(inp)
    else:
        sites = set()
        for (z, y, x) in inp:
            for kz in range(3):
                for ky in range(3):
                    for kx in range(3):
                        nz, ny, nx = z + 1 - kz, y + 1 - ky, x + 1 - kx  # out * 2 - pad + k = in
                        if nz % 2 or ny % 2 or nx % 2:
                            continue
                        o = (nz // 2, ny // 2, nx // 2)
                        if all(0 <= o[a] < oshape[a] for a in range(3)):
                            sites.add(o)
        out_index = {o: i for i, o in enumerate(sorted(sites))}
    out = torch.zeros(len(out_index), weight.shape[0], dtype=torch.float64)
    for kz in range(3):
        for ky in range(3):
            for kx in range(3):
                pin, pout = [], []
                for (z, y, x), i in inp.items():
                    nz, ny, nx = z + 1 - kz, y + 1 - ky, x + 1 - kx
                    if stride == 2:
                        if nz % 2 or ny % 2 or nx % 2:
                            continue
                        o = (nz // 2, ny // 2, nx // 2)
                    else:
                        o = (nz, ny, nx)
                    j = out_index.get(o)
                    if j is not None:
                        pin.append(i)
                        pout.append(j)
                if pin:
                    out.index_add_(0, torch.tensor(pout), feats[pin].double() @ weight[:, :, kz, ky, kx].t().double())
    ocoords = torch.tensor(sorted(out_index, key=out_index.get), dtype=torch.long).reshape(-1, 3)
    return out.float(), ocoords, oshape


@pytest.mark.parametrize("case", ["odd extents", "duplicates", "border sites"])
def test_sparse_cnn_three_derivations_agree(case):
    """SURVEY section 8 row a11 (network.py:74-161; spconv is not importable, requirements.txt:18): the oracle's index-grid
    rule book, spconv's pair-list algorithm restated with a hash map, and the dense-masked emulation behind the goldens must
    give the same network output on meshes the goldens do not cover -- odd voxel extents at every level, several vertices in
    one voxel (first one wins), active sites on the grid border."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ref_import as RI
    from morphablediffusion_amd.spec import full_manifest
    from morphablediffusion_amd.weights import seeded_state_dict
    g = torch.Generator().manual_seed({"odd extents": 3, "duplicates": 4, "border sites": 5}[case])
    shape = {"odd extents": [37, 45, 27], "duplicates": [24, 32, 28], "border sites": [20, 20, 20]}[case]
    n = 900
    coords = torch.stack([torch.randint(0, s, (n,), generator=g) for s in shape], 1)
    if case == "border sites":
        coords[:200, 0] = 0
        coords[200:400, 1] = shape[1] - 1
        coords[400:500, 2] = torch.where(torch.rand(100, generator=g) < 0.5, 0, shape[2] - 1)
    key = (coords[:, 0] * shape[1] + coords[:, 1]) * shape[2] + coords[:, 2]
    if case == "duplicates":
        coords = torch.cat([coords, coords[:300]], 0)  # 300 voxels hold two vertices with DIFFERENT features
    else:  # unique voxels
        _, first = np.unique(key.numpy(), return_index=True)
        coords = coords[np.sort(first)]
    feats = torch.randn(coords.shape[0], 16, generator=g)
    vcfg = VolumeConfig()
    W = {k: v for k, v in seeded_state_dict(full_manifest(gi.SMALL_UNET, vcfg), 7).items() if k.startswith("spatial_volume.xyzc_net.")}
    P = "spatial_volume.xyzc_net."
    want = O.sparse_conv_net(W, feats, coords, shape)[0]  # derivation 1: the oracle
    # the representative of a voxel is its FIRST vertex (what the oracle and the engine implement)
    key = ((coords[:, 0] * shape[1] + coords[:, 1]) * shape[2] + coords[:, 2]).numpy()
    _, first = np.unique(key, return_index=True)
    keep = np.sort(first)
    uc, uf = coords[keep], feats[keep]
    # derivation 2: hash-map pair lists
    x, c2, sh2 = uf, uc, list(shape)
    for blk, nconv in (("conv0", 2), ("down0", 1), ("conv1", 2), ("down1", 1), ("conv2", 3)):
        for i in range(nconv):
            x, c2, sh2 = _hashmap_sparse_conv(x, c2, sh2, W[f"{P}{blk}.{3 * i}.weight"], 2 if blk.startswith("down") else 1)
            x = O._bn_relu(W, f"{P}{blk}.{3 * i + 1}", x)
    got2 = torch.zeros([x.shape[1]] + sh2)
    got2[:, c2[:, 0], c2[:, 1], c2[:, 2]] = x.t()
    # derivation 3: the dense-masked emulation (conv3d * mask), layers assembled as network.py:98-161 does
    t = RI._make_sparse_conv_tensor(uf, torch.cat([torch.zeros(len(uc), 1, dtype=torch.long), uc], 1), shape, 1)
    with torch.no_grad():
        for blk, nconv in (("conv0", 2), ("down0", 1), ("conv1", 2), ("down1", 1), ("conv2", 3)):
            for i in range(nconv):
                w = W[f"{P}{blk}.{3 * i}.weight"]
                conv = RI._SparseConv3d(w.shape[1], w.shape[0], 3, 2, padding=1) if blk.startswith("down") else RI._SubMConv3d(w.shape[1], w.shape[0], 3)
                conv.weight.data.copy_(w)
                bn = torch.nn.BatchNorm1d(w.shape[0], eps=1e-3, momentum=0.01).eval()
                q = f"{P}{blk}.{3 * i + 1}"
                bn.weight.data.copy_(W[q + ".weight"]); bn.bias.data.copy_(W[q + ".bias"])
                bn.running_mean.copy_(W[q + ".running_mean"]); bn.running_var.copy_(W[q + ".running_var"])
                t = RI._SparseSequential(conv, bn, torch.nn.ReLU())(t)
    got3 = t.dense()[0]
    assert list(want.shape) == list(got2.shape) == list(got3.shape), (want.shape, got2.shape, got3.shape)
    assert want.abs().max() > 0
    for name, got in (("hash-map pair lists", got2), ("dense-masked emulation", got3)):
        err = ((got - want).abs().max() / want.abs().max()).item()
        assert (got != 0).eq(want != 0).all(), f"{case}: active sites differ ({name})"
        assert err <= 2e-5, f"{case}: {name} vs oracle {err:.2e}"
