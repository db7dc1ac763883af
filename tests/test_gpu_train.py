"""GPU: f2 -- the reference's training_step (morphable_diffusion.py:520-549) + loss.backward() in the HIP
engine, against the reference's own run on the same draws (tests/golden/train_small.npz: reduced width, B = 4 samples with
different cameras, N = 4 views, all four branches of the condition dropout): loss, prediction, the gradient of EVERY UNet
parameter (856 tensors, of which the 170 of get_trainable_parameters(), attention.py:140-142) and the gradient w.r.t. the
frustum volumes.  Then the optimiser: ArenaAdamW against torch.optim.AdamW, the in-place re-pack against a fresh load.
Tolerances: the engine computes on fp16 MFMA operands with fp32 accumulation (forward AND backward); the configuration's own
dtype, bf16 (BASELINE.json config 4), carries 4e-3 per operation.  Loss 1e-3, prediction 2e-3.  Gradients, relative L2 per
tensor: the 686 tensors of the UNet trunk <= 7e-3 (round 6: 1.5 x the measured worst 4.7e-3, median 2.4e-3 = sqrt(#layers) x the
fp16 operand rounding; was 1e-2).  The 170 DepthTransformer tensors <= 5e-2 (measured: worst 3.7e-2, median 1.0e-2): their backward pass goes through
three ReLU masks and a softmax over nearly uniform depth weights that are re-derived from the block's input, which carries the
forward pass's fp16 rounding (5e-4 .. 1e-3) -- a relative input perturbation eps flips ~eps of the mask bits and moves these
gradients by ~sqrt(eps): tests/test_host_cpu.py::test_depth_transformer_gradient_sensitivity shows the reference arithmetic
itself (fp32 oracle) doing exactly that.  The implementation itself is checked without that conditioning in
test_depth_transformer_backward_exact_inputs below (same inputs on both sides: <= 2e-3)."""
import os

import numpy as np
import pytest
import torch

from morphablediffusion_amd.spec import VolumeConfig
from tests import golden_inputs as gi
from tests.test_gpu_model import compare
from tests.test_oracle_golden import _train_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
P = "model.diffusion_model."


def make_train_model(ucfg, vcfg, N, workspace_gb=8.0, precision_level=2, train_mode=True, **kw):
    from morphablediffusion_amd.model import SyncMultiviewDiffusion
    cfg = dict(volume_dims=list(ucfg.volume_dims), image_size=ucfg.image_size, in_channels=8, out_channels=4,
               model_channels=ucfg.model_channels, attention_resolutions=[4, 2, 1], num_res_blocks=2,
               channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True, transformer_depth=1,
               context_dim=768, use_checkpoint=True, legacy=False)
    m = SyncMultiviewDiffusion(
        unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": cfg},
        scheduler_config={"target": "ldm.lr_scheduler.LambdaLinearScheduler",
                          "params": dict(warm_up_steps=[100], cycle_lengths=[100000], f_start=[0.02], f_max=[1.0], f_min=[1.0])},
        finetune_unet=True, projection=vcfg.projection, view_num=N, image_size=vcfg.input_image_size, cfg_scale=2.0,
        batch_view_num=4, sample_steps=50, workspace_gb=workspace_gb, precision_level=precision_level, train_mode=train_mode, **kw)
    m.load_state_dict(gi.full_weights(ucfg, vcfg))
    m.model.drop_conditions = True
    return m


def _unet_range(eng):
    """[0, hi) of the flat arenas = the UNet's parameters (model.diffusion_model.* sorts before spatial_volume.* / time_embed.*)."""
    return min(o for k, (o, n, s) in eng.param_table.items() if not k.startswith(P))


def _inputs():
    g = np.load(os.path.join(G, "train_small.npz"))
    batch, x0, x_in, clip, ts, noise, ti, dr = _train_inputs(g)
    dev = {k: v.cuda() for k, v in batch.items()}
    prepared = (x0.cuda(), clip.cuda(), {"x": x_in.cuda()})
    return g, dev, prepared, dict(time_steps=ts, noise=noise, target_index=ti, drop_random=dr)


def test_training_step_forward_and_loss_vs_reference():
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, workspace_gb=6.0, train_mode=False)
    loss = m.training_step(dev, prepared=prepared, **draws)
    compare(m.last_noise_predict, g, "noise_predict", rel=2e-3, mx=1e-2)
    want = float(np.asarray(g["loss.full"])[0])
    rl = abs(float(loss) - want) / want
    print(f"[parity] training loss {float(loss):.6f} vs reference {want:.6f}: rel {rl:.2e}")
    assert rl <= 1e-3
    # eval-mode BatchNorm would be a different (wrong) forward: the train-mode statistics are really used
    x0, ts, noise = prepared[0], draws["time_steps"].cuda(), draws["noise"].cuda()
    m.eval()
    sv_eval = m.spatial_volume.construct_spatial_volume(m.add_noise(x0, ts, noise)[0], m.embed_time(ts), m.get_viewpoint_embedding(dev), dev)
    m.train()
    sv_train = m.spatial_volume.construct_spatial_volume(m.add_noise(x0, ts, noise)[0], m.embed_time(ts), m.get_viewpoint_embedding(dev), dev)
    assert not torch.allclose(sv_eval, sv_train, rtol=1e-3, atol=1e-5)
    # the host-side draw order reproduces the reference's CPU stream: same seed -> same time steps / target views
    torch.manual_seed(int(g["seed_draws"]))
    m.training_step(dev, prepared=prepared, drop_random=draws["drop_random"])
    m.engine.close()


def _grad_report(m, g, scale):
    eng = m.engine
    names = [str(n) for n in g["grad_names"]]
    rows = []
    for n, want_norm in zip(names, g["grad_norms"]):
        got = eng.param_view(P + n, grad=True).detach().float().cpu() / scale
        assert torch.isfinite(got).all(), n
        if want_norm == 0.0:  # attn2.to_q / to_k / norm2: exactly zero in the reference, exactly zero here
            assert float(got.abs().max()) == 0.0, n
            continue
        a, b, _ = gi.unpack_compare(got, g, "grad." + n)
        rl = ((a - b).norm() / (b.norm() + 1e-30)).item()
        nr = abs(float(got.double().norm()) - want_norm) / want_norm
        rows.append((rl, nr, n))
    return rows


def test_training_step_every_unet_gradient_vs_reference():
    """loss.backward() of the reference gives the gradient of all 856 UNet parameter tensors; the engine's backward pass
    (forward tape -> per-block backward on the MFMA kernels) must reproduce each of them."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, loss_scale=float(os.environ.get("MVD_TEST_LS", 65536.0)),
                         recompute=False, precision_level=int(os.environ.get("MVD_TEST_PL", 2)))
    params = m.model.get_trainable_parameters()
    assert len(params) == 10 * 17 and all(isinstance(p_, torch.nn.Parameter) and p_.grad is not None for p_ in params)
    assert len(m.model.diffusion_model.named_parameters_all()) == len(g["grad_names"])
    m.engine.zero_grad()
    loss = m.training_step(dev, prepared=prepared, **draws)
    want = float(np.asarray(g["loss.full"])[0])
    assert abs(float(loss) - want) <= 1e-3 * want
    compare(m.last_noise_predict, g, "noise_predict", rel=2e-3, mx=1e-2)
    rows = _grad_report(m, g, m.loss_scale)
    dump = os.environ.get("MVD_GRAD_DUMP")
    if dump:  # development aid: the whole table, in the reference's registration order
        with open(dump, "w") as f:
            for rl, nr, n in rows:
                f.write(f"{rl:.3e} {nr:+.3e} {n}\n")
    rows.sort(reverse=True)
    for rl, nr, n in rows[:25]:
        print(f"[parity] worst grad {n}: relL2={rl:.2e} norm err {nr:.2e}")
    cond = [r for r in rows if r[2].startswith(("middle_conditions.", "output_conditions."))]
    rest = [r for r in rows if not r[2].startswith(("middle_conditions.", "output_conditions."))]
    print(f"[parity] gradients: {len(cond)} DepthTransformer tensors worst {max(r[0] for r in cond):.2e} median "
          f"{sorted(r[0] for r in cond)[len(cond) // 2]:.2e}; {len(rest)} other UNet tensors worst {max(r[0] for r in rest):.2e} "
          f"median {sorted(r[0] for r in rest)[len(rest) // 2]:.2e}")
    assert len(cond) == 170
    assert max(r[0] for r in cond) <= 5e-2, cond[0]  # measured 3.65e-2 (insensitive to the precision policy: ReLU-mask flips)
    assert sorted(r[0] for r in cond)[len(cond) // 2] <= 1.5e-2
    assert max(r[0] for r in rest) <= 7e-3, rest[0]  # measured 4.69e-3 (profiles/r05_z_train_grad_vs_precision_level.txt) x 1.5
    # gradient w.r.t. the frustum volumes (before the dropout): the entry point of the conditioner's backward
    for res_, d in m.last_dsrc.items():
        a, b, _ = gi.unpack_compare(d.cpu() / m.loss_scale, g, f"dsrc.{res_}")
        rl = ((a - b).norm() / (b.norm() + 1e-30)).item()
        print(f"[parity] d loss / d frustum volume {res_}: relL2={rl:.2e}")
        assert rl <= 5e-2  # downstream of the DepthTransformers' ReLU masks, see the module docstring
    # torch semantics: a second backward ACCUMULATES
    g1 = m.engine.flat_grads.clone()
    m.training_step(dev, prepared=prepared, **draws)
    hi = _unet_range(m.engine)
    assert torch.allclose(m.engine.flat_grads[:hi], 2 * g1[:hi], rtol=1e-6, atol=0)
    # the conditioner's scatters use unordered atomic adds: equal to rounding, not bit for bit
    d = (m.engine.flat_grads[hi:] - 2 * g1[hi:]).norm() / (2 * g1[hi:]).norm()
    assert g1[hi:].abs().max() > 0 and d <= 1e-5, d
    m.engine.close()


def test_training_step_full_width_gradients_vs_reference():
    """The FULL-WIDTH training step (916.9 M-parameter UNet, B = 2) against the reference's own training_step + loss.backward()
    run on the CPU (tools/make_goldens.py --only-train-full -> tests/golden/train_full.npz): loss, prediction and a sample of 26
    gradient tensors -- every block kind, every resolution level, both ends of the network, six DepthTransformer tensors.  Bounds
    from what the full-width step measures (tighter than at reduced width: wider reductions average the operand rounding)."""
    path = os.path.join(G, "train_full.npz")
    g = np.load(path)
    batch, x0, x_in, clip, ts, noise, ti, dr = _train_inputs(g)
    dev = {k: v.cuda() for k, v in batch.items()}
    prepared = (x0.cuda(), clip.cuda(), {"x": x_in.cuda()})
    N = int(g["N"])
    m = make_train_model(gi.FULL_UNET, VolumeConfig(num_views=N), N, workspace_gb=40.0, loss_scale=65536.0, recompute=True)
    m.train_conditioner = False  # the golden holds UNet gradients (frustum volumes as leaves)
    m.engine.zero_grad()
    loss = m.training_step(dev, prepared=prepared, time_steps=ts, noise=noise, target_index=ti, drop_random=dr)
    want = float(np.asarray(g["loss.full"])[0])
    rl = abs(float(loss) - want) / want
    print(f"[parity] full-width training loss {float(loss):.6f} vs reference {want:.6f}: rel {rl:.2e}")
    assert rl <= 1e-3
    compare(m.last_noise_predict, g, "noise_predict", rel=2e-3, mx=1e-2)
    rows = sorted(_grad_report(m, g, m.loss_scale), reverse=True)
    for rl_, nr, n in rows:
        print(f"[parity] full-width grad {n}: relL2={rl_:.2e} norm err {nr:.2e}")
    cond = [r for r in rows if r[2].startswith(("middle_conditions.", "output_conditions."))]
    rest = [r for r in rows if not r[2].startswith(("middle_conditions.", "output_conditions."))]
    assert len(cond) == 6 and len(rest) == 20
    # round 6: 1.5 x the measured worst (profiles/r05_z_pytest_full.log: DepthTransformer 1.63e-2, trunk 2.60e-3); was 5e-2 / 1e-2
    assert max(r[0] for r in cond) <= 2.5e-2, cond[0]
    assert max(r[0] for r in rest) <= 4e-3, rest[0]
    m.engine.close()


@pytest.mark.parametrize("cond_index,res_", [(0, 4), (2, 8), (5, 16), (9, 32)])
def test_depth_transformer_backward_exact_inputs(cond_index, res_):
    """One DepthTransformer's backward with the SAME input, context volume and output gradient on both sides (oracle autograd
    in fp32 on the CPU vs mvd_train_cond_backward): no forward-pass rounding in the inputs, so no mask flips -- what is left is
    the arithmetic of the backward itself (extended-precision fp16 MFMA operands, fp32 elsewhere)."""
    from oracle import mvd_oracle as O
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=4)
    m = make_train_model(ucfg, vcfg, 4, workspace_gb=6.0)
    W = gi.full_weights(ucfg, vcfg)
    pre = P + ("middle_conditions" if cond_index == 0 else f"output_conditions.{cond_index - 1}")
    keys = [k for k in W if k.startswith(pre + ".")]
    assert len(keys) == 17
    dim = W[pre + ".proj_in.0.weight"].shape[1]
    Cc = W[pre + ".proj_context.0.weight"].shape[1]
    B, D = 2, 48 * res_ // 32
    gen = torch.Generator().manual_seed(100 + cond_index)
    x = torch.randn(B, dim, res_, res_, generator=gen)
    ctx = torch.randn(B, Cc, D, res_, res_, generator=gen) * 0.7
    ctx[1] = 0  # a sample whose condition was dropped: all-zero volume
    dout = torch.randn(B, dim, res_, res_, generator=gen) * 1e-2
    Wl = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in W.items()}
    xx, cc = x.clone().requires_grad_(True), ctx.clone().requires_grad_(True)
    O.depth_transformer(Wl, pre, xx, cc).backward(dout)
    m.engine.zero_grad()
    dx, dc = m.engine.train_cond_backward(cond_index, x, ctx, dout)
    errs = {"dx": ((dx.cpu() - xx.grad).norm() / xx.grad.norm()).item(), "dctx": ((dc.cpu() - cc.grad).norm() / cc.grad.norm()).item()}
    for k in keys:
        got = m.engine.param_view(k, grad=True).cpu()
        errs[k[len(pre) + 1:]] = ((got - Wl[k].grad).norm() / (Wl[k].grad.norm() + 1e-30)).item()
    worst = max(errs, key=errs.get)
    print(f"[parity] DepthTransformer {cond_index} backward from exact inputs: worst {worst} {errs[worst]:.2e}, dx {errs['dx']:.2e}, dctx {errs['dctx']:.2e}")
    assert errs[worst] <= 2e-3, errs
    m.engine.close()


def test_recompute_equals_keep_all_and_is_reproducible():
    """Activation checkpointing per block (the reference's use_checkpoint: True, diffusionmodules/util.py:102-148) re-runs the
    same kernels on the same inputs: bit-identical gradients to the keep-everything tape; and a repeated step is bit-identical."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, loss_scale=65536.0, recompute=False)
    outs = []
    for rec in (False, True, True):
        m.recompute = rec
        m.engine.zero_grad()
        loss = m.training_step(dev, prepared=prepared, **draws)
        torch.cuda.synchronize()
        outs.append((float(loss), m.engine.flat_grads.clone(), m.last_noise_predict.clone()))
    hi = _unet_range(m.engine)
    assert outs[0][0] == outs[1][0] == outs[2][0]
    assert torch.equal(outs[1][1][:hi], outs[2][1][:hi]) and torch.equal(outs[1][2], outs[2][2])
    assert torch.equal(outs[0][1][:hi], outs[1][1][:hi])
    for a, b_ in ((outs[1][1], outs[2][1]), (outs[0][1], outs[1][1])):  # conditioner gradients: atomic scatters, equal to rounding
        assert ((a[hi:] - b_[hi:]).norm() / b_[hi:].norm()).item() <= 1e-5
    m.engine.close()


def test_training_step_full_width_finite_and_reproducible():
    """The 917 M-parameter UNet (configs/facescape.yaml's widths), B = 2 samples: one training step with per-block recompute --
    loss and every gradient finite at the default loss scale, no optimiser step skipped, a repeated step bit-identical, the
    keep-all tape gives the same bits."""
    from morphablediffusion_amd import synthetic
    N, B = 4, 2
    ucfg, vcfg = gi.FULL_UNET, VolumeConfig(num_views=N)
    m = make_train_model(ucfg, vcfg, N, workspace_gb=40.0, recompute=True)
    m.model.drop_conditions = False  # configs/facescape.yaml:11
    b0 = synthetic.make_batch(N, "perspective", 600, mesh_seed=1)
    batch = {k: v.repeat(B, *([1] * (v.dim() - 1))).clone().cuda() for k, v in b0.items()}
    gen = torch.Generator().manual_seed(5)
    prepared = ((torch.randn(B, N, 4, 32, 32, generator=gen) * 0.8).cuda(), torch.randn(B, 1, 768, generator=gen).cuda(),
                {"x": (torch.randn(B, 4, 32, 32, generator=gen) * 0.18215).cuda()})
    draws = dict(time_steps=torch.tensor([301, 777]), noise=torch.randn(B, N, 4, 32, 32, generator=gen),
                 target_index=torch.tensor([[1], [3]]))
    outs = []
    for rec in (True, True, False):
        m.recompute = rec
        m.engine.zero_grad()
        loss = m.training_step(batch, prepared=prepared, **draws)
        torch.cuda.synchronize()
        outs.append((float(loss), m.engine.flat_grads.clone()))
    g = outs[0][1]
    assert np.isfinite(outs[0][0]) and torch.isfinite(g).all()
    tab = m.engine.param_table
    nz = sum(1 for k, (o, n, s_) in tab.items() if k.startswith(P) and g[o:o + n].abs().max() > 0)
    total = sum(1 for k in tab if k.startswith(P))
    print(f"[property] full-width training step: loss {outs[0][0]:.4f}, {nz} of {total} UNet tensors with a non-zero gradient, "
          f"|g|max {g.abs().max().item() / m.loss_scale:.3e} (x loss scale {m.loss_scale:.0f} = {g.abs().max().item():.3e})")
    assert total - nz == 16 * 4  # attn2.to_q / to_k / norm2.weight / norm2.bias of the 16 SpatialTransformers: exactly zero
    hi = _unet_range(m.engine)
    assert g[hi:].abs().max() > 0  # spatial_volume / time_embed gradients are there too
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1][:hi], outs[1][1][:hi])
    assert outs[2][0] == outs[0][0] and torch.equal(outs[2][1][:hi], outs[0][1][:hi])
    m.learning_rate = 5e-5
    (opt,), _ = m.configure_optimizers()
    opt.step()
    assert opt.steps_skipped == 0 and opt.steps_done == 1
    m.engine.close()


def test_adamw_step_and_repack():
    """configure_optimizers (morphable_diffusion.py:627-646): parameter groups, LambdaLR, and one optimiser step against
    torch.optim.AdamW on copies of the same parameters / gradients; the re-packed engine equals a fresh load of the updated
    state_dict bit for bit."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    m = make_train_model(ucfg, vcfg, N, loss_scale=65536.0, recompute=True)
    m.learning_rate = 5e-5
    (opt,), (sched,) = m.configure_optimizers()
    assert [len(gr["params"]) for gr in opt.param_groups][0] == len(g["grad_names"])
    assert abs(opt.param_groups[0]["lr"] - 5e-5 * 0.02) < 1e-12 and abs(opt.param_groups[1]["lr"] - 5e-4 * 0.02) < 1e-12  # warm-up f_start
    opt.zero_grad()
    m.training_step(dev, prepared=prepared, **draws)
    eng = m.engine
    p0, g0 = eng.flat_params.clone(), eng.flat_grads.clone() / m.loss_scale
    # torch's AdamW on the same numbers (one tensor per group range: the UNet keys sort before spatial_volume / time_embed)
    tab = eng.param_table
    lo = min(o for k, (o, n, s) in tab.items() if not k.startswith(P))
    ref_p = [p0[:lo].clone().requires_grad_(True), p0[lo:].clone().requires_grad_(True)]
    ref_p[0].grad, ref_p[1].grad = g0[:lo].clone(), g0[lo:].clone()
    ref = torch.optim.AdamW([{"params": [ref_p[0]], "lr": opt.param_groups[0]["lr"]}, {"params": [ref_p[1]], "lr": opt.param_groups[1]["lr"]}])
    ref.step()
    opt.step()
    sched["scheduler"].step()
    assert opt.steps_done == 1 and opt.steps_skipped == 0
    want = torch.cat([ref_p[0].detach(), ref_p[1].detach()])
    # only real parameter slots are compared (the arena pads each tensor to 64 floats)
    worst = 0.0
    for k, (o, n, s) in tab.items():
        d = (eng.flat_params[o:o + n] - want[o:o + n]).abs().max().item()
        worst = max(worst, d / (want[o:o + n].abs().max().item() + 1e-12))
    print(f"[parity] AdamW step vs torch.optim.AdamW: worst normalised difference {worst:.2e}")
    assert worst <= 2e-6
    assert (eng.flat_params - p0).abs().max() > 0
    # the re-packed weights are what a fresh load of the updated state_dict gives
    x, t, ctx, sd = gi.unet_inputs(ucfg, Bv=2)
    sdc = {k: v.cuda() for k, v in sd.items()}
    out_repacked = eng.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), sdc)
    W2 = {k: eng.param_view(k).detach().cpu().clone() for k in tab}
    for k, v in gi.full_weights(ucfg, vcfg).items():
        W2.setdefault(k, v)  # BatchNorm running statistics (buffers, not in the arena)
    m2 = make_train_model(ucfg, vcfg, N, train_mode=False)
    m2.load_state_dict(W2)
    out_fresh = m2.engine.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), sdc)
    assert torch.equal(out_repacked, out_fresh)
    # an overflowing gradient skips the update and halves the loss scale (GradScaler's rule)
    eng.flat_grads[5] = float("inf")
    before = eng.flat_params.clone()
    opt.step()
    assert opt.steps_skipped == 1 and m.loss_scale == 32768.0 and torch.equal(eng.flat_params, before)
    m.engine.close()
    m2.engine.close()


def test_optimizer_state_round_trip_resumes_bit_identically():
    """the Adam moments, the bias-correction step and the loss scale live outside torch's Optimizer.state -- a
    checkpoint made of model.state_dict() + optimizer.state_dict() must resume so that update N+1 is the one an uninterrupted
    run makes, bit for bit (torch.optim.AdamW round-trips its state the same way, morphable_diffusion.py:642)."""
    import io
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)

    def one_step(m, opt):
        opt.zero_grad()
        m.training_step(dev, prepared=prepared, **draws)
        opt.step()

    m = make_train_model(ucfg, vcfg, N, loss_scale=65536.0, recompute=True)
    m.learning_rate = 5e-5
    (opt,), _ = m.configure_optimizers()
    one_step(m, opt)
    one_step(m, opt)
    m.loss_scale = 16384.0  # as if two overflows had happened: must survive the round trip
    buf = io.BytesIO()
    torch.save({"state_dict": m.state_dict(), "optimizer": opt.state_dict()}, buf)
    assert m.global_step == 2
    one_step(m, opt)  # update 3 of the uninterrupted run
    want = m.engine.flat_params.clone()
    m.engine.close()
    # resume in a fresh model / engine / optimiser from the checkpoint bytes
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    assert ck["optimizer"]["arena"]["step"] == 2 and ck["optimizer"]["arena"]["exp_avg"].abs().max() > 0
    m2 = make_train_model(ucfg, vcfg, N, loss_scale=65536.0, recompute=True)
    m2.load_state_dict(ck["state_dict"])
    m2.learning_rate = 5e-5
    (opt2,), _ = m2.configure_optimizers()
    opt2.load_state_dict(ck["optimizer"])
    assert opt2.steps_done == 2 and m2.loss_scale == 16384.0
    one_step(m2, opt2)
    hi = _unet_range(m2.engine)
    assert torch.equal(m2.engine.flat_params[:hi], want[:hi]), "update 3 after the resume differs from the uninterrupted run"
    # (the conditioner's scatter adjoints use hardware fp32 atomics: its parameters agree to rounding, not bit for bit)
    assert torch.allclose(m2.engine.flat_params[hi:], want[hi:], rtol=1e-5, atol=1e-7)
    # a checkpoint without the arenas (written by torch.optim.AdamW itself, or before ArenaAdamW carried them): no KeyError any
    # more -- the moments restart from zero with a warning when the per-parameter state does not line up ...
    with pytest.warns(UserWarning, match="restart from zero"):
        opt2.load_state_dict({k: v for k, v in ck["optimizer"].items() if k != "arena"})
    assert opt2.steps_done == 0 and float(m2.engine.flat_m.abs().max()) == 0.0 and float(m2.engine.flat_v.abs().max()) == 0.0
    # ... and are adopted when it does: a torch.optim.AdamW-style state over this optimiser's own parameters
    mine = [p_ for g_ in opt2.param_groups for p_ in g_["params"]]
    fake_state = {i: {"step": torch.tensor(7.0), "exp_avg": torch.full_like(p_, 0.25), "exp_avg_sq": torch.full_like(p_, 0.5)}
                  for i, p_ in enumerate(mine)}
    groups, k0 = [], 0
    for g_ in ck["optimizer"]["param_groups"]:
        n = len(g_["params"])
        groups.append(dict(g_, params=list(range(k0, k0 + n))))
        k0 += n
    with pytest.warns(UserWarning, match="adopted"):
        opt2.load_state_dict({"state": fake_state, "param_groups": groups})
    assert opt2.steps_done == 7
    covered = sum(p_.numel() for p_ in mine)
    assert abs(float(m2.engine.flat_m.sum()) - 0.25 * covered) <= 1e-3 * covered
    # the overflow check follows the CURRENT loss scale (it was fixed at construction)
    m2.loss_scale = 1.0
    assert opt2.dynamic_scale is False
    m2.loss_scale = 4096.0
    assert opt2.dynamic_scale is True
    m2.engine.close()


def test_training_step_default_prepare_path():
    """training_step(batch) with no overrides: prepare() VAE-encodes the target views itself, time steps are drawn
    before it, sampling afterwards still works on the same context."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, loss_scale=65536.0)

    class FakePosterior:
        def __init__(self, x):
            self.m = torch.nn.functional.avg_pool2d(x, 8)[:, :1].repeat(1, 4, 1, 1)

        def sample(self):
            return self.m + 0.1 * torch.randn(self.m.shape).to(self.m.device)

        def mode(self):
            return self.m

    class FakeVAE:
        def encode(self, x):
            return FakePosterior(x)

    class FakeClip:
        def encode(self, x):
            return torch.randn(x.shape[0], 1, 768).to(x.device)

    m.first_stage_model, m.clip_image_encoder = FakeVAE(), FakeClip()
    B = 2
    batch = {k: v[:B] for k, v in dev.items()}
    batch["target_image"] = torch.rand(B, N, 256, 256, 3, device="cuda") * 2 - 1
    batch["input_image"] = torch.rand(B, 256, 256, 3, device="cuda") * 2 - 1
    torch.manual_seed(3)
    loss = m.training_step(batch)
    assert torch.isfinite(loss) and m.engine.flat_grads.abs().max() > 0
    m.eval()
    x = m.sampler.denoise_apply(torch.randn(B, N, 4, 32, 32, device="cuda"), {"x": prepared[2]["x"][:B]}, prepared[1][:B],
                                torch.full((B,), 501, device="cuda", dtype=torch.long), 25, 2.0, batch_view_num=N, batch=batch)
    assert torch.isfinite(x).all()
    m.engine.close()


def test_drop_scheme_thresholds():
    """UNetWrapper.get_drop_scheme (morphable_diffusion.py:84-93): the four bands of the uniform draw."""
    from morphablediffusion_amd.model import UNetWrapper
    w = UNetWrapper.__new__(UNetWrapper)
    w.drop_scheme = "default"
    u = torch.tensor([0.0, 0.05, 0.051, 0.1, 0.101, 0.15, 0.151, 0.2, 0.201, 0.99])
    dc, dv, dx, da = w.get_drop_scheme(10, "cpu", u)
    assert da.tolist() == [True, True] + [False] * 8
    assert dx.tolist() == [False, False, True, True] + [False] * 6
    assert dv.tolist() == [False] * 4 + [True, True] + [False] * 4
    assert dc.tolist() == [False] * 6 + [True, True] + [False] * 2
    w.drop_scheme = "other"
    with pytest.raises(NotImplementedError):
        w.get_drop_scheme(2, "cpu")


@pytest.mark.parametrize("mesh", ["distinct voxels", "duplicate voxels"])
def test_conditioner_backward_stages_vs_oracle_autograd(mesh):
    """mvd_train_conditioner_backward for one sample against fp32 autograd through the oracle's conditioner, fed the SAME random
    dL/d(frustum volumes): the intermediate gradients it exposes (d 32^3 volume, d step embedding) and every parameter gradient
    of spatial_volume.* / time_embed.*.  The frustum network runs on fp16 operands (forward and backward: d volume 4e-4); the 2-D
    encoder is re-computed in extended precision because the sparse CNN behind it has nine BatchNorm + ReLU layers whose masks
    are re-derived from its output (an fp16-rounded encoder moved these gradients by 4-11e-2, measured; now the median is 7e-4).
    What remains at 1-2e-2 are bias gradients -- sums over thousands of rows with heavy cancellation, which amplify the 4e-4 of
    the incoming gradient.  Bounds: worst 3e-2, median 2e-3 (measured values printed).
    "duplicate voxels": several vertices per 5 mm voxel (what real FLAME meshes have) -- the later ones' rows copy the first
    one's, so the gather-form data gradient of the sparse CNN folds their output gradients into the representative's row and
    leaves their own input gradient zero (k_cond_bwd.hip: sparse_fold_dups_kernel)."""
    from morphablediffusion_amd import synthetic
    from oracle import mvd_oracle as O
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    m = make_train_model(ucfg, vcfg, N, workspace_gb=8.0)
    W = gi.full_weights(ucfg, vcfg)
    if mesh == "distinct voxels":
        batch = synthetic.make_batch(N, "perspective", 500, mesh_seed=1)
    else:
        from morphablediffusion_amd import batch as BT
        verts = synthetic.ellipsoid_mesh(900, 3, radii=(0.09, 0.11, 0.10), dedup=False)
        batch = BT.build_batch(torch.zeros(256, 256, 3), verts, num_views=N)
        coord = batch["coord"][0]
        key = (coord[:, 0].long() * 4096 + coord[:, 1].long()) * 4096 + coord[:, 2].long()
        ndup = key.numel() - torch.unique(key).numel()
        assert ndup > 50, ndup
        batch = {k: v for k, v in batch.items() if torch.is_tensor(v)}
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(1, N, 4, 32, 32, generator=gen) * 0.8
    ts = torch.tensor([421])
    tidx = 2
    keys = [k for k in W if k.startswith(("spatial_volume.", "time_embed.")) and not k.endswith(("running_mean", "running_var"))]
    Wl = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in W.items()}
    v_embed = O.viewpoint_embedding(batch)
    t_embed = O.embed_time(Wl, ts, vcfg.time_dim)
    t_embed.retain_grad()
    sv = O.construct_spatial_volume(Wl, vcfg, x, t_embed, v_embed, batch, train=True)
    sv.retain_grad()
    fd = O.construct_view_frustum_volume(Wl, vcfg, sv, t_embed, v_embed, torch.tensor([[tidx]]), batch)
    dsrc = {k: torch.randn(v.shape, generator=gen) * (0.5 ** i) for i, (k, v) in enumerate(sorted(fd.items(), reverse=True))}
    sum((fd[k] * dsrc[k]).sum() for k in fd).backward()
    dev = {k: v.cuda() for k, v in batch.items()}
    m.spatial_volume._set_sample(dev, 0)
    m.engine.zero_grad()
    dvol, dfused, dfeats, dtemb = m.engine.train_conditioner_backward(x[0].cuda(), int(ts[0]), v_embed[0].cuda(), tidx,
                                                                      {k: v.cuda() for k, v in dsrc.items()}, debug=True)

    def rel(a, b):
        return ((a.detach().cpu() - b).norm() / (b.norm() + 1e-30)).item()

    e_vol, e_t = rel(dvol, sv.grad[0]), rel(dtemb, t_embed.grad[0])
    print(f"[parity] conditioner backward: d volume {e_vol:.2e}, d step embedding {e_t:.2e}, |d fused| {dfused.norm().item():.3e}, "
          f"|d feats| {dfeats.norm().item():.3e}")
    errs = {}
    for k in keys:
        got = m.engine.param_view(k, grad=True).cpu()
        assert torch.isfinite(got).all(), k
        errs[k] = rel(got, Wl[k].grad)
    if os.environ.get("MVD_GRAD_DUMP"):
        with open(os.environ["MVD_GRAD_DUMP"], "w") as f:
            for k in keys:
                f.write(f"{errs[k]:.3e} {Wl[k].grad.norm().item():.3e} {k}\n")
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:8]
    for k, e in worst:
        print(f"[parity] conditioner grad {k}: relL2={e:.2e}")
    med = sorted(errs.values())[len(errs) // 2]
    print(f"[parity] conditioner backward: {len(errs)} parameter tensors, worst {worst[0][1]:.2e}, median {med:.2e}")
    assert e_vol <= 2e-3 and e_t <= 3e-3
    # (the duplicate-voxel mesh: the FiLM projections of the 2-D encoder measured 4.0e-2 with both forms of the sparse backward)
    assert worst[0][1] <= (3e-2 if mesh == "distinct voxels" else 6e-2) and med <= 2e-3, (worst[0], med)
    m.engine.close()


def test_train_mode_batchnorm_register_form_is_bit_identical(monkeypatch):
    """The sparse CNN's train-mode BatchNorm1d + ReLU with a thread's rows held in registers (one read of the column, all loads in
    flight) against the looped three-pass kernel (MVD_BN_LOOP=1): the same partial sums in the same order -- the 32^3 volume
    built in train mode (nine such layers, masks drawn from their statistics) must be bit for bit the same."""
    from morphablediffusion_amd import synthetic
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    m = make_train_model(ucfg, vcfg, N, workspace_gb=8.0)
    batch = {k: v.cuda() for k, v in synthetic.make_batch(N, "perspective", 5023, mesh_seed=2).items()}
    gen = torch.Generator().manual_seed(3)
    x = (torch.randn(1, N, 4, 32, 32, generator=gen) * 0.8).cuda()
    v_embed = torch.zeros(1, N, 4).cuda()
    v_embed[..., 2] = 1.0
    t_embed = m.embed_time(torch.tensor([300]).cuda())
    m.train()
    vols = {}
    for form in ("register", "loop"):
        if form == "loop":
            monkeypatch.setenv("MVD_BN_LOOP", "1")
        else:
            monkeypatch.delenv("MVD_BN_LOOP", raising=False)
        m.spatial_volume.invalidate()
        vols[form] = m.spatial_volume.construct_spatial_volume(x, t_embed, v_embed, batch).cpu().clone()
    assert torch.isfinite(vols["register"]).all() and vols["register"].abs().max() > 0
    assert torch.equal(vols["register"], vols["loop"])
    m.engine.close()


def test_sparse_cnn_matrix_core_form_equals_site_form(monkeypatch):
    """The sparse voxel CNN on the fp32 matrix cores (tiled gather-GEMM forward, gather-form data gradient through the flipped /
    inverse tables with duplicate rows folded, pair-list weight gradient) against the one-site-per-workgroup kernels with the
    scatter-form data gradient (MVD_SPARSE_VALU=1), on a mesh with duplicate voxels, train-mode BatchNorm: the same fp32
    arithmetic in another summation order.  The 32^3 volume then differs in the last bits, the fp16 frustum network behind it
    re-rounds it, and dL/d(volume) -- which does not depend on the sparse backward at all -- already differs by 3e-4 between the
    two runs: that is the floor of this comparison (measured: volume / fused / step-embedding gradients 3.0-3.3e-4, the 149
    parameter gradients worst 1.0e-3).  Bounds 1e-3 / 5e-3; a wrong fold or table showed as 0.6."""
    from morphablediffusion_amd import synthetic, batch as BT
    N = 4
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    verts = synthetic.ellipsoid_mesh(900, 3, radii=(0.09, 0.11, 0.10), dedup=False)
    batch = {k: v.cuda() for k, v in BT.build_batch(torch.zeros(256, 256, 3), verts, num_views=N).items() if torch.is_tensor(v)}
    gen = torch.Generator().manual_seed(11)
    x = (torch.randn(N, 4, 32, 32, generator=gen) * 0.8).cuda()
    v_embed = torch.zeros(N, 4).cuda()
    v_embed[:, 2] = 1.0
    res = {}
    for form in ("site", "matrix-core"):
        if form == "site":
            monkeypatch.setenv("MVD_SPARSE_VALU", "1")
        else:
            monkeypatch.delenv("MVD_SPARSE_VALU", raising=False)
        m = make_train_model(ucfg, vcfg, N, workspace_gb=8.0)
        m.spatial_volume._set_sample(batch, 0)
        if form == "site":  # the shapes of dL/d(frustum volumes) from a forward of the frustum stage
            t_embed = m.embed_time(torch.tensor([421]).cuda())
            sv = m.spatial_volume.construct_spatial_volume(x[None], t_embed, v_embed[None], batch)
            vf, _ = m.spatial_volume.construct_view_frustum_volume(sv, t_embed, v_embed[None], torch.tensor([[2]]).cuda(), batch)
            dsrc = {k: (torch.randn(v.shape, generator=gen) * 0.5).cuda() for k, v in sorted(vf.items())}
        m.engine.zero_grad()
        dvol, dfused, dfeats, dtemb = m.engine.train_conditioner_backward(x, 421, v_embed, 2, dsrc, debug=True)
        keys = [k for k in m.engine.param_table if k.startswith(("spatial_volume.", "time_embed."))]
        res[form] = (dvol.cpu(), dfused.cpu(), dtemb.cpu(), {k: m.engine.param_view(k, grad=True).cpu().clone() for k in keys})
        m.engine.close()

    def rel(a, b):
        return ((a - b).norm() / (b.norm() + 1e-30)).item()

    a, b = res["matrix-core"], res["site"]
    e = [rel(a[i], b[i]) for i in range(3)]
    errs = sorted(((rel(a[3][k], b[3][k]), k) for k in b[3] if b[3][k].norm() > 0), reverse=True)
    print(f"[property] sparse CNN, matrix-core vs site form: d volume {e[0]:.2e}, d fused {e[1]:.2e}, d step embedding {e[2]:.2e}, "
          f"{len(errs)} parameter gradients worst {errs[0][0]:.2e} ({errs[0][1]})")
    assert max(e) <= 1e-3 and errs[0][0] <= 5e-3, (e, errs[:3])


def test_gradient_buckets_cover_the_unet_once_and_are_final_at_their_events():
    """mvd_train_grad_bucket*: the UNet's gradients as buckets in the order the backward pass completes them (one per chain of
    blocks + one for what the stacked end-of-step kernels write).  Their ranges cover model.diffusion_model.* exactly once, and
    every range is FINAL when its event is recorded: with the snapshot hook the engine copies each bucket right behind its event,
    and the copy equals the gradients at the end of the step bit for bit (a range written after its event would differ).
    Checked on the step that builds the ranges and on the next one (cached ranges)."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, recompute=True)
    eng = m.engine
    hi = _unet_range(eng)
    for rep in range(2):
        eng.zero_grad()
        snap = torch.full_like(eng.flat_grads, float("nan"))
        eng.set_bucket_snapshot(snap)
        m.training_step(dev, prepared=prepared, **draws)
        torch.cuda.synchronize()
        eng.set_bucket_snapshot(None)
        buckets = eng.grad_buckets()
        spans = sorted((o, o + n) for b in buckets for o, n in b)
        assert spans[0][0] == 0 and spans[-1][1] == hi, (spans[0], spans[-1], hi)
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])), "gaps or overlaps between bucket ranges"
        assert len(buckets) == 26, [len(b) for b in buckets]  # 12 output chains, the middle block, 12 input chains, the rest
        final = eng.flat_grads[:hi]
        assert torch.isfinite(snap[:hi]).all() and final.abs().max() > 0
        assert torch.equal(snap[:hi], final), f"step {rep}: a bucket was not final at its event"
        assert torch.isnan(snap[hi:]).all()  # nothing outside the UNet range is in a bucket
    print(f"[property] gradient buckets: {len(buckets)} buckets, {sum(len(b) for b in buckets)} ranges, final at their events")
    eng.close()


def test_training_step_conditioner_gradients_vs_reference():
    """The complete training step: after the UNet's backward, the conditioner's (per sample, from dL/d(its frustum volumes))
    fills the gradients of spatial_volume.* and time_embed.* -- all 149 tensors against the reference's loss.backward().
    They sit downstream of the DepthTransformers' gradient w.r.t. the volumes (2.2-2.8e-2, see the module docstring), so the
    bound is 6e-2 (measured values printed)."""
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, recompute=True)
    m.engine.zero_grad()
    m.training_step(dev, prepared=prepared, **draws)
    rows = []
    for n, want_norm in zip([str(x) for x in g["cond_names"]], g["cond_norms"]):
        got = m.engine.param_view(n, grad=True).detach().float().cpu() / m.loss_scale
        assert torch.isfinite(got).all() and want_norm > 0, n
        a, b, _ = gi.unpack_compare(got, g, "gradc." + n)
        rows.append((((a - b).norm() / (b.norm() + 1e-30)).item(), n))
    rows.sort(reverse=True)
    for rl, n in rows[:8]:
        print(f"[parity] conditioner grad {n}: relL2={rl:.2e}")
    print(f"[parity] conditioner gradients in the full step: {len(rows)} tensors, worst {rows[0][0]:.2e}, median {rows[len(rows) // 2][0]:.2e}")
    assert rows[0][0] <= 6e-2, rows[0]
    m.engine.close()


def test_checkpoint_export_tracks_batchnorm_and_reloads_bit_identically():
    """What a fine-tuning run saves: ``state_dict()`` of a training-mode model = the loaded checkpoint with the masters and the
    sparse CNN's BatchNorm running statistics as training left them (nn.BatchNorm1d(momentum=0.01), network.py:105: one update
    per sample and train-mode forward).  Checked: untouched before training; running statistics after one training step against
    the oracle's chained update; loading the export into a fresh inference model reproduces the trained model's eval forward
    bit for bit."""
    from morphablediffusion_amd.model import SyncMultiviewDiffusion
    from oracle import mvd_oracle as O
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    ucfg, vcfg = gi.SMALL_UNET, VolumeConfig(num_views=N)
    W = gi.full_weights(ucfg, vcfg)
    m = make_train_model(ucfg, vcfg, N, loss_scale=65536.0, recompute=True)
    sd0 = m.state_dict()
    assert list(sd0) == list(W) and all(torch.equal(sd0[k].cpu().float(), W[k].float()) for k in W)
    m.learning_rate = 5e-5
    (opt,), _ = m.configure_optimizers()
    opt.zero_grad()
    m.training_step(dev, prepared=prepared, **draws)
    opt.step()
    sd1 = m.state_dict()
    assert list(sd1) == list(W) and all(sd1[k].shape == W[k].shape and sd1[k].dtype == W[k].dtype for k in W)
    changed = [k for k in W if not torch.equal(sd1[k].cpu(), W[k])]
    assert any(k.startswith(P) for k in changed) and any(k.startswith("spatial_volume.") and k.endswith(".weight") for k in changed)
    # running statistics: the oracle's train-mode forward on the same noisy latents, B samples in sequence
    x0, ts, noise = prepared[0].cpu(), draws["time_steps"], draws["noise"]
    x_noisy = m.add_noise(x0.cuda(), ts.cuda(), noise.cuda())[0].cpu()
    batch = {k: v.cpu() for k, v in dev.items()}
    upd = {}
    O.construct_spatial_volume(W, vcfg, x_noisy, m.embed_time(ts.cuda()).cpu(), m.get_viewpoint_embedding(batch), batch, train=True,
                               bn_update=upd)
    assert len(upd) == 18  # nine BatchNorm layers
    worst = 0.0
    for k, want in upd.items():
        got = sd1[k].cpu()
        assert not torch.equal(got, W[k])
        step = (want - W[k]).abs().max().item()  # the size of the update itself: the error is measured against it
        worst = max(worst, (got - want).abs().max().item() / step)
    print(f"[parity] BatchNorm running statistics after one training step ({x_noisy.shape[0]} samples): worst error / update = {worst:.2e}")
    assert worst <= 2e-3
    assert int(m.engine.lib.mvd_train_bn_calls(m.engine._ctx)) == x_noisy.shape[0]
    # a fresh inference model loaded from the export == the trained model in eval mode (re-packed in place), bit for bit
    m.eval()
    v_embed, t_embed = m.get_viewpoint_embedding(dev), m.embed_time(ts.cuda())
    sv_a = m.spatial_volume.construct_spatial_volume(x_noisy.cuda(), t_embed, v_embed, dev)
    x, t, ctx, sdict = gi.unet_inputs(ucfg, Bv=2)
    sdc = {k: v.cuda() for k, v in sdict.items()}
    out_a = m.engine.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), sdc)
    m2 = make_train_model(ucfg, vcfg, N, train_mode=False)
    m2.load_state_dict({k: v.cpu() for k, v in sd1.items()})
    m2.eval()
    sv_b = m2.spatial_volume.construct_spatial_volume(x_noisy.cuda(), t_embed, v_embed, dev)
    out_b = m2.engine.unet_forward(x.cuda(), t.cuda(), ctx.cuda(), sdc)
    assert torch.equal(sv_a, sv_b) and torch.equal(out_a, out_b)
    m.engine.close()
    m2.engine.close()
