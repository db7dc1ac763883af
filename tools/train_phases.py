"""Where one training step's time goes (development aid): the phases of SyncMultiviewDiffusion.training_step + optimiser, each
bracketed by device synchronisation.  python tools/train_phases.py [B]   (FRESH=1: new batch tensors every step, KEEP=1: no
activation recompute, PROFILE_COND=1: ten extra conditioner backward passes for rocprofv3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import unet_kwargs
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = 16
ucfg, vcfg = UNetConfig(), VolumeConfig(num_views=N)
m = SyncMultiviewDiffusion(unet_config={"target": "x.DepthWiseAttention", "params": unet_kwargs(ucfg)}, finetune_unet=True, view_num=N,
                           image_size=256, workspace_gb=96.0, train_mode=True, recompute=os.environ.get("KEEP") is None)
m.load_state_dict(seeded_state_dict(full_manifest(ucfg, vcfg), 7))
(opt,), _ = m.configure_optimizers()
b0 = synthetic.make_batch(N, "perspective", 5023, mesh_seed=1)
batch = {k: v.repeat(B, *([1] * (v.dim() - 1))).clone().cuda() for k, v in b0.items()}
g = torch.Generator().manual_seed(1)
x0 = (torch.randn(B, N, 4, 32, 32, generator=g) * 0.8).cuda()
clip = torch.randn(B, 1, 768, generator=g).cuda()
x_in = (torch.randn(B, 4, 32, 32, generator=g) * 0.18215).cuda()
ts = torch.randint(0, 1000, (B,), generator=g)
noise = torch.randn(B, N, 4, 32, 32, generator=g).cuda()
ti = torch.randint(0, N, (B, 1), generator=g)
T = {}


def tick(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return r


for it in range(2 if os.environ.get('PROFILE_COND') else 4):
    if it == 1:
        T.clear()
    if os.environ.get("MVD_LAYER_TIMING"):
        print("[step-begin]", file=sys.stderr, flush=True)  # tools/layer_agg.py aggregates the GEMM lines of the last step
    dev = "cuda"
    tsd, tid = ts.cuda(), ti.cuda()
    if os.environ.get("FRESH"):  # a new batch per step (fresh tensors): the per-sample tables are rebuilt inside the step
        batch = {k: v.clone() for k, v in batch.items()}
    tick("zero_grad", opt.zero_grad)
    x_noisy, nz = m.add_noise(x0, tsd, noise)
    m.train()
    v_embed = m.get_viewpoint_embedding(batch)
    t_embed = m.embed_time(tsd)
    sv = tick("cond fwd: spatial volume", lambda: m.spatial_volume.construct_spatial_volume(x_noisy, t_embed, v_embed, batch))
    clip_, vf, xc = tick("cond fwd: frustum", lambda: m.get_target_view_feats(x_in, sv, clip, t_embed, v_embed, tid, batch))
    ar = torch.arange(B, device=dev)[:, None]
    target = nz[ar, tid][:, 0].contiguous()
    x_t = x_noisy[ar, tid][:, 0]
    pred, loss, dsrc = tick("unet fwd+bwd", lambda: m.model.train_step(x_t, tsd, clip_, vf, xc, target, loss_scale=m.loss_scale, recompute=m.recompute))

    def cb():
        for bi in range(B):
            m.spatial_volume._set_sample(batch, bi)
        if os.environ.get("PER_SAMPLE"):  # the one-sample entry, sample by sample (the frustum network at batch 1)
            for bi in range(B):
                m.spatial_volume._set_sample(batch, bi)
                m.engine.train_conditioner_backward(x_noisy[bi], int(ts[bi]), v_embed[bi], int(ti[bi, 0]), {k: v[bi:bi + 1] for k, v in dsrc.items()})
        else:
            m.engine.train_conditioner_backward_batch(list(range(B)), x_noisy, ts.tolist(), v_embed, ti[:, 0].tolist(), dsrc)
    tick("cond bwd", cb)
    tick("adamw", lambda: m.engine.lib.mvd_train_adamw_step(m.engine._ctx, *[__import__("ctypes").c_float(v) for v in (1e-6, 1e-5, 0.9, 0.999, 1e-8, 0.01)], it + 1, __import__("ctypes").c_float(1.0 / m.loss_scale), 1, None, None))
    tick("repack", m.engine.repack)
if os.environ.get("PROFILE_COND"):  # rocprofv3 aid: 10 more conditioner backward passes, nothing else
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        cb()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"conditioner backward x {10 * B}: host enqueue {1e3 * (t1 - t0) / (10 * B):.2f} ms per sample, "
          f"with the device drained {1e3 * (t2 - t0) / (10 * B):.2f} ms per sample")
tot = sum(T.values())
for k, v in T.items():
    print(f"{k:28s} {1e3 * v / max(1, it):8.2f} ms/step  {100 * v / tot:5.1f} %")
print(f"{'total':28s} {1e3 * tot / max(1, it):8.2f} ms/step  (B = {B}, recompute = {m.recompute})")
