"""hipGraph evaluation: ONE denoising step (fixed DDIM index) captured with torch.cuda.CUDAGraph and replayed, next to the
same step launched eagerly.  python tools/graph_step.py [--simulate-gpus G] [--steps K]
The capture bakes the step's scalar arguments (schedule coefficients, timestep) into the graph, so this is a measurement of
the launch path only, not a sampling loop."""
import argparse, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import unet_kwargs
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--simulate-gpus", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
dev, N = "cuda:0", 16
ucfg, vcfg = UNetConfig(image_size=32), VolumeConfig(num_views=N, projection="perspective", input_image_size=256)
W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)
model = SyncMultiviewDiffusion(unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": unet_kwargs(ucfg)},
                               projection="perspective", view_num=N, image_size=256, cfg_scale=2.0, device=dev, workspace_gb=48.0)
model.load_state_dict(W)
model.eval()
sampler = model.sampler
if args.simulate_gpus:
    sampler.simulate_world = args.simulate_gpus
lo, hi = sampler.view_range(N)
batch = {k: v.to(dev) for k, v in synthetic.make_batch(N, "perspective", 5023, mesh_seed=1, image_size=256,
                                                        radii=(0.22, 0.28, 0.25)).items()}
x_T, x_in, clip = [t.to(dev) for t in synthetic.make_latents(N, 32, seed=6033)]
x = x_T[:, lo:hi].contiguous()
noise = torch.randn(1, N, 4, 32, 32, device=dev)[:, lo:hi].contiguous()
index = 25
step = int(sampler.ddim_timesteps[index])
ts = torch.full((1,), step, device=dev, dtype=torch.long)

def one(xx):
    return sampler.denoise_apply(xx, {"x": x_in}, clip, ts, index, 2.0, batch_view_num=hi - lo, batch=batch, noise=noise,
                                 host_steps=[step])

with torch.no_grad():
    for _ in range(3):
        y = one(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = one(x)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"eager: {eager:.3f} ms/step", flush=True)
    g = torch.cuda.CUDAGraph()
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                yg = one(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            yg = one(x)
        torch.cuda.synchronize()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / args.steps * 1e3
        rel = ((yg - y).norm() / y.norm()).item()
        print(f"graph replay: {rep:.3f} ms/step (eager {eager:.3f}); replayed output vs eager relL2 = {rel:.2e}", flush=True)
    except Exception as e:  # noqa: BLE001
        print("capture failed:", repr(e)[:600], flush=True)
