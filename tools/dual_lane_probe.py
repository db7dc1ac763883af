"""Round 6 experiment: do two independent half-batch chains on two HIP streams of ONE process overlap on the GPU?
Two engines (own weights, own workspace), each running the 8-views-per-pass step of the N = 16 headline configuration on its own
stream, enqueued back to back by one host thread -- against one engine running all 16 views in one pass.  (Round 4 tried two
PROCESSES: the GPU time-slices processes, no gain.)   python tools/dual_lane_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N, dev = 16, "cuda:0"
ucfg, vcfg = UNetConfig(image_size=32), VolumeConfig(num_views=N)
W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)


def make(ws_gb):
    m = SyncMultiviewDiffusion(unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": bench.unet_kwargs(ucfg)},
                               view_num=N, image_size=256, cfg_scale=2.0, device=dev, workspace_gb=ws_gb)
    m.load_state_dict(W)
    m.eval()
    return m


batch = {k: v.to(dev) for k, v in synthetic.make_batch(N, "perspective", 5023, mesh_seed=1, image_size=256, radii=(0.22, 0.28, 0.25)).items()}
x_T, x_in, clip = [t.to(dev) for t in synthetic.make_latents(N, 32, seed=6033)]
info = {"x": x_in}
noise = torch.randn(1, N, 4, 32, 32, device=dev, generator=torch.Generator(device=dev).manual_seed(123))


def stepper(m, lo, hi, bvn):
    s = m.sampler
    nsteps = len(s.ddim_timesteps)
    x = x_T[:, lo:hi].contiguous()
    nz = noise[:, lo:hi].contiguous()
    if hi - lo < N:
        s.simulate_world = N // (hi - lo)

    def one(i):
        nonlocal x
        index = nsteps - 1 - (i % nsteps)
        step = int(s.ddim_timesteps[index])
        x = s.denoise_apply(x, info, clip, s._time_steps(1, step, dev), index, 2.0, batch_view_num=bvn, is_step0=index == 0,
                            batch=batch, noise=nz, host_steps=[step])
    return one


def timed(fns, streams, steps=STEPS, warm=3):
    with torch.no_grad():
        for i in range(warm):
            for f, st in zip(fns, streams):
                with torch.cuda.stream(st):
                    f(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            for f, st in zip(fns, streams):
                with torch.cuda.stream(st):
                    f(warm + i)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps, 1e3 * t_host / steps


mA = make(40.0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
full, hfull = timed([stepper(mA, 0, 16, 16)], [sA])
print(f"one engine, 16 views in one pass            : {full:7.3f} ms per step (host enqueue {hfull:.2f})", flush=True)
seq, hseq = timed([stepper(mA, 0, 16, 8)], [sA])
print(f"one engine, 16 views as 2 passes of 8       : {seq:7.3f} ms per step (host enqueue {hseq:.2f})", flush=True)
half, hhalf = timed([stepper(mA, 0, 8, 8)], [sA])
print(f"one engine, 8 of 16 views (half a step)     : {half:7.3f} ms per half step (host enqueue {hhalf:.2f})", flush=True)
mB = make(40.0)
both, hboth = timed([stepper(mA, 0, 8, 8), stepper(mB, 8, 16, 8)], [sA, sB])
print(f"two engines x 8 views on two streams        : {both:7.3f} ms per full step (host enqueue {hboth:.2f})", flush=True)
both1, hboth1 = timed([stepper(mA, 0, 8, 8), stepper(mB, 8, 16, 8)], [sA, sA])
print(f"two engines x 8 views on ONE stream (control): {both1:7.3f} ms per full step (host enqueue {hboth1:.2f})", flush=True)

# ---- the same with the host taken out: each engine's half step captured as a hipGraph, replayed on one / two streams
def capture(m, lo, hi):
    f = stepper(m, lo, hi, 8)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.no_grad():
        with torch.cuda.stream(s):
            f(25)
            f(25)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            f(25)
    torch.cuda.synchronize()
    return g


def timed_graphs(gs, streams, steps=STEPS):
    for _ in range(3):
        for g, st in zip(gs, streams):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, st in zip(gs, streams):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


try:
    gA, gB = capture(mA, 0, 8), capture(mB, 8, 16)
    print(f"graph: one half step alone                  : {timed_graphs([gA], [sA]):7.3f} ms per half step", flush=True)
    print(f"graph: two half steps on ONE stream         : {timed_graphs([gA, gB], [sA, sA]):7.3f} ms per full step", flush=True)
    print(f"graph: two half steps on TWO streams        : {timed_graphs([gA, gB], [sA, sB]):7.3f} ms per full step", flush=True)
    mF = stepper(mA, 0, 16, 16)
    gF = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.no_grad():
        with torch.cuda.stream(s):
            mF(25)
            mF(25)
        torch.cuda.synchronize()
        with torch.cuda.graph(gF):
            mF(25)
    print(f"graph: the full 16-view step                : {timed_graphs([gF], [sA]):7.3f} ms per full step", flush=True)
except Exception as e:  # noqa: BLE001
    print("graph part failed:", repr(e)[:800], flush=True)
