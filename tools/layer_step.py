"""Per-GEMM timing of ONE denoising step (MVD_LAYER_TIMING=1 makes the engine time every GEMM launch with HIP events and
a host sync, i.e. isolated layer times).  Prints the aggregated table of the last step.
  MVD_LAYER_TIMING=1 python tools/layer_step.py [--simulate-gpus G] 2> layers.log ; python tools/layer_agg.py layers.log <n>"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import N_VIEWS, unet_kwargs
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--simulate-gpus", type=int, default=0)
args = ap.parse_args()
ucfg, vcfg = UNetConfig(), VolumeConfig(num_views=N_VIEWS)
W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)
model = SyncMultiviewDiffusion(unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": unet_kwargs(ucfg)},
                               view_num=N_VIEWS, image_size=256, cfg_scale=2.0, workspace_gb=32.0)
model.load_state_dict(W)
model.eval()
s = model.sampler
if args.simulate_gpus:
    s.simulate_world = args.simulate_gpus
lo, hi = s.view_range(N_VIEWS)
batch = {k: v.cuda() for k, v in synthetic.make_batch(N_VIEWS, "perspective", 5023, mesh_seed=1).items()}
x_T, x_in, clip = [t.cuda() for t in synthetic.make_latents(N_VIEWS, 32, seed=6033)]
x = x_T[:, lo:hi].contiguous()
noise = torch.randn(x.shape, device="cuda")
for i in range(3):
    sys.stderr.write("[step-begin]\n")
    sys.stderr.flush()
    ts = torch.full((1,), 481, device="cuda", dtype=torch.long)
    x = s.denoise_apply(x, {"x": x_in}, clip, ts, 24, 2.0, batch_view_num=hi - lo, batch=batch, noise=noise, host_steps=[481])
    torch.cuda.synchronize()
