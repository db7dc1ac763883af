"""One full-width denoising step (2 views, CFG 2.0, seeded weights and inputs) saved to a file: the optimised code paths of
the engine are compared with their general forms by running this under different environment switches and
comparing the outputs (tests/test_gpu_variants.py).   python tools/variant_step.py out.pt"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import unet_kwargs
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict

dev, N = "cuda:0", 2
ucfg, vcfg = UNetConfig(image_size=32), VolumeConfig(num_views=N, projection="perspective", input_image_size=256)
W = seeded_state_dict(full_manifest(ucfg, vcfg), 11)
model = SyncMultiviewDiffusion(unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": unet_kwargs(ucfg)},
                               projection="perspective", view_num=N, image_size=256, cfg_scale=2.0, device=dev, workspace_gb=16.0)
model.load_state_dict(W)
model.eval()
sampler = model.sampler
batch = {k: v.to(dev) for k, v in synthetic.make_batch(N, "perspective", 5023, mesh_seed=3, image_size=256,
                                                        radii=(0.22, 0.28, 0.25)).items()}
x_T, x_in, clip = [t.to(dev) for t in synthetic.make_latents(N, 32, seed=77)]
g = torch.Generator(device=dev).manual_seed(5)
noise = torch.randn(1, N, 4, 32, 32, device=dev, generator=g)
index = 20
step = int(sampler.ddim_timesteps[index])
ts = torch.full((1,), step, device=dev, dtype=torch.long)
with torch.no_grad():
    x_prev, eps = sampler.denoise_apply(x_T, {"x": x_in}, clip, ts, index, 2.0, batch_view_num=N, batch=batch, noise=noise,
                                        host_steps=[step], return_eps=True)
torch.cuda.synchronize()
assert torch.isfinite(eps).all()
torch.save({"x_prev": x_prev.cpu(), "eps": eps.cpu()}, sys.argv[1])
print("saved", sys.argv[1], float(eps.norm()))
