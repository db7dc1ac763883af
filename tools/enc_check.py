"""Fused 2-D encoder (k_enc.hip) vs the layer-by-layer path: per-view vertex features of one seeded sample.
  python tools/enc_check.py out_a.pt ; MVD_NO_FUSED_ENC=1 python tools/enc_check.py out_b.pt ; python tools/enc_check.py out_a.pt out_b.pt"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 3:
    a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
    print(f"fused vs layered: relL2 = {((a - b).norm() / b.norm()).item():.3e}, max abs = {(a - b).abs().max().item():.3e}, "
          f"ref max abs = {b.abs().max().item():.3e}")
    sys.exit(0)
from bench import unet_kwargs
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.model import SyncMultiviewDiffusion
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig, full_manifest
from morphablediffusion_amd.weights import seeded_state_dict
dev, N = "cuda:0", 16
ucfg, vcfg = UNetConfig(model_channels=64, image_size=32), VolumeConfig(num_views=N)
W = seeded_state_dict(full_manifest(ucfg, vcfg), 7)
kw = unet_kwargs(ucfg)
model = SyncMultiviewDiffusion(unet_config={"target": "ldm.models.diffusion.attention.DepthWiseAttention", "params": kw},
                               view_num=N, image_size=256, cfg_scale=2.0, device=dev, workspace_gb=4.0)
model.load_state_dict(W)
model.eval()
batch = {k: v.to(dev) for k, v in synthetic.make_batch(N, "perspective", 5023, mesh_seed=1).items()}
x_T, x_in, clip = [t.to(dev) for t in synthetic.make_latents(N, 32, seed=6033)]
ts = torch.full((1,), 500, device=dev, dtype=torch.long)
te = model.embed_time(ts)
ve = model.get_viewpoint_embedding(batch).to(dev)
model.spatial_volume._set_sample(batch, 0)
idx = torch.arange(N, dtype=torch.int32, device=dev)
eng = model.engine
vf = eng.vertex_view_features(x_T[0], te[0], ve[0], idx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    vf = eng.vertex_view_features(x_T[0], te[0], ve[0], idx)
torch.cuda.synchronize()
print(f"vertex_view_features: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call, finite = {bool(torch.isfinite(vf).all())}")
torch.save(vf.cpu(), sys.argv[1])
