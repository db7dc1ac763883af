import sys, torch
sys.path.insert(0, ".")
from tests import golden_inputs as gi
from tests.test_gpu_model import make_model, to_dev
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.spec import VolumeConfig
import os
N = int(os.environ.get("DET_N", "4")); BVN = int(os.environ.get("DET_BVN", str(N)))
UCFG = gi.FULL_UNET if os.environ.get("DET_FULL") else gi.SMALL_UNET
m = make_model(UCFG, VolumeConfig(num_views=N), N, workspace_gb=float(os.environ.get("DET_WS", "6")))
batch = to_dev(synthetic.make_batch(N, "perspective", 600, mesh_seed=1))
x_T, x_in, clip = [t.cuda() for t in synthetic.make_latents(N, 32, seed=6033)]
noise = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(1)).cuda()
def step():
    ts = torch.full((1,), int(m.sampler.ddim_timesteps[20]), dtype=torch.long, device="cuda")
    return m.sampler.denoise_apply(x_T, {"x": x_in}, clip, ts, 20, 2.0, batch_view_num=BVN, batch=batch, noise=noise)
import os
ctxm = torch.cuda.stream(torch.cuda.Stream()) if os.environ.get("DET_STREAM") else __import__("contextlib").nullcontext()
ctxm.__enter__()
ref = step()
bad = 0
REPS = int(os.environ.get("DET_REPS", "100"))
for i in range(REPS):
    o = step()
    if not torch.equal(o, ref):
        bad += 1
        print(i, "diff", (o - ref).abs().max().item(), "of", ref.abs().max().item())
print("mismatches", bad, "of", REPS, "| env:", {k: v for k, v in os.environ.items() if k.startswith(("MVD_", "DET_"))})
