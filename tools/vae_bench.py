"""First-stage decoder timing: 16 views batched (the tail of `SyncMultiviewDiffusion.sample`)."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VaeConfig, VolumeConfig, vae_decoder_manifest
from morphablediffusion_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=24.0)
e.load_state_dict(seeded_state_dict(vae_decoder_manifest(VaeConfig()), 7))
z = torch.randn(B, 4, 32, 32, device="cuda") * 4
e.vae_decode(z); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): e.vae_decode(z)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"vae decode B={B}: {dt*1e3:.2f} ms  {622e9*B/dt/1e12:.0f} TFLOP/s")
