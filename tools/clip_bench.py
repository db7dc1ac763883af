"""Time the engine's CLIP image embedding (ViT-L/14, seeded weights) for B images of 256^2."""
import sys, time
import torch
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import ClipConfig, UNetConfig, VolumeConfig, clip_manifest
from morphablediffusion_amd.weights import seeded_state_dict

cfg = ClipConfig()
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=2.0)
e.load_state_dict(seeded_state_dict(clip_manifest(cfg), 0))
for B in (1, 4, 16):
    x = (torch.rand(B, 3, 256, 256) * 2 - 1).cuda()
    for _ in range(3):
        e.clip_encode(x)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 20
    for _ in range(n):
        e.clip_encode(x)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    gf = B * (2 * 256 * 588 * 1024 + 24 * (2 * 257 * 1024 * 1024 * 12 + 4 * 257 * 257 * 1024)) / 1e9
    print(f"clip_encode B={B}: {ms:.2f} ms  ({gf / ms:.1f} TFLOP/s)")
