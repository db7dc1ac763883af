"""GroupNorm(32)+SiLU kernel timing at the UNet's shapes (HIP events around back-to-back launches, no layout conversion)."""
import sys
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
out = []
for (B, C, hw) in ((32, 320, 32), (32, 640, 32), (32, 960, 32), (32, 640, 16), (32, 1280, 16), (32, 1920, 16), (32, 1280, 8), (32, 2560, 8),
                   (32, 1920, 8), (32, 1280, 4), (32, 2560, 4), (4, 320, 32), (4, 640, 16), (4, 1280, 8)):
    ms = e.bench_group_norm(B, C, hw * hw)
    mb = B * C * hw * hw * 6 / 1e6
    out.append(f"B={B} C={C} {hw}x{hw}: {ms*1e3:6.1f} us {mb/ms/1e3:5.2f} TB/s")
print("\n".join(out))
