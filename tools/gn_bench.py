"""GroupNorm(32)+SiLU kernel timing at the UNet's shapes (run under rocprofv3 --kernel-trace --stats for kernel times)."""
import sys, time
import torch
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
for (B, C, hw) in ((32, 320, 32), (32, 640, 32), (32, 960, 32), (32, 640, 16), (32, 1280, 16), (32, 1920, 16), (32, 1280, 8), (32, 2560, 8),
                   (4, 320, 32), (4, 640, 16), (4, 1280, 8)):
    x = torch.randn(B, C, hw, hw, device="cuda")
    g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    e.op_group_norm(x, 32, g, b, 1e-5, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): e.op_group_norm(x, 32, g, b, 1e-5, 1)
    torch.cuda.synchronize()
    print(f"B={B} C={C} {hw}x{hw}: {(time.perf_counter()-t0)/10*1e6:.0f} us per op call (incl. layout conversion), {B*C*hw*hw*6/1e6:.0f} MB")
