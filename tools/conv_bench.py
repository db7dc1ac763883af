"""Times the dominant implicit-GEMM shapes in isolation (HIP events on the launch stream)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
shapes = [(32, 320, 32, 32, 320), (32, 640, 16, 16, 640), (32, 1280, 8, 8, 1280), (32, 1280, 16, 16, 640), (4, 320, 32, 32, 320), (4, 1280, 8, 8, 1280)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
for B, C, H, W, Co in shapes:
    ms = e.bench_conv(B, C, H, W, Co, iters=10)
    fl = 2.0 * B * H * W * Co * 9 * C
    print(f"conv B={B} {C}->{Co} @{H}x{W}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
