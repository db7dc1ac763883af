import sys, os
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
for B in (8, 16, 32, 48, 64):
    ms = e.bench_conv(B, 320, 32, 32, 320, iters=10)
    fl = 2.0 * B * 1024 * 320 * 9 * 320
    print(f"B={B} WGs={B*8*2}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF")
for B in (16, 32, 64):
    ms = e.bench_conv(B, 640, 16, 16, 640, iters=10)
    fl = 2.0 * B * 256 * 640 * 9 * 640
    print(f"640ch B={B} WGs={B*2*5}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF")
