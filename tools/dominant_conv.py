"""Runs only the dominant kernel (level-32 3x3 conv, 16 views x CFG) a few times: target of the rocprofv3 --pmc passes."""
import sys
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
ms = e.bench_conv(32, 320, 32, 32, 320, iters=20)
print(f"conv 320->320 @32x32 B=32: {ms*1e3:.1f} us, {2.0*32*1024*320*2880/ms/1e9:.0f} TF")
