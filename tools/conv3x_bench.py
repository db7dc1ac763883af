"""conv3x (csrc/k_conv3x.hip) against the halo kernel it replaces (csrc/k_conv3.hip) on the UNet's ResBlock shapes at the headline
batch (32 UNet samples).  python tools/conv3x_bench.py        (MVD_NO_CONV3X=1 selects the old kernel)"""
import os, sys
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig

e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
tag = "halo " if os.environ.get("MVD_NO_CONV3X") else "conv3x"
for (B, C, H, Cout) in [(32, 320, 32, 320), (32, 640, 32, 320), (32, 960, 32, 320), (32, 640, 16, 640), (32, 1280, 16, 640),
                        (32, 320, 16, 640)] + ([(4, 320, 32, 320), (4, 640, 16, 640)] if not os.environ.get("CX_SHORT") else []):
    ms = e.bench_conv(B, C, H, H, Cout, iters=20)
    fl = 2.0 * B * H * H * Cout * C * 9
    print(f"{tag} B={B} {C:5d}->{Cout:4d} {H}x{H}: {ms * 1e3:8.1f} us  {fl / ms * 1e-9:6.0f} TFLOP/s ({fl / ms * 1e-9 / 2500:.3f})", flush=True)
e.close()
