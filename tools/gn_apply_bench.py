"""GroupNorm as it is (one launch per (sample, group) slice, statistics + apply) against the APPLY pass alone (statistics given):
what the norm would cost if its statistics came out of the producer's epilogue.  python tools/gn_apply_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
print(f"{'B':>3s} {'C':>5s} {'HW':>5s} {'G':>3s} {'one launch us':>14s} {'apply only us':>14s} {'MB moved':>9s} {'apply GB/s':>10s}")
for (B, C, HW, G) in [(32, 320, 1024, 32), (32, 640, 1024, 32), (32, 960, 1024, 32), (32, 640, 256, 32), (32, 1280, 256, 32), (32, 1920, 256, 32),
                      (32, 1280, 64, 32), (32, 2560, 64, 32), (32, 1280, 16, 32), (32, 128, 1024, 8), (32, 256, 256, 8), (4, 320, 1024, 32), (4, 640, 256, 32),
                      (4, 1280, 64, 32)]:
    a = e.bench_group_norm(B, C, HW, G, iters=30) * 1e3
    b = e.bench_group_norm(B, C, HW, G, iters=30, apply_only=True) * 1e3
    mb = B * HW * C * 6 / 1e6
    print(f"{B:3d} {C:5d} {HW:5d} {G:3d} {a:14.1f} {b:14.1f} {mb:9.1f} {mb / b * 1e3:10.0f}")
