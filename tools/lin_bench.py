"""Dense GEMM microbenchmark: time vs K for the layer shapes of the denoiser."""
import sys
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
def run(M, K, N, **kw):
    ms = e.bench_linear(M, K, N, iters=20, **kw)
    fl = 2.0 * M * K * N
    print(f"M={M:6d} K={K:5d} N={N:5d} {kw}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TF")
for K in (64, 128, 320, 640, 1280, 2560):
    run(32768, K, 320)
for K in (64, 320, 640, 1280, 2560):
    run(8192, K, 640)
run(32768, 320, 320, resid=True)
run(32768, 320, 320, out_half=True)
run(8192, 640, 640, resid=True)
run(8192, 640, 640, out_half=True)
run(2048, 1280, 1280, resid=True)
run(32768, 320, 2560, geglu=True, out_half=True)
run(32768, 320, 2560, out_half=True)
run(8192, 640, 5120, geglu=True, out_half=True)
run(8192, 640, 5120, out_half=True)
run(32768, 320, 640, out_half=True)
