"""Linear-layer microbenchmark, warm (back-to-back, operands cached) vs cold (caches evicted before every launch)."""
import sys
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
def run(M, K, N, **kw):
    w = e.bench_linear(M, K, N, iters=20, **kw)
    c = e.bench_linear(M, K, N, iters=10, cold=True, **kw)
    print(f"M={M:6d} K={K:5d} N={N:5d} {kw}: warm {w*1e3:6.1f} us, cold {c*1e3:6.1f} us", flush=True)
run(32768, 320, 320)
run(32768, 320, 320, resid=True)
run(32768, 320, 320, resid=True, bias=True, rowbias=True)
run(32768, 320, 320, out_half=True)
run(32768, 1280, 320, out_half=True, resid=True, bias=True)
run(32768, 320, 960, out_half=True)
run(32768, 320, 2560, geglu=True, out_half=True, bias=True)
run(8192, 640, 640, resid=True, bias=True, rowbias=True)
run(8192, 640, 640, out_half=True)
run(8192, 2560, 640, out_half=True, resid=True, bias=True)
run(8192, 640, 5120, geglu=True, out_half=True, bias=True)
run(2048, 1280, 1280, resid=True, bias=True, rowbias=True)
run(2048, 5120, 1280, out_half=True, resid=True, bias=True)
run(2048, 1280, 10240, geglu=True, out_half=True, bias=True)
