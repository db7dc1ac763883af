"""Per-workgroup phase timeline of the register-staged implicit GEMM on fp32-source Linear layers (investigation build:
make EXTRA=-DMVD_TIMELINE BUILD=build_tl LIB=../libmvd_hip_tl.so)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd import lib as L
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
lib = L.load()
NB = 8192
def tl():
    buf = (C.c_ulonglong * (NB * 8))()
    assert lib.mvd_debug_igemm_timeline(buf, NB * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 8).astype(np.int64)
    return t[t[:, 0] != 0]
for (M, K, N) in ((32768, 960, 320), (32768, 640, 320), (8192, 1920, 640), (2048, 2560, 1280), (32768, 320, 128), (8192, 640, 256)):
    tl()
    e.bench_linear(M, K, N, iters=1, a_f32=True, bias=True)
    t = tl()
    ms = e.bench_linear(M, K, N, iters=20, a_f32=True, bias=True)
    tl()
    r = (t - t[:, 0].min()) * 0.01
    med = lambda x: float(np.median(x))
    st = np.sort(r[:, 0])
    print(f"M={M} K={K} N={N}: {ms*1e3:.1f} us back-to-back ({2.0*M*K*N/ms/1e9:.0f} TF, {(M*K*4+M*N*4)/ms/1e9:.2f} TB/s), {len(t)} workgroups | "
          f"late starters (> 2 us) {(st > 2.0).sum()} | setup {med(r[:,6]-r[:,0]):.2f} | first tile staged {med(r[:,1]-r[:,0]):.2f} | step0 {med(r[:,2]-r[:,1]):.2f} "
          f"step1 {med(r[:,3]-r[:,2]):.2f} | main loop {med(r[:,4]-r[:,1]):.2f} | epilogue {med(r[:,5]-r[:,4]):.2f} | block total {med(r[:,5]-r[:,0]):.2f} | "
          f"kernel span {r[:,5].max():.2f}")
