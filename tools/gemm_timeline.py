"""Per-workgroup phase timeline of the LDS-DMA GEMM (needs the investigation build:
make -C morphablediffusion_amd/csrc BUILD=build_tl LIB=../libmvd_hip_tl.so EXTRA=-DMVD_TIMELINE, then
MVD_LIB_PATH=$PWD/morphablediffusion_amd/libmvd_hip_tl.so python tools/gemm_timeline.py).

Timestamps (100 MHz wall clock, 10 ns ticks) per workgroup: 0 start, 1 first k-step landed, 2 / 3 end of k-steps 0 / 1,
4 main loop of the (last) tile done, 5 end of the kernel."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from morphablediffusion_amd import lib as L  # noqa: E402
from morphablediffusion_amd.engine import Engine  # noqa: E402
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig  # noqa: E402

e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
lib = L.load()
NB = 4096


def timeline():
    buf = (C.c_ulonglong * (NB * 16))()
    assert lib.mvd_debug_timeline(buf, NB * 16) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 16).astype(np.int64)
    return t[t[:, 0] != 0]


def run(M, K, N, **kw):
    timeline()  # clear
    ms1 = e.bench_linear(M, K, N, iters=1, **kw)
    t = timeline()
    ms = e.bench_linear(M, K, N, iters=20, **kw)
    timeline()
    fl = 2.0 * M * K * N
    t0 = t[:, 0].min()
    r = (t - t0) * 0.01  # us
    med = lambda x: float(np.median(x))
    print(f"M={M:6d} K={K:5d} N={N:5d} {kw}: {ms*1e3:6.1f} us back-to-back ({fl/ms/1e9:6.1f} TF), single {ms1*1e3:6.1f} us, "
          f"{len(t)} workgroups")
    print(f"   start spread {r[:,0].max():5.2f} | setup {med(r[:,6]-r[:,0]):5.2f} issue0 {med(r[:,7]-r[:,6]):5.2f} | to first data {med(r[:,1]-r[:,0]):5.2f} | k-step0 {med(r[:,2]-r[:,1]):5.2f} "
          f"k-step1 {med(r[:,3]-r[:,2]):5.2f} | main loop {med(r[:,4]-r[:,1]):5.2f} | epilogue {med(r[:,5]-r[:,4]):5.2f} "
          f"\n   step0: kk0 {med(r[:,8]-r[:,1]):5.2f} kk1+dma {med(r[:,9]-r[:,8]):5.2f} kk2-3 {med(r[:,10]-r[:,9]):5.2f} vmwait {med(r[:,11]-r[:,10]):5.2f} barrier {med(r[:,2]-r[:,11]):5.2f}"
          f" | step1: kk0 {med(r[:,12]-r[:,2]):5.2f} kk1+dma {med(r[:,13]-r[:,12]):5.2f} kk2-3 {med(r[:,14]-r[:,13]):5.2f} vmwait {med(r[:,15]-r[:,14]):5.2f} barrier {med(r[:,3]-r[:,15]):5.2f}\n  "
          f"| block total med {med(r[:,5]-r[:,0]):5.2f} max {float((r[:,5]-r[:,0]).max()):5.2f} | kernel span {r[:,5].max():5.2f}")


run(32768, 320, 320)
run(32768, 320, 320, resid=True)
run(32768, 320, 320, out_half=True)
run(32768, 960, 320, resid=True)
run(32768, 1280, 320, out_half=True, resid=True)
run(32768, 320, 960, out_half=True)
run(32768, 320, 2560, geglu=True, out_half=True)
run(8192, 640, 640)
run(8192, 640, 640, resid=True)
run(8192, 640, 640, out_half=True)
run(8192, 2560, 640, out_half=True)
run(8192, 640, 5120, geglu=True, out_half=True)
run(8192, 640, 1920, out_half=True)
run(2048, 1280, 1280, resid=True)
run(2048, 1280, 1280, out_half=True)
run(2048, 5120, 1280, out_half=True)
run(2048, 1280, 10240, geglu=True, out_half=True)
run(2048, 1280, 3840, out_half=True)
run(98304, 128, 128, out_half=True)
