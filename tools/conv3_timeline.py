"""Per-workgroup phase timeline of the halo 3x3 conv (investigation build: make EXTRA=-DMVD_TIMELINE BUILD=build_tl
LIB=../libmvd_hip_tl.so; MVD_LIB_PATH=.../libmvd_hip_tl.so python tools/conv3_timeline.py)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd import lib as L
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
lib = L.load()
NB = 4096
def tl():
    buf = (C.c_ulonglong * (NB * 8))()
    assert lib.mvd_debug_conv3_timeline(buf, NB * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 8).astype(np.int64)
    return t[t[:, 0] != 0]
for (B, Cin, S, Co) in ((32, 320, 32, 320), (32, 640, 32, 320), (32, 640, 16, 640), (32, 1280, 8, 1280), (32, 1280, 16, 640)):
    tl()
    e.bench_conv(B, Cin, S, S, Co, iters=1)
    t = tl()
    ms = e.bench_conv(B, Cin, S, S, Co, iters=10)
    r = (t - t[:, 0].min()) * 0.01
    med = lambda x: float(np.median(x))
    fl = 2.0 * B * S * S * Co * Cin * 9
    nsteps = "?"
    print(f"B={B} Cin={Cin} {S}x{S} Cout={Co}: {ms*1e3:.1f} us back-to-back ({fl/ms/1e9:.0f} TF), {len(t)} workgroups | start spread {r[:,0].max():.2f} | "
          f"setup {med(r[:,6]-r[:,0]):.2f} | to first data {med(r[:,1]-r[:,0]):.2f} | step0 {med(r[:,2]-r[:,1]):.2f} step1 {med(r[:,3]-r[:,2]):.2f} | "
          f"main loop {med(r[:,4]-r[:,1]):.2f} | epilogue {med(r[:,5]-r[:,4]):.2f} | block total {med(r[:,5]-r[:,0]):.2f} | kernel span {r[:,5].max():.2f}")
