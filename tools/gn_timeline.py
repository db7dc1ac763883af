"""Per-workgroup phase timeline of gn_group_kernel (investigation build: make EXTRA=-DMVD_TIMELINE BUILD=build_tl LIB=../libmvd_hip_tl.so)."""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd import lib as L
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
lib = L.load()
NB = 4096
def tl():
    buf = (C.c_ulonglong * (NB * 8))()
    assert lib.mvd_debug_gn_timeline(buf, NB * 8) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 8).astype(np.int64)
    return t[t[:, 0] != 0]
for (B, Cc, hw) in ((32, 320, 32), (32, 640, 32), (32, 640, 16), (32, 1280, 16), (32, 1280, 8)):
    tl()
    ms1 = e.bench_group_norm(B, Cc, hw * hw, iters=1)
    t = tl()
    ms = e.bench_group_norm(B, Cc, hw * hw, iters=20)
    r = (t - t[:, 0].min()) * 0.01
    med = lambda x: float(np.median(x))
    print(f"B={B} C={Cc} {hw}x{hw}: {ms*1e3:.1f} us back-to-back, {len(t)} workgroups | start spread {r[:,0].max():.2f} (median start {med(r[:,0]):.2f}) | "
          f"loads+sum {med(r[:,1]-r[:,0]):.2f} | reduce1 {med(r[:,2]-r[:,1]):.2f} | var+reduce2 {med(r[:,3]-r[:,2]):.2f} | normalise+store {med(r[:,4]-r[:,3]):.2f} | "
          f"block total {med(r[:,4]-r[:,0]):.2f} | kernel span {r[:,4].max():.2f}")
    st = np.sort(r[:, 0])
    print("   start-time percentiles (us): " + " ".join(f"p{p}={st[int(len(st) * p / 100) - (p == 100)]:.2f}" for p in (10, 25, 50, 60, 70, 75, 80, 90, 100))
          + f" | late starters (> 2 us): {(st > 2.0).sum()} of {len(st)}")
