"""Halo-conv tile/split-K sweep: python tools/conv_bench3.py  (env MVD_HALO_BN / MVD_HALO_SK are read per process)."""
import os, subprocess, sys
SHAPES = [(32, 640, 16, 640), (32, 1280, 8, 1280), (32, 1280, 16, 640), (32, 1920, 16, 640), (32, 2560, 8, 1280), (32, 960, 16, 640), (32, 320, 32, 320), (32, 640, 32, 320), (4, 320, 32, 320), (4, 640, 16, 640), (4, 1280, 8, 1280)]
if len(sys.argv) > 1:
    sys.path.insert(0, "/root/repo")
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
    out = []
    for (B, C, S, Co) in SHAPES:
        ms = e.bench_conv(B, C, S, S, Co, iters=10)
        out.append(f"{ms*1e3:6.1f}")
    print(" ".join(out))
else:
    print("shapes:", SHAPES)
    for bn in (128, 160):
        for sk in (1, 2, 3, 4, 5, 8):
            env = dict(os.environ, MVD_HALO_BN=str(bn), MVD_HALO_SK=str(sk))
            r = subprocess.run([sys.executable, __file__, "x"], env=env, capture_output=True, text=True)
            print(f"bn={bn} sk={sk}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}")
