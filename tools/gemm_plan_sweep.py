"""Sweep of the LDS-DMA GEMM's column-tile width and split-K for the small-M layer shapes (one process per setting: the
overrides are read once).  python tools/gemm_plan_sweep.py            -> runs every setting in subprocesses
                            python tools/gemm_plan_sweep.py --one      -> one setting (MVD_DENSE_BN / MVD_DENSE_SK from the env)"""
import os, subprocess, sys
sys.path.insert(0, "/root/repo")
SHAPES = [(2048, 1280, 1280), (2048, 3840, 1280), (2048, 2560, 1280), (2048, 5120, 1280), (2048, 1280, 3840), (8192, 640, 640),
          (8192, 1920, 640), (8192, 2560, 640), (512, 11520, 1280), (512, 1280, 1280), (512, 5120, 1280), (32768, 320, 320),
          (32768, 1280, 320)]
if "--one" in sys.argv:
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
    out = []
    for (M, K, N) in SHAPES:
        ms = e.bench_linear(M, K, N, iters=20, resid=True)
        out.append(f"{ms*1e3:6.1f}")
    print(f"bn={os.environ.get('MVD_DENSE_BN','-'):>3} sk={os.environ.get('MVD_DENSE_SK','-'):>2} | " + " ".join(out), flush=True)
else:
    print("shapes (M,K,N): " + " ".join(f"{m}x{k}x{n}" for m, k, n in SHAPES), flush=True)
    for bn in ("", "64", "128", "160"):
        for sk in ("", "1", "2", "3", "4", "6", "8", "12"):
            if bn == "" and sk != "":
                continue
            env = dict(os.environ)
            if bn: env["MVD_DENSE_BN"] = bn
            if sk: env["MVD_DENSE_SK"] = sk
            subprocess.run([sys.executable, __file__, "--one"], env=env, stderr=subprocess.DEVNULL)
