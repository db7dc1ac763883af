"""Round 6: the LDS-DMA GEMM's 128-row (four-wave) tiles against the 256-row ones on the transformer-block shapes whose 256-row tile
grid leaves CUs idle (one process per setting: the overrides are read once).
    python tools/gemm_bm_sweep.py          -> every setting in subprocesses; first line = the planner's own choice (no override)
    python tools/gemm_bm_sweep.py --one    -> one setting (MVD_DENSE_BM / MVD_DENSE_BN / MVD_DENSE_SK from the environment)
Times in microseconds per launch (20 back-to-back launches, fp32 output + fp32 residual: the to_out / proj_out epilogue)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(8192, 640, 640), (8192, 2560, 640), (8192, 640, 1920), (2048, 1280, 1280), (2048, 5120, 1280), (2048, 1280, 3840),
          (512, 1280, 1280), (512, 5120, 1280), (4096, 640, 640), (1024, 1280, 1280), (32768, 320, 320)]
if "--one" in sys.argv:
    from morphablediffusion_amd.engine import Engine
    from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
    e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
    out = []
    for (M, K, N) in SHAPES:
        ms = e.bench_linear(M, K, N, iters=20, resid=True)
        out.append(f"{ms*1e3:6.1f}")
    print(f"bm={os.environ.get('MVD_DENSE_BM','-'):>3} bn={os.environ.get('MVD_DENSE_BN','-'):>3} sk={os.environ.get('MVD_DENSE_SK','-'):>2} | "
          + " ".join(out), flush=True)
else:
    print("shapes (M,K,N): " + " ".join(f"{m}x{k}x{n}" for m, k, n in SHAPES), flush=True)
    settings = [("", "", ""), ("NO128", "", "")]
    for bm, bns in (("256", ("64", "128", "160")), ("128", ("96", "128", "160"))):
        for bn in bns:
            for sk in ("1", "2", "4"):
                settings.append((bm, bn, sk))
    for bm, bn, sk in settings:
        env = dict(os.environ)
        if bm == "NO128":
            env["MVD_NO_BM128"] = "1"
        elif bm:
            env.update(MVD_DENSE_BM=bm, MVD_DENSE_BN=bn, MVD_DENSE_SK=sk)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, stderr=subprocess.DEVNULL)
