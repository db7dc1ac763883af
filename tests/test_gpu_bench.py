"""GPU: the bench.py contract, as the driver invokes it.  One JSON line on stdout with the fields the driver reads;
N > 1 through `python -m torch.distributed.run` exactly as the driver launches it, on this one-GPU box with the
test knobs MVD_DIST_BACKEND=gloo / MVD_FORCE_DEVICE=0 (RCCL needs one GPU per rank; everything else is the same code)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines       # exactly ONE line on stdout
    return json.loads(lines[0])


def _check(d, n_gpus, steps, warmup):
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-6
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1


def test_bench_single_gpu_line():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _one_json_line(p.stdout)
    _check(d, 1, 2, 1)
    assert d["config"]["views_per_gpu"] == 16


def test_bench_two_ranks_as_the_driver_launches_it():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MVD_DIST_BACKEND="gloo", MVD_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _one_json_line(p.stdout)        # rank 0 only
    _check(d, 2, 2, 1)
    assert d["config"]["views_per_gpu"] == 8 and d["scaling"] == "strong"
