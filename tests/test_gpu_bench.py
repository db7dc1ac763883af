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


def test_bench_line_carries_counter_traffic_when_the_pmc_summary_is_of_this_build():
    """roofline.traffic comes from profiles/pmc_traffic.json (rocprofv3 --pmc passes) and is reported only when that file is
    stamped with the hash of THIS tree's library sources: a committed summary of another build must read as null, a current
    one as a number."""
    sys.path.insert(0, ROOT)
    from morphablediffusion_amd.lib import csrc_sha16
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        pmc = json.load(f)
    p = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    r = _one_json_line(p.stdout)["roofline"]
    current = pmc.get("csrc_sha16") == csrc_sha16() and r["kernel"] in pmc["bytes_per_launch"]
    print(f"[property] pmc_traffic.json stamp {pmc.get('csrc_sha16')} vs sources {csrc_sha16()}: traffic = {r['traffic']}")
    assert (r["traffic"] is not None) == current
    if current:
        assert r["traffic"] > 0


def test_bench_train_at_config3_per_gpu_shape():
    """BASELINE configs[3] at the PER-GPU shape its stated scale implies (batch 140 over 8 GPUs = 18 samples per GPU, FaceScape's
    bilinear topology = 26317 vertices per mesh, a new mesh per sample, bf16): the complete training step -- conditioner and UNet
    forward, loss, both backward passes, AdamW, re-pack -- runs at full width with finite, non-increasing-by-much losses and no
    skipped optimiser step.  (Parity of that step against the reference is tests/test_gpu_train*.py at 2 - 4 samples.)"""
    env = dict(os.environ, MVD_DTYPE="bf16")
    p = subprocess.run([sys.executable, "bench.py", "--config", "train", "--train-batch", "18", "--train-vertices", "26317",
                        "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _one_json_line(p.stdout)
    assert d["unit"] == "samples/s" and d["dtype"] == "bf16" and d["n_gpus"] == 1 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 18 and d["config"]["mesh_vertices_after_voxel_dedup"] > 15000
    first, last = d["loss_first_last"]
    assert 0.5 < first < 2.0 and 0.5 < last < 2.0, d["loss_first_last"]   # MSE of a random-init UNet against unit noise
    assert d["optimizer_steps_skipped"] == 0
    assert abs(d["value"] * d["ms_per_step"] / 1e3 - 18.0) < 1e-6
    print(f"[property] config-3 per-GPU shape: {d['ms_per_step']:.1f} ms per step, {d['value']:.1f} samples/s, "
          f"losses {first:.4f} -> {last:.4f}")
