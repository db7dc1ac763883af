"""GPU: the training step in bfloat16 -- BASELINE.json configs[3] ("finetune_unet training step bf16"; reference step
morphable_diffusion.py:520-549, optimiser :627-646).  libmvd_hip_bf16.so is the SAME source tree built with bfloat16 MFMA operands
and storage (csrc/common.h MVD_BF16), fp32 accumulation, fp32 master weights / moments, and NO loss scaling (bf16 has fp32's
exponent range); MVD_DTYPE=bf16 selects it per process, hence the subprocess.

Bounds.  bf16 carries an 8-bit significand against fp16's 11: every operand rounding is 8x coarser (2^-9 = 2.0e-3 against 2.4e-4),
and the measured errors are 8x the fp16 ones throughout (profiles/r04_b_train_dtype_check.txt: prediction 5.5e-3 / 7.0e-4, trunk
gradients worst 3.7e-2 / 4.6e-3, median 1.7e-2 / 2.4e-3, DepthTransformer gradients worst 1.2e-1 / 3.9e-2).  Asserted: loss 2e-3,
prediction 1.2e-2, trunk gradients worst 8e-2 (= 8 x the fp16 bound) median 4e-2, DepthTransformer tensors (three ReLU masks and a
near-uniform softmax re-derived from a rounded input: tests/test_host_cpu.py::test_depth_transformer_gradient_sensitivity) worst
0.3 median 0.1, the direction of the WHOLE gradient (cosine >= 0.9995 to the reference's; measured 0.99997), bit-reproducibility,
no skipped optimiser step at loss scale 1, and that six AdamW steps descend like the fp16 run.  The fp16 library runs the same
script as the control."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(dtype):
    env = dict(os.environ, MVD_DTYPE=dtype)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_dtype_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_training_step_bf16_vs_reference_and_fp16_control():
    if not os.path.exists(os.path.join(ROOT, "morphablediffusion_amd", "libmvd_hip_bf16.so")):
        pytest.fail("libmvd_hip_bf16.so is missing: __graft_entry__.build() makes it (make -C morphablediffusion_amd/csrc bf16)")
    b = _run("bf16")
    f = _run("f16")
    for d in (b, f):
        print(f"[parity] training step in {d['dtype']}: loss rel {d['loss_rel_err']:.2e}, prediction {d['pred_rel_l2']:.2e}, trunk gradients "
              f"worst {d['grad_trunk_worst']:.2e} median {d['grad_trunk_median']:.2e}, DepthTransformer gradients worst "
              f"{d['grad_dt_worst']:.2e} median {d['grad_dt_median']:.2e}, cosine {d['grad_cosine']:.6f}, loss scale {d['loss_scale']:g}, "
              f"6 steps: {d['losses'][0]:.4f} -> {d['losses'][-1]:.4f}")
    assert b["dtype"] == "bf16" and f["dtype"] == "f16" and b["loss_scale"] == 1.0
    assert b["finite"] and b["bit_reproducible"] and b["steps_skipped"] == 0
    assert b["loss_rel_err"] <= 2e-3 and b["pred_rel_l2"] <= 1.2e-2
    # round 6: 1.5 x what this step measures (profiles/r05_z_pytest_full.log: trunk worst 3.59e-2 median 1.73e-2, DepthTransformer
    # worst 1.16e-1 median 3.38e-2); until round 5 the bounds were 8e-2 / 4e-2 and 0.3 / 0.1
    assert b["grad_trunk_worst"] <= 5.5e-2 and b["grad_trunk_median"] <= 2.6e-2
    assert b["grad_dt_worst"] <= 0.18 and b["grad_dt_median"] <= 5.1e-2
    assert b["grad_cosine"] >= 0.9995 and f["grad_cosine"] >= 0.9999
    assert b["losses"][-1] < 0.9 * b["losses"][0], b["losses"]
    # the control keeps the fp16 bounds of tests/test_gpu_train.py
    assert f["loss_rel_err"] <= 1e-3 and f["pred_rel_l2"] <= 2e-3 and f["grad_trunk_worst"] <= 7e-3
    # both dtypes descend alike on the same batch (same optimiser, fp32 masters)
    assert abs(b["losses"][-1] - f["losses"][-1]) <= 0.1 * f["losses"][0]


def test_full_width_training_gradients_bf16_vs_reference_golden():
    """The full-width step (916.9 M parameters, B = 2) in bfloat16 against tests/golden/train_full.npz -- the reference's own
    training_step + loss.backward() -- in a process of its own (tools/train_full_check.py).  Bounds = 8 x the fp16 ones of
    tests/test_gpu_train.py::test_training_step_full_width_gradients_vs_reference would allow 8e-2 / 0.4 (8- against 11-bit
    significands); asserted is what the full-width step measures with headroom -- trunk tensors 4e-2 (measured worst 2.4e-2,
    median 9.8e-3), DepthTransformer tensors 8e-2 (measured worst 4.9e-2, median 3.1e-2), loss 1e-3 (5.3e-5): at full width the
    wider reductions average the operand rounding further than at the reduced width of the test above."""
    if not os.path.exists(os.path.join(ROOT, "morphablediffusion_amd", "libmvd_hip_bf16.so")):
        pytest.fail("libmvd_hip_bf16.so is missing")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "train_full_check.py")], cwd=ROOT, env=dict(os.environ, MVD_DTYPE="bf16"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print(f"[parity] full-width training step in bf16: loss rel {d['loss_rel_err']:.2e}, trunk gradients worst {d['grad_trunk_worst']:.2e} "
          f"({d['worst_trunk']}) median {d['grad_trunk_median']:.2e}, DepthTransformer worst {d['grad_dt_worst']:.2e} ({d['worst_dt']}) "
          f"median {d['grad_dt_median']:.2e}")
    assert d["dtype"] == "bf16" and d["n_cond"] == 6 and d["n_rest"] == 20
    assert d["loss_rel_err"] <= 1e-3
    assert d["grad_trunk_worst"] <= 4e-2 and d["grad_dt_worst"] <= 8e-2
