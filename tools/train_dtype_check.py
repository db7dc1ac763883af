"""One training step of the reduced-width model against the reference's own training_step + loss.backward() golden
(tests/golden/train_small.npz), in the dtype of the loaded library (MVD_DTYPE=f16 | bf16), plus a short optimisation run.
Prints ONE JSON line: tests/test_gpu_train_bf16.py asserts on it, `python tools/train_dtype_check.py` shows it."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morphablediffusion_amd import lib as L  # noqa: E402
from morphablediffusion_amd.spec import VolumeConfig  # noqa: E402
from tests import golden_inputs as gi  # noqa: E402
from tests.test_gpu_train import P, _grad_report, _inputs, _unet_range, make_train_model  # noqa: E402


def main():
    dtype = L.load().mvd_compute_dtype().decode()
    g, dev, prepared, draws = _inputs()
    N = int(g["N"])
    # bf16 has fp32's exponent range: no loss scaling (scale 1); fp16 keeps the default dynamic scale
    ls = 1.0 if dtype == "bf16" else 65536.0
    m = make_train_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, loss_scale=ls, recompute=False)
    m.engine.zero_grad()
    loss = float(m.training_step(dev, prepared=prepared, **draws))
    want = float(np.asarray(g["loss.full"])[0])
    a, b, _ = gi.unpack_compare(m.last_noise_predict.detach().float().cpu(), g, "noise_predict")
    pred = ((a - b).norm() / b.norm()).item()
    rows = _grad_report(m, g, m.loss_scale)
    cond = sorted(r[0] for r in rows if r[2].startswith(("middle_conditions.", "output_conditions.")))
    rest = sorted(r[0] for r in rows if not r[2].startswith(("middle_conditions.", "output_conditions.")))
    # cosine between the whole UNet gradient and the reference's, over the sampled entries of the golden
    num = den_a = den_b = 0.0
    eng = m.engine
    for n in [str(x) for x in g["grad_names"]]:
        got = eng.param_view(P + n, grad=True).detach().float().cpu() / m.loss_scale
        if ("grad." + n + ".full") not in g and ("grad." + n + ".sample") not in g:
            continue
        x, y, _ = gi.unpack_compare(got, g, "grad." + n)
        num += float((x.double() * y.double()).sum())
        den_a += float((x.double() ** 2).sum())
        den_b += float((y.double() ** 2).sum())
    cos = num / (den_a ** 0.5 * den_b ** 0.5 + 1e-300)
    g1 = eng.flat_grads.clone()
    hi = _unet_range(eng)
    eng.zero_grad()
    loss2 = float(m.training_step(dev, prepared=prepared, **draws))
    repro = loss2 == loss and bool(torch.equal(eng.flat_grads[:hi], g1[:hi]))
    # a short optimisation run on the same batch: the loss must go down, the optimiser must not skip steps
    m.learning_rate = 2e-4
    (opt,), _ = m.configure_optimizers()
    for gr in opt.param_groups:
        gr["lr"] = 2e-4 if gr is opt.param_groups[0] else 2e-3
    losses = []
    for _ in range(6):
        opt.zero_grad()
        losses.append(float(m.training_step(dev, prepared=prepared, **draws)))
        opt.step()
    out = {"dtype": dtype, "loss": loss, "loss_ref": want, "loss_rel_err": abs(loss - want) / want, "pred_rel_l2": pred,
           "grad_trunk_worst": rest[-1], "grad_trunk_median": rest[len(rest) // 2], "grad_dt_worst": cond[-1],
           "grad_dt_median": cond[len(cond) // 2], "grad_cosine": cos, "bit_reproducible": repro, "loss_scale": m.loss_scale,
           "losses": losses, "steps_skipped": opt.steps_skipped,
           "finite": bool(torch.isfinite(eng.flat_params).all() and torch.isfinite(eng.flat_grads).all())}
    m.engine.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
