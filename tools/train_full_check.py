"""The FULL-WIDTH training step against the reference's gradients (tests/golden/train_full.npz: training_step + loss.backward() of the
imported reference on the CPU, 916.9 M-parameter UNet, B = 2) in the dtype MVD_DTYPE selects -- one JSON line.  The fp16 run is
tests/test_gpu_train.py::test_training_step_full_width_gradients_vs_reference; this script exists so that the bfloat16 build
(BASELINE configs[3]'s dtype) is held to the same golden from a process of its own (tests/test_gpu_train_bf16.py)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morphablediffusion_amd import lib
from morphablediffusion_amd.spec import VolumeConfig
from tests import golden_inputs as gi
from tests.test_gpu_train import G, _grad_report, _train_inputs, make_train_model

g = np.load(os.path.join(G, "train_full.npz"))
batch, x0, x_in, clip, ts, noise, ti, dr = _train_inputs(g)
dev = {k: v.cuda() for k, v in batch.items()}
N = int(g["N"])
bf16 = lib.DTYPE == "bf16"
m = make_train_model(gi.FULL_UNET, VolumeConfig(num_views=N), N, workspace_gb=40.0, loss_scale=1.0 if bf16 else 65536.0, recompute=True)
m.train_conditioner = False
m.engine.zero_grad()
loss = m.training_step(dev, prepared=(x0.cuda(), clip.cuda(), {"x": x_in.cuda()}), time_steps=ts, noise=noise, target_index=ti,
                       drop_random=dr)
want = float(np.asarray(g["loss.full"])[0])
pred = m.last_noise_predict.float().cpu().numpy()
ref = np.asarray(g["noise_predict.full"]) if "noise_predict.full" in g.files else None
rows = sorted(_grad_report(m, g, m.loss_scale), reverse=True)
cond = [r for r in rows if r[2].startswith(("middle_conditions.", "output_conditions."))]
rest = [r for r in rows if not r[2].startswith(("middle_conditions.", "output_conditions."))]
out = {"dtype": lib.DTYPE, "loss_rel_err": abs(float(loss) - want) / want, "n_cond": len(cond), "n_rest": len(rest),
       "grad_dt_worst": max(r[0] for r in cond), "grad_dt_median": float(np.median([r[0] for r in cond])),
       "grad_trunk_worst": max(r[0] for r in rest), "grad_trunk_median": float(np.median([r[0] for r in rest])),
       "worst_dt": cond[0][2], "worst_trunk": rest[0][2]}
if ref is not None and ref.shape == pred.shape:
    out["pred_rel_l2"] = float(np.linalg.norm(pred - ref) / np.linalg.norm(ref))
print(json.dumps(out))
m.engine.close()
