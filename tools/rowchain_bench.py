"""Times the row-chain kernel (csrc/k_rowchain.hip) at the headline shape (32 UNet samples x 32 x 32 pixels, C = 320) against
the layered path it replaces (op_bench-style, through the C ABI).  python tools/rowchain_bench.py [iters]"""
import sys
import torch
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
sys.path.insert(0, "tests")
from test_gpu_ops import _st_tail_case, _st_tail_ref

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=6.0)
C, rows, T = 320, 32768, 1024
for ao, po in ((False, False), (True, False), (True, True)):
    d = _st_tail_case(C, rows, T, ao, po)
    got, ms = e.op_st_tail(iters=iters, **d)
    sub = {k: (v[:4096] if torch.is_tensor(v) and v.shape[0] == rows else v) for k, v in d.items()}
    if ao:
        sub["rowbias"] = d["rowbias"][:4096 // T]
    want = _st_tail_ref(sub)
    err = ((got[:4096].cpu() - want).norm() / want.norm()).item()
    fl = 2.0 * rows * C * C * (12 + (1 if ao else 0) + (1 if po else 0))
    print(f"rowchain C={C} rows={rows} ao={int(ao)} po={int(po)}: {ms * 1e3:8.1f} us  {fl / ms * 1e-9:7.0f} TFLOP/s  "
          f"({fl / ms * 1e-9 / 2500:.3f} of peak)  relL2(first 4096 rows)={err:.2e}", flush=True)
d = _st_tail_case(C, rows, T, True, True)
got, ms = e.op_st_tail(iters=iters, xp_out=True, **d)
print(f"rowchain C={C} rows={rows} ao=1 po=2 (extended-precision proj_out): {ms * 1e3:8.1f} us", flush=True)
g = torch.Generator().manual_seed(3)
r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
t0, qkv, ms = e.op_st_head(r(rows, C), r(C, C, sc=C ** -0.5), r(C), 1 + 0.1 * r(C), 0.1 * r(C), r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5),
                           r(C, C, sc=C ** -0.5), iters=iters)
fl = 2.0 * rows * C * C * 4
print(f"rowhead  C={C} rows={rows} (proj_in, LayerNorm1, q|k|v): {ms * 1e3:8.1f} us  {fl / ms * 1e-9:7.0f} TFLOP/s", flush=True)
g = torch.Generator().manual_seed(3)
t0, qkv, ms = e.op_st_head(r(rows, C), r(C, C, sc=C ** -0.5), r(C), 1 + 0.1 * r(C), 0.1 * r(C), r(C, C, sc=C ** -0.5), r(C, C, sc=C ** -0.5),
                           r(C, C, sc=C ** -0.5), iters=iters, xp=True)
print(f"rowhead  C={C} rows={rows} extended-precision proj_in: {ms * 1e3:8.1f} us", flush=True)
e.close()
