"""Build-container experiment (CPU, imported reference): WHERE does the fp16-operand error of the UNet come from?

Every GEMM-shaped op (conv2d / conv3d / linear / einsum) rounds its operands to fp16 only while a module of the selected
group is executing; everything else stays fp32.  Error variances of the groups add (approximately), so
(relL2 of group)^2 / (relL2 of all)^2 is the group's share of the end-to-end error.  Decides which layers get extended
precision (precision policy).  Not shipped, not used by tests.

  python tools/precision_probe3.py [small|full] [init|trained]
"""
import re
import sys
import time

sys.path.insert(0, "tools")
sys.path.insert(0, ".")
import torch
import torch.nn.functional as F

import ref_import
from tests import golden_inputs as gi

ns = ref_import.import_reference()
import make_goldens as mg

width = sys.argv[1] if len(sys.argv) > 1 else "small"
style = sys.argv[2] if len(sys.argv) > 2 else "init"
cfg = gi.SMALL_UNET if width == "small" else gi.FULL_UNET
m = mg.load_unet(ns, cfg, style)
x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2, seed=13 if style == "trained" else 11)
r16 = lambda z: z.half().float()
orig = dict(conv2d=F.conv2d, conv3d=F.conv3d, linear=F.linear, einsum=ns.mattention.einsum, dein=None)
import ldm.models.diffusion.attention as datt
STATE = {"on": 0}
A = lambda z: r16(z) if STATE["on"] > 0 else z
F.conv2d = lambda i, w, b=None, *a, **k: orig["conv2d"](A(i), A(w), b, *a, **k)
F.conv3d = lambda i, w, b=None, *a, **k: orig["conv3d"](A(i), A(w), b, *a, **k)
F.linear = lambda i, w, b=None: orig["linear"](A(i), A(w), b)
ns.mattention.einsum = lambda eq, a, b: orig["einsum"](eq, A(a), A(b))

names = {mod: n for n, mod in m.named_modules()}
leaf_gemm = [n for n, mod in m.named_modules() if isinstance(mod, (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.Linear))]
attn_mods = [n for n, mod in m.named_modules() if isinstance(mod, (ns.mattention.CrossAttention, datt.DepthAttention))]


def run(selector):
    """selector(name) -> True: GEMMs executed inside that module are rounded."""
    hooks = []
    for n, mod in m.named_modules():
        if n and selector(n) and (n in leaf_gemm or n in attn_mods):
            hooks.append(mod.register_forward_pre_hook(lambda *_: STATE.__setitem__("on", STATE["on"] + 1)))
            hooks.append(mod.register_forward_hook(lambda *_: STATE.__setitem__("on", STATE["on"] - 1)))
    with torch.no_grad():
        o = m(x, t, ctx, source_dict=sd)
    for h in hooks:
        h.remove()
    assert STATE["on"] == 0
    return o


with torch.no_grad():
    ref = m(x, t, ctx, source_dict=sd)
rel = lambda o: ((o - ref).norm() / ref.norm()).item()
allerr = rel(run(lambda n: True))
print(f"{width}/{style}: all GEMM operands fp16: relL2 = {allerr:.3e}")
groups = {
    "time_embed + emb_layers (per-sample linears)": lambda n: n.startswith("time_embed") or ".emb_layers." in n,
    "conv_in (input_blocks.0)": lambda n: n.startswith("input_blocks.0."),
    "out conv": lambda n: n.startswith("out."),
    "ResBlock convs (in/out_layers, skip)": lambda n: bool(re.search(r"\.(in_layers|out_layers|skip_connection)", n)),
    "ST proj_in/proj_out": lambda n: bool(re.search(r"\.\d+\.1\.(proj_in|proj_out)$", n)) or bool(re.search(r"middle_block\.1\.(proj_in|proj_out)$", n)),
    "ST attn1 (q,k,v,out + softmax operands)": lambda n: ".attn1" in n,
    "ST attn2": lambda n: ".attn2" in n,
    "ST feed-forward": lambda n: ".ff." in n,
    "down/up sample convs": lambda n: n.endswith(".op") or bool(re.search(r"\.\d+\.\d+\.conv$", n)),
    "DepthTransformers (all)": lambda n: "conditions" in n,
    "  depth proj_in": lambda n: "conditions" in n and ".proj_in." in n,
    "  depth proj_context": lambda n: "conditions" in n and ".proj_context." in n,
    "  depth attention (q,k,v,out)": lambda n: "conditions" in n and ".depth_attn" in n,
    "  depth proj_out convs": lambda n: "conditions" in n and ".proj_out." in n,
    "level-0 blocks only (input 1-2, output 9-11)": lambda n: bool(re.match(r"(input_blocks\.[12]\.|output_blocks\.(9|10|11)\.)", n)),
    "output_blocks.11 + out": lambda n: n.startswith("output_blocks.11.") or n.startswith("out."),
}
tot = 0.0
for tag, sel in groups.items():
    t0 = time.time()
    e = rel(run(sel))
    print(f"  {tag:55s} relL2 = {e:.3e}   share = {100 * e * e / allerr ** 2:5.1f} %   ({time.time() - t0:.0f} s)")
