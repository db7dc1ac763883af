"""Ablation on the seeded full-width UNet (same weights/inputs as tests/golden/unet_full.npz): which fp16
roundings dominate the end-to-end error?  CPU only, build container."""
import sys, time
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import ref_import
from tests import golden_inputs as gi
ns = ref_import.import_reference()
import make_goldens as mg
cfg = gi.FULL_UNET
m = mg.load_unet(ns, cfg)
x, t, ctx, sd = gi.unet_inputs(cfg, Bv=2)
r16 = lambda z: z.half().float()
orig = dict(conv2d=F.conv2d, conv3d=F.conv3d, linear=F.linear, einsum=ns.mattention.einsum)
def patch(wr, ar, small_exact):
    W = r16 if wr else (lambda z: z)
    A = r16 if ar else (lambda z: z)
    F.conv2d = lambda i, w, b=None, *a, **k: orig['conv2d'](A(i), W(w), b, *a, **k)
    F.conv3d = lambda i, w, b=None, *a, **k: orig['conv3d'](A(i), W(w), b, *a, **k)
    def lin(i, w, b=None):
        if small_exact and i.dim() == 2 or (small_exact and i.shape[-2] == 1):   # per-sample vectors (emb, attn2 k/v)
            return orig['linear'](i, w, b)
        return orig['linear'](A(i), W(w), b)
    F.linear = lin
    ns.mattention.einsum = lambda eq, a, b: orig['einsum'](eq, A(a), A(b))
def unpatch():
    F.conv2d, F.conv3d, F.linear = orig['conv2d'], orig['conv3d'], orig['linear']; ns.mattention.einsum = orig['einsum']
with torch.no_grad():
    ref = m(x, t, ctx, source_dict=sd)
    for tag, args in [('weights+acts fp16', (1,1,0)), ('weights+acts fp16, per-sample linears exact', (1,1,1)),
                      ('weights only fp16', (1,0,0)), ('acts only fp16', (0,1,0))]:
        patch(*args); o = m(x, t, ctx, source_dict=sd); unpatch()
        print(f'{tag}: relL2 = {((o-ref).norm()/ref.norm()).item():.3e}')
