"""Build-container experiment: how far is an fp16-operand / fp32-accumulate UNet from the fp32 reference?

Runs the imported reference UNet (full width) on CPU in fp32, then again with every GEMM-shaped op's
operands (and optionally every layer output) rounded to fp16, and prints error norms.  Decides the
storage policy documented in DESIGN.md.  Not shipped, not used by tests.
"""
import sys, time
sys.path.insert(0, 'tools')
import torch, torch.nn as nn, torch.nn.functional as F
import ref_import
ns = ref_import.import_reference()

torch.manual_seed(0)
cfg = dict(volume_dims=[64,128,256,512], image_size=32, in_channels=8, out_channels=4, model_channels=320,
           attention_resolutions=[4,2,1], num_res_blocks=2, channel_mult=[1,2,4,4], num_heads=8,
           use_spatial_transformer=True, transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False)
m = ns.attention.DepthWiseAttention(**cfg).eval()
g = torch.Generator().manual_seed(1)
for name, p in m.named_parameters():
    if p.abs().max() == 0 and name.endswith('weight'):
        fan_in = p[0].numel()
        p.data = (torch.rand(p.shape, generator=g) * 2 - 1) * (1.0 / fan_in) ** 0.5
Bv = 2
x = torch.randn(Bv, 8, 32, 32, generator=g)
t = torch.tensor([481, 481])
ctx = torch.randn(Bv, 1, 768, generator=g)
sd = {32: torch.randn(Bv, 64, 48, 32, 32, generator=g), 16: torch.randn(Bv, 128, 24, 16, 16, generator=g),
      8: torch.randn(Bv, 256, 12, 8, 8, generator=g), 4: torch.randn(Bv, 512, 6, 4, 4, generator=g)}
sd[32][1] = 0; sd[16][1] = 0; sd[8][1] = 0; sd[4][1] = 0; ctx[1] = 0

def r16(z):
    return z.half().float()

with torch.no_grad():
    t0 = time.time(); ref = m(x, t, ctx, source_dict=sd); print('fp32 fwd', time.time() - t0, 's; out rms', ref.pow(2).mean().sqrt().item(), 'max', ref.abs().max().item())

    orig = dict(conv2d=F.conv2d, conv3d=F.conv3d, linear=F.linear, einsum=ns.mattention.einsum, gn=F.group_norm, ln=F.layer_norm)
    def patch(round_out):
        o = (lambda z: r16(z)) if round_out else (lambda z: z)
        F.conv2d = lambda i, w, b=None, *a, **k: o(orig['conv2d'](r16(i), r16(w), b, *a, **k))
        F.conv3d = lambda i, w, b=None, *a, **k: o(orig['conv3d'](r16(i), r16(w), b, *a, **k))
        F.linear = lambda i, w, b=None: o(orig['linear'](r16(i), r16(w), b))
        ns.mattention.einsum = lambda eq, a, b: o(orig['einsum'](eq, r16(a), r16(b)))
        if round_out:
            F.group_norm = lambda *a, **k: r16(orig['gn'](*a, **k))
            F.layer_norm = lambda *a, **k: r16(orig['ln'](*a, **k))
    def unpatch():
        F.conv2d, F.conv3d, F.linear = orig['conv2d'], orig['conv3d'], orig['linear']
        ns.mattention.einsum = orig['einsum']; F.group_norm = orig['gn']; F.layer_norm = orig['ln']
    def report(tag, out):
        d = out - ref
        print(f'{tag}: max|d|/max|ref| = {(d.abs().max()/ref.abs().max()).item():.3e}  relL2 = {(d.norm()/ref.norm()).item():.3e}  '
              f'max elementwise rel(|ref|>0.1max) = {(d.abs()/ref.abs())[ref.abs()>0.1*ref.abs().max()].max().item():.3e}')
    patch(False); outB = m(x, t, ctx, source_dict=sd); unpatch(); report('policy B (fp16 operands, fp32 storage)', outB)
    hooks = []
    patch(True)
    for mod in m.modules():
        if isinstance(mod, (ns.openaimodel.ResBlock, ns.mattention.BasicTransformerBlock, ns.mattention.SpatialTransformer, ns.attention.DepthTransformer)):
            hooks.append(mod.register_forward_hook(lambda mod, i, o: r16(o)))
    outA = m(x, t, ctx, source_dict=sd); unpatch(); report('policy A (fp16 everywhere)', outA)
