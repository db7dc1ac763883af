"""Isolated reproduction attempt of the side-stream non-determinism: engine B computes the frustum volumes
(frustum gather + FrustumTV3DNet) on its own stream while engine A -- a separate context with its own workspace -- keeps
the GPU busy with ONE kind of kernel on another stream.  Any difference from the volumes computed on an idle GPU is a
corruption caused purely by co-execution.  Round-2 result (MI355X): only the LDS-DMA kernels (conv3_dma, gemm_dma) as
aggressor change the frustum volumes (15-20 of 20 runs); GroupNorm, copies and short kernels never do; tools/race_micro2.py
shows that single torch / engine kernels are not victims -- see DESIGN.md section 4 ("Side stream: what the race is")."""
import sys, os, torch
sys.path.insert(0, ".")
from tests import golden_inputs as gi
from tests.test_gpu_model import make_model, to_dev
from morphablediffusion_amd import synthetic
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig

N = 16
m = make_model(gi.SMALL_UNET, VolumeConfig(num_views=N), N, workspace_gb=12.0)
batch = to_dev(synthetic.make_batch(N, "perspective", 5023, mesh_seed=1))
x_T, _, _ = [t.cuda() for t in synthetic.make_latents(N, 32, seed=6033)]
ts = torch.full((1,), 481, dtype=torch.long, device="cuda")
t_embed, v_embed = m.embed_time(ts), m.get_viewpoint_embedding(batch)
sv = m.spatial_volume.construct_spatial_volume(x_T, t_embed, v_embed, batch)
idx = torch.arange(N)[None]
frustum = lambda: m.spatial_volume.construct_view_frustum_volume(sv, t_embed, v_embed, idx, batch)[0]
ref = frustum()
torch.cuda.synchronize()

eA = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
g = torch.Generator().manual_seed(0)
x = torch.randn(32, 320, 16, 16, generator=g).cuda()
w = (torch.randn(640, 320, 3, 3, generator=g) * 0.02).cuda()
a = torch.randn(8192, 640, generator=g).cuda()
wl = (torch.randn(1280, 640, generator=g) * 0.03).cuda()
xg = torch.randn(32, 640, 16, 16, generator=g).cuda()
gam, bet = torch.ones(640).cuda(), torch.zeros(640).cuda()
loads = {
    "idle": lambda: None,
    "halo conv (conv3_dma)": lambda: eA.op_conv(x, w),
    "halo conv split-K 2": lambda: eA.op_conv(x, w, force_splitk=2),
    "lds-dma gemm (gemm_dma)": lambda: eA.op_linear(a, wl, a_half=True),
    "gemm_dma 3x3 stride-2 conv (gathered taps, padded borders)": lambda: eA.op_conv(x, w, stride=2),
    "group norm": lambda: eA.op_group_norm(xg, 32, gam, bet, 1e-5, 1),
    "torch copy": lambda: a.clone(),
}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in loads.items():
    bad = 0
    for rep in range(20):
        torch.cuda.synchronize()
        with torch.cuda.stream(sA):
            for _ in range(12):
                fn()
        with torch.cuda.stream(sB):
            out = frustum()
        torch.cuda.synchronize()
        if not all(torch.equal(out[k], ref[k]) for k in ref):
            bad += 1
    print(f"{name:28s}: {bad} of 20 frustum volumes differ from the idle-GPU result")
