"""Which kernels are VICTIMS when an LDS-DMA GEMM (engine A, own context / workspace / stream) co-executes?"""
import sys, torch
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig

eA = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
eB = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
g = torch.Generator().manual_seed(0)
a = torch.randn(8192, 640, generator=g).cuda()
wl = (torch.randn(1280, 640, generator=g) * 0.03).cuda()
big = torch.randn(1 << 25, generator=g).cuda()              # 128 MB
x3 = torch.randn(4, 64, 24, 32, 32, generator=g).cuda()
w3 = (torch.randn(64, 64, 3, 3, 3, generator=g) * 0.02).cuda()
xg = torch.randn(16, 64, 48 * 32 * 32 // 16, generator=g).cuda().reshape(16, 64, 96, 32)  # two-pass GroupNorm (49152*... rows)
gam, bet = torch.ones(64).cuda(), torch.zeros(64).cuda()
xs = torch.randn(32, 128, 16, 16, generator=g).cuda()
ws = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
victims = {
    "torch elementwise (x*1.5+1, 128 MB)": lambda: big * 1.5 + 1.0,
    "torch copy (128 MB)": lambda: big.clone(),
    "engine B: 3-D conv (igemm, 64 KB LDS, register staged)": lambda: eB.op_conv3d(x3, w3),
    "engine B: GroupNorm (two-pass, 20-25 KB static LDS)": lambda: eB.op_group_norm(xg, 8, gam, bet, 1e-5, 1),
    "engine B: 3x3 conv 128ch @16x16 (halo conv, LDS-DMA)": lambda: eB.op_conv(xs, ws),
}
aggressors = {"LDS-DMA GEMM": lambda: eA.op_linear(a, wl, a_half=True), "torch matmul": lambda: a @ wl.t()}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
for an, af in aggressors.items():
    for vn, vf in victims.items():
        with torch.cuda.stream(sB):
            ref = vf()
        torch.cuda.synchronize()
        bad, worst = 0, 0.0
        for rep in range(25):
            with torch.cuda.stream(sA):
                for _ in range(12):
                    af()
            with torch.cuda.stream(sB):
                out = vf()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad += 1
                d = (out != ref)
                worst = max(worst, d.float().mean().item())
        print(f"aggressor {an:14s} victim {vn:58s}: {bad:2d} of 25 differ (max fraction of elements {worst:.2e})")
