"""Bitwise repeatability of the GEMM / conv kernels while another stream saturates HBM (exposes hand-counted DMA waits
that are too loose: quiet runs hide them because the data always arrives early)."""
import sys, torch
sys.path.insert(0, ".")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=8.0)
g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
big_a = torch.empty(1 << 28, device="cuda"); big_b = torch.empty(1 << 28, device="cuda")  # 1 GiB each
idx = torch.randint(0, 1 << 22, (1 << 22,), device="cuda")
tab = torch.empty(1 << 22, 64, device="cuda")
def noise(n=6):
    with torch.cuda.stream(side):
        for _ in range(n):
            tab.index_select(0, idx)        # random 256-byte gathers over 1 GiB (latency spikes for everyone)
            big_a.copy_(big_b)
cases = {}
x = torch.randn(8, 320, 32, 32, generator=g).cuda(); w = (torch.randn(320, 320, 3, 3, generator=g) * 0.02).cuda()
cases["conv3x3 halo 320->320 @32 B8"] = lambda: e.op_conv(x, w)
x2 = torch.randn(8, 640, 16, 16, generator=g).cuda(); w2 = (torch.randn(640, 640, 3, 3, generator=g) * 0.02).cuda()
cases["conv3x3 halo 640->640 @16 B8 (split-K)"] = lambda: e.op_conv(x2, w2)
a = torch.randn(8192, 640, generator=g).cuda(); wl = (torch.randn(1280, 640, generator=g) * 0.03).cuda()
cases["linear 8192x640 -> 1280 (dma)"] = lambda: e.op_linear(a, wl, a_half=True)
a3 = torch.randn(2048, 1280, generator=g).cuda(); wl3 = (torch.randn(1280, 1280, generator=g) * 0.03).cuda()
cases["linear 2048x1280 -> 1280 (dma split-K)"] = lambda: e.op_linear(a3, wl3, a_half=True)
x3 = torch.randn(2, 64, 24, 32, 32, generator=g).cuda(); w3 = (torch.randn(128, 64, 3, 3, 3, generator=g) * 0.02).cuda()
cases["conv3d 64->128 s2"] = lambda: e.op_conv3d(x3, w3, stride=2)
x4 = torch.randn(2, 128, 12, 16, 16, generator=g).cuda(); w4 = (torch.randn(128, 128, 3, 3, 3, generator=g) * 0.02).cuda()
cases["conv3d 128->128 s1"] = lambda: e.op_conv3d(x4, w4)
xs = torch.randn(8, 128, 16, 16, generator=g).cuda(); ws_ = (torch.randn(128, 128, 3, 3, generator=g) * 0.03).cuda()
cases["conv3x3 128->128 @16 B8 (small UNet level 1)"] = lambda: e.op_conv(xs, ws_)
xs2 = torch.randn(8, 64, 32, 32, generator=g).cuda(); ws2 = (torch.randn(128, 64, 3, 3, generator=g) * 0.03).cuda()
cases["conv3x3 s2 64->128 @32 B8 (downsample)"] = lambda: e.op_conv(xs2, ws2, stride=2)
al = torch.randn(2048, 128, generator=g).cuda(); wll = (torch.randn(1024, 128, generator=g) * 0.05).cuda()
cases["linear 2048x128 -> 1024 geglu"] = lambda: e.op_linear(al, wll, geglu=True, a_half=True)
wl5 = (torch.randn(384, 128, generator=g) * 0.05).cuda()
cases["linear 2048x128 -> 384 (qkv)"] = lambda: e.op_linear(al, wl5, a_half=True)
al6 = torch.randn(2048, 512, generator=g).cuda(); wl6 = (torch.randn(128, 512, generator=g) * 0.05).cuda(); r6 = torch.randn(2048, 128, generator=g).cuda()
cases["linear 2048x512 -> 128 + resid (ff2)"] = lambda: e.op_linear(al6, wl6, resid=r6, a_half=True)
q = torch.randn(8, 256, 128, generator=g).cuda()
cases["attention T=256 d=16"] = lambda: e.op_attention(q, q * 0.5, q * 0.25, 8)
gx = torch.randn(8, 128, 16, 16, generator=g).cuda(); gg_ = torch.randn(128, generator=g).cuda()
cases["group norm 128 @16"] = lambda: e.op_group_norm(gx, 32, gg_, gg_, 1e-5, 1)
for name, fn in cases.items():
    ref = fn(); torch.cuda.synchronize()
    bad = 0
    for i in range(60):
        noise()
        o = fn()
        torch.cuda.synchronize()
        if not torch.equal(o, ref):
            bad += 1
            if bad <= 2: print("   diff", (o - ref).abs().max().item(), "of", ref.abs().max().item())
    print(f"{name}: {bad} of 60 differ under load")
