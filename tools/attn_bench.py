"""Self-attention kernel timing at the UNet's three shapes (B = 16 views x CFG)."""
import sys, time
import torch
sys.path.insert(0, "/root/repo")
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig
e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=4.0)
for (B, T, heads, d) in ((32, 1024, 8, 40), (32, 256, 8, 80), (32, 64, 8, 160)):
    C = heads * d
    q, k, v = (torch.randn(B, T, C, device="cuda") for _ in range(3))
    e.op_attention(q, k, v, heads); torch.cuda.synchronize()
    # op_attention converts layouts around the kernel; time the kernel through the profiler-free difference of two runs
    t0 = time.perf_counter()
    for _ in range(10): e.op_attention(q, k, v, heads)
    torch.cuda.synchronize()
    print(f"B={B} T={T} d={d}: {(time.perf_counter()-t0)/10*1e6:.0f} us per op_attention call (incl. layout conversion)")
