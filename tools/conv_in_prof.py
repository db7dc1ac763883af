"""Kernel-level timing of the UNet's first convolution: the exact-fp32 vector-ALU form (conv_in_f32_kernel) beside the MFMA form,
both through mvd_op_conv at the headline shape (32 samples, 8 -> 320 channels, 32 x 32).  Run under
  rocprofv3 --kernel-trace --stats -d <dir> -- python tools/conv_in_prof.py
and read the two kernels' average durations from the stats table."""
import torch
from morphablediffusion_amd.engine import Engine
from morphablediffusion_amd.spec import UNetConfig, VolumeConfig

e = Engine(UNetConfig(model_channels=64), VolumeConfig(), workspace_gb=2.0)
torch.manual_seed(0)
x, w, b = torch.randn(32, 8, 32, 32, device="cuda"), torch.randn(320, 8, 3, 3, device="cuda") * 0.2, torch.randn(320, device="cuda")
for _ in range(20):
    e.op_conv(x, w, b, force_splitk=-2)
    e.op_conv(x, w, b)
torch.cuda.synchronize()
e.close()
