"""What a pure streaming pass costs on this box: fp32 -> fp16 cast of [rows][C] tensors (4 B read + 2 B write per element, the traffic
of a GroupNorm apply), torch's own elementwise kernel and the library's rows_f32_to_f16 -- the floor under any GroupNorm form."""
import time, torch
for rows, C in [(32768, 320), (32768, 640), (32768, 960), (8192, 640), (8192, 1280), (2048, 1280)]:
    x = torch.randn(rows, C, device="cuda")
    for _ in range(5):
        y = x.half()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        y = x.half()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 * 1e6
    print(f"{rows:6d} x {C:5d}: torch .half() {us:6.1f} us  = {rows * C * 6 / us / 1e6:6.2f} TB/s")
