"""Do two HIP streams of one process overlap on this box at all?  Chains of small kernels (a 64-workgroup elementwise kernel and a
small matmul) on one stream vs split over two streams; eager and as two hipGraphs.  python tools/stream_overlap_micro.py"""
import os, time, torch
dev = "cuda:0"
n = 64 * 256 * 4
xs = [torch.randn(n, device=dev) for _ in range(2)]
ms = [torch.randn(512, 512, device=dev, dtype=torch.float16) for _ in range(2)]


def chain(i, reps=300):
    x, m = xs[i], ms[i]
    for _ in range(reps):
        x = x * 1.0001 + 0.5
        m = (m @ m) * 0.01
    return x, m


def run(streams, reps=5):
    for st in streams:
        with torch.cuda.stream(st):
            chain(0, 10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                chain(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
print(f"eager: one chain on one stream        {run([s0]):8.3f} ms")
print(f"eager: two chains on ONE stream       {run([s0, s0]):8.3f} ms")
print(f"eager: two chains on TWO streams      {run([s0, s1]):8.3f} ms")
gs = []
for i in range(2):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chain(i)
    gs.append(g)
torch.cuda.synchronize()


def rung(pairs, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, st in pairs:
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


print(f"graph: one chain                      {rung([(gs[0], s0)]):8.3f} ms")
print(f"graph: two chains on ONE stream       {rung([(gs[0], s0), (gs[1], s0)]):8.3f} ms")
print(f"graph: two chains on TWO streams      {rung([(gs[0], s0), (gs[1], s1)]):8.3f} ms")
