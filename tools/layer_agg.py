"""Aggregate MVD_LAYER_TIMING lines (stderr of bench.py) of the LAST `denoise` pass: identical GEMM descriptors summed."""
import re, sys, collections
raw = open(sys.argv[1]).read()
if "[step-begin]" in raw:       # tools/layer_step.py marks the steps: take the last one
    raw = raw.rsplit("[step-begin]", 1)[1]
lines = [l for l in raw.splitlines() if l.startswith("[gemm]")]
n = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else len(lines)   # GEMM launches per step
last = lines[-n:]
agg = collections.OrderedDict()
for l in last:
    m = re.match(r"\[gemm\] (.*?)\s+([\d.]+) us\s+(\d+) TF", l)
    k, us, tf = m.group(1), float(m.group(2)), int(m.group(3))
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += us; a[2] += us * tf
tot = sum(a[1] for a in agg.values())
print(f"{len(last)} GEMM launches, {tot / 1e3:.2f} ms")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{a[1]:8.1f} us {a[0]:3d} x {a[1] / a[0]:7.1f} us {a[2] / a[1]:5.0f} TF  {k}")
