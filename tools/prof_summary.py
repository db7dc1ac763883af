"""Summarise a rocprofv3 kernel_stats.csv: total kernel time, launches, top kernels (per bench step)."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
skip = ("pack_weight", "fold_", "copyBuffer", "permute_geglu", "relu_beta", "f32_to_f16", "fill_pattern", "at::native")
tot = calls = 0
out = []
for r in rows:
    if any(s in r["Name"] for s in skip):
        continue
    t, c = float(r["TotalDurationNs"]) / 1e6, int(r["Calls"])
    tot += t; calls += c
    out.append((t, c, r["Name"][:70]))
print(f"per step: kernel time {tot/steps:.2f} ms in {calls/steps:.0f} launches ({1e3*tot/calls:.1f} us avg)")
for t, c, n in sorted(out, reverse=True)[:14]:
    print(f"  {t/steps:7.3f} ms {c/steps:6.0f} x {1e3*t/c:7.1f} us  {n}")
