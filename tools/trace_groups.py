"""Per-shape kernel times from a rocprofv3 kernel_trace.csv: dispatches whose name contains argv[2], in launch order,
averaged in consecutive groups of argv[3] (the first of every group = warm-up, dropped)."""
import csv, sys
path, pat, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if pat in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
names = [r["Kernel_Name"] for r in rows]
out = []
for i in range(0, len(d) - n + 1, n):
    g = d[i + 1:i + n]
    out.append(f"{sum(g) / len(g):.1f}")
print(" ".join(out))
