"""Per-layer version of precision_probe3: one forward per GEMM-shaped leaf (and per attention module) with ONLY that layer's
operands rounded to fp16; prints the layers sorted by their share of the end-to-end error variance.
  python tools/precision_probe4.py [small|full] [init|trained] [topN]"""
import sys
sys.argv = sys.argv[:3] + ["--"] if len(sys.argv) < 4 else sys.argv
import runpy
import io, contextlib
top = int(sys.argv[3]) if sys.argv[3] != "--" else 40
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    g = runpy.run_path("tools/precision_probe3.py")
m, run, rel, allerr = g["m"], g["run"], g["rel"], g["allerr"]
leaf, attn = g["leaf_gemm"], g["attn_mods"]
rows = []
inside_attn = lambda n: any(n.startswith(a + ".") for a in attn)
for n in leaf:
    if inside_attn(n):
        continue
    rows.append((n, rel(run(lambda k, n=n: k == n))))
for a in attn:
    rows.append((a + " [attention: projections + softmax operands]", rel(run(lambda k, a=a: k == a or k.startswith(a + ".")))))
rows.sort(key=lambda r: -r[1])
print(f"all: {allerr:.3e}")
cum = 0.0
for n, e in rows[:top]:
    cum += 100 * e * e / allerr ** 2
    print(f"{n:70s} {e:.3e}  share {100 * e * e / allerr ** 2:5.1f} %  cum {cum:5.1f} %")
