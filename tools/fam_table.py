"""Family table of a bench.py JSON line (the engine's event probe): python tools/fam_table.py bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.2f} {d['unit']}  dominant: {d['roofline'].get('kernel')} frac {d['roofline']['frac']:.3f}")
tot = sum(f["ms_per_step"] for f in d["families"])
for f in d["families"]:
    print(f"{f['family']:32s} {f['ms_per_step']:7.3f} ms {100*f['ms_per_step']/tot:5.1f}% {f['launches_per_step']:6.1f} launches "
          f"{1e3*f['ms_per_step']/max(f['launches_per_step'],1):6.1f} us avg {f.get('tflops',0):7.1f} TF/s {f.get('gbs',0):7.1f} GB/s")
print(f"sum of bracketed families {tot:.3f} ms")
