"""Pretty one-line summary of a bench.py JSON line read from stdin (tag as argv[1])."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", "steps/s=%.2f ms/step=%.2f conv=%.0fTF views/gpu=%s" % (
    d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["config"]["views_per_gpu"]))
