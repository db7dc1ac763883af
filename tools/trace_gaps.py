"""Timeline of one denoising step from a rocprofv3 --kernel-trace CSV: per launch start / duration / gap to the previous
launch on the same queue, and totals (kernel time, idle gaps, per-queue overlap).

  rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras
  python tools/trace_gaps.py DIR/.../t_kernel_trace.csv [step_index] > timeline.txt
Steps are delimited by build_cfg_input_kernel (one launch per step)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:44]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    step_pick = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    ev = []
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    ev.sort()
    marks = [i for i, e in enumerate(ev) if "build_cfg_input_kernel" in e[2]]
    if len(marks) < 2:
        print("fewer than two steps in the trace")
        return
    pick = step_pick if step_pick >= 0 else len(marks) - 1 + step_pick
    a, b = marks[pick], marks[pick + 1]
    step = ev[a:b]
    t0 = step[0][0]
    wall = (ev[b][0] - t0) / 1e3
    last_end = {}
    ksum = 0.0
    gap_sum = defaultdict(float)
    fam = defaultdict(lambda: [0, 0.0])
    # union of busy intervals over all queues
    busy, cur_s, cur_e = 0.0, None, None
    for s, e, n, q in step:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"# step {pick}: {len(step)} launches, wall {wall:.1f} us, GPU busy (union over queues) {busy / 1e3:.1f} us, idle {wall - busy / 1e3:.1f} us")
    lines = []
    for s, e, n, q in step:
        d = (e - s) / 1e3
        ksum += d
        g = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        if 0 < g < 100:
            gap_sum[q] += g
        f = short(n)
        fam[f][0] += 1
        fam[f][1] += d
        lines.append(f"{(s - t0) / 1e3:9.1f} q{q:>2} {d:7.1f} us  gap {g:6.1f}  {f}")
    print(f"# kernel time {ksum:.1f} us; gaps per queue: " + ", ".join(f"q{q}: {v:.1f} us" for q, v in gap_sum.items()))
    for f, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"#   {t:8.1f} us {c:4d} x {t / c:6.1f}  {f}")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
