"""Per-kernel-family table of one bench.py run under rocprofv3, regenerated from files under profiles/:

  python tools/prof_summary.py <kernel_stats.csv | results.db> <bench.json> [total_steps] [out.json]

* kernel time, launches, average duration per family come from rocprofv3 (--kernel-trace --stats; CSV or the SQLite db);
* algorithmic FLOPs / bytes per step and family come from the bench line's "families" (booked by the engine's probe at each
  launch: every operand and the result once);
* TF/s, GB/s and the fraction of the binding roof (2.5 PFLOP/s dense fp16 MFMA, 8 TB/s HBM) follow.
total_steps = identical denoising steps the profiled process executed (warmup + 2 survey + timed, with --no-extras)."""
import csv
import json
import re
import sys

PEAK_TF, PEAK_GBS = 2500.0, 8000.0
path, bench = sys.argv[1], json.load(open(sys.argv[2]))
extra = sys.argv[3:]
json_out = next((a for a in extra if a.endswith(".json")), None)  # optional: stamped per-family durations for bench.py's frac_rocprof
nums = [a for a in extra if not a.endswith(".json")]
steps = float(nums[0]) if nums else bench["steps"] + bench["warmup"] + 2


def rows_from(path):
    if path.endswith(".db"):
        import sqlite3
        c = sqlite3.connect(path)
        return [(n, k, t) for n, k, t in c.execute("select name, count(*), sum(end-start) from kernels group by name")]
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])))
    return out


def family(name):
    m = re.search(r"(gemm_dma_kernel|conv3_dma_kernel|igemm_kernel)<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(2).split(",")]
        if m.group(1) == "igemm_kernel":
            return f"igemm_kernel<{1 if a[0] == 'true' else 0},{a[1]}>"
        if m.group(1) == "gemm_dma_kernel":
            # <BM, BN, MODE, PLAIN>: the engine books the plain and the general instance under one family, the four-wave
            # 128-row tiles as gemm_dma_kernel<128xBN>
            if len(a) >= 4:
                return f"gemm_dma_kernel<128x{a[1]}>" if a[0] == "128" else f"gemm_dma_kernel<{a[1]},{a[2]}>"
            a = a[:2]
        return f"{m.group(1)}<{','.join(a)}>"
    m = re.search(r"conv3x_kernelILi(\d+)E", name) or re.search(r"conv3x_kernel<\s*(\d+)", name)
    if m:  # mangled or demangled; the engine books conv3x_kernel<NF> for both tile geometries
        return f"conv3x_kernel<{m.group(1)}>"
    for key, fam in (("rowhead_kernel", "rowhead_kernel"), ("rowchain_kernel", "rowchain_kernel"), ("splitk_reduce", "splitk_reduce_kernel"), ("gn_", "group_norm"), ("layernorm", "layernorm"),
                     ("depth_attn", "depth_attn_kernel"), ("attn_kernel", "attn_kernel")):
        if key in name:
            return fam
    return None


skip = ("conv3x_pack", "rowchain_pack", "rowhead_pack", "bias_fold", "pack_weight", "fold_", "copyBuffer", "permute_geglu", "relu_beta", "f32_to_f16_kernel", "fill_pattern", "at::native",
        "pack_upconv", "fillBuffer")
fam_t, fam_n, other, tot, calls = {}, {}, [], 0.0, 0
for name, n, t in rows_from(path):
    if any(s in name for s in skip):
        continue
    tot += t
    calls += n
    f = family(name)
    if f is None:
        other.append((t, n, name))
        continue
    fam_t[f] = fam_t.get(f, 0.0) + t
    fam_n[f] = fam_n.get(f, 0) + n
alg = {f["family"]: f for f in bench.get("families", [])}
print(f"per step: {tot / steps / 1e6:.2f} ms of kernel time in {calls / steps:.0f} launches; step wall time {bench['ms_per_step']:.2f} ms "
      f"({bench['value']:.1f} steps/s)")
print(f"{'family':30s} {'ms/step':>8s} {'share':>6s} {'launch/step':>11s} {'us avg':>7s} {'GFLOP/step':>10s} {'TF/s':>6s} {'MB/step':>8s} "
      f"{'GB/s':>6s} {'roof':>5s} {'frac':>5s}")
for f, t in sorted(fam_t.items(), key=lambda kv: -kv[1]):
    ms = t / steps / 1e6
    a = alg.get(f)
    line = f"{f:30s} {ms:8.3f} {100 * t / tot:5.1f}% {fam_n[f] / steps:11.1f} {t / fam_n[f] / 1e3:7.1f}"
    if a and a.get("tflops") is not None:
        # the probe's TF/s and GB/s are (work / event time); scale the work per step to rocprof's time
        gflop = a["tflops"] * a["ms_per_step"]          # TF/s * ms = GFLOP
        mb = a["gbs"] * a["ms_per_step"]                # GB/s * ms = MB
        tf, gbs = gflop / ms, mb / ms
        tfrac, bfrac = tf / PEAK_TF, gbs / PEAK_GBS
        roof = "mfma" if tfrac >= bfrac else "hbm"
        line += f" {gflop:10.1f} {tf:6.0f} {mb:8.1f} {gbs:6.0f} {roof:>5s} {max(tfrac, bfrac):5.3f}"
    print(line)
for t, n, name in sorted(other, reverse=True)[:10]:
    print(f"{name[:60]:60s} {t / steps / 1e6:8.3f} ms/step {n / steps:7.1f} launches/step")

# machine-readable: rocprofv3's average duration per family, stamped with the hash of the library sources and the workload --
# bench.py prints roofline.frac_rocprof from it (profiles/kernel_durations.json) next to its own event-bracket figure
if json_out:
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from morphablediffusion_amd.lib import csrc_sha16
    json.dump({"source": "rocprofv3 --kernel-trace --stats over bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras: TotalDurationNs / Calls "
                         "per kernel family (kernel begin to end, no launch path)",
               "config": bench.get("config", {}).get("name"), "csrc_sha16": csrc_sha16(),
               "us_per_launch": {f: fam_t[f] / fam_n[f] / 1e3 for f in fam_t}, "launches_per_step": {f: fam_n[f] / steps for f in fam_t},
               "kernel_ms_per_step": tot / steps / 1e6}, open(json_out, "w"), indent=1)
