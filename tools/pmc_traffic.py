"""HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter CSVs) over
`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras` (5 identical steps), next to the algorithmic bytes the bench line
books per family:  python tools/pmc_traffic.py counters_FETCH_SIZE.csv counters_WRITE_SIZE.csv bench_line.json [steps]
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md 'HBM'); WRITE_SIZE is uncalibrated on
gfx950 (same guide) and shown as reported.  Units: MB = 1e6 bytes per step on BOTH sides (the counters report KiB:
value x 1024 bytes; round 5 printed them as MiB beside algorithmic MB, which made every ratio 4.9 % low)."""
import collections, csv, json, re, sys

steps = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
json_out = sys.argv[5] if len(sys.argv) > 5 else None  # {family: HBM-side bytes per launch}: what bench.py reports as roofline.traffic
PAT = r"(gemm_dma_kernel|conv3_dma_kernel|igemm_kernel|conv3x_kernel)<[^>]*>|rowhead_kernel|rowchain_kernel|attn_kernel|depth_attn|gn_|layernorm|splitk_reduce|sparse_conv|target_encoder"


def fam_of(name):
    m = re.search(r"conv3x_kernelILi(\d+)ELi(\d+)", name) or re.search(r"conv3x_kernel<\s*(\d+)\s*,\s*(\d+)", name)
    if m:  # mangled or demangled: the engine books conv3x_kernel<NF> (both tile geometries)
        return f"conv3x_kernel<{m.group(1)}>"
    m = re.search(PAT, name)
    if not m:
        return "other"
    k = m.group(0)
    m2 = re.match(r"(gemm_dma_kernel)<([^>]*)>", k)
    if m2:  # <BM, BN, MODE, PLAIN> -> <BN,MODE> (256-row tiles) or <128xBN> (four-wave tiles), as the engine books it
        a = [x.strip() for x in m2.group(2).split(",")]
        if len(a) >= 4:
            k = f"gemm_dma_kernel<128x{a[1]}>" if a[0] == "128" else f"gemm_dma_kernel<{a[1]},{a[2]}>"
        else:
            k = f"gemm_dma_kernel<{a[0]},{a[1]}>"
    m3 = re.match(r"(conv3_dma_kernel)<([^>]*)>", k)
    if m3:
        k = "conv3_dma_kernel<" + ",".join(x.strip() for x in m3.group(2).split(",")) + ">"
    m5 = re.match(r"conv3x_kernel<([^>]*)>", k)
    if m5:
        k = "conv3x_kernel<" + m5.group(1).strip() + ">"
    m4 = re.match(r"igemm_kernel<([^>]*)>", k)
    if m4:
        a = [x.strip() for x in m4.group(1).split(",")]
        k = f"igemm_kernel<{1 if a[0] == 'true' else 0},{a[1]}>"
    return {"gn_": "group_norm", "depth_attn": "depth_attn_kernel", "splitk_reduce": "splitk_reduce_kernel"}.get(k, k)


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = fam_of(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])  # KB
        n[k].add(r["Dispatch_Id"])
    return tot, n


fe, nf = load(sys.argv[1], "FETCH_SIZE")
wr, _ = load(sys.argv[2], "WRITE_SIZE")
line = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
alg = {f["family"]: f.get("gbs", 0.0) * f["ms_per_step"] * 1e6 for f in line["families"]}  # GB/s x ms = MB -> bytes
print(f"{'family':34s} {'launches':>8s} {'fetch x2':>10s} {'write':>9s} {'algorithmic':>12s} {'(fetch x2 + write)/alg':>22s}")
for k in sorted(fe, key=lambda k: -fe[k]):
    f2, w = 2 * fe[k] * 1024.0 / 1e6 / steps, wr.get(k, 0.0) * 1024.0 / 1e6 / steps  # KiB -> bytes -> MB (1e6)
    a = alg.get(k)
    a_mb = a / 1e6 if a else None
    print(f"{k:34s} {len(nf[k]) / steps:8.1f} {f2:10.1f} {w:9.1f} {a_mb if a_mb is None else round(a_mb, 1)!s:>12s} "
          f"{'' if not a_mb else round((f2 + w) / a_mb, 2)!s:>22s}")

tot_b = sum(2 * fe[k] + wr.get(k, 0.0) for k in fe) * 1024.0 / steps
alg_b = sum(v for v in alg.values() if v)
print(f"whole step: {tot_b / 1e9:.2f} GB HBM-side (2 x FETCH + WRITE, bytes) vs {alg_b / 1e9:.2f} GB algorithmic as the engine books it "
      f"= {tot_b / alg_b if alg_b else float('nan'):.2f} x")

if json_out:
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                     "--no-extras; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch, family average",
           "config": line["config"].get("name"), "bytes_per_launch": {}}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from morphablediffusion_amd.lib import csrc_sha16
    out["csrc_sha16"] = csrc_sha16()  # bench.py refuses the file when the library sources have changed since
    for k in fe:
        n = len(nf[k])
        if n:
            out["bytes_per_launch"][k] = (2 * fe[k] + wr.get(k, 0.0)) * 1024.0 / n
    json.dump(out, open(json_out, "w"), indent=1)
