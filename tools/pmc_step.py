"""Whole-step MFMA-busy from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over bench.py (counter CSV):
  python tools/pmc_step.py <counter_collection.csv>
MFMA busy of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); per family and for all launches
together (= the step's MFMA-busy while kernels run; the denominators are kernel-active cycles, launch gaps excluded)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
val = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    val[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
fam = collections.defaultdict(lambda: [0.0, 0.0, 0])
for (_, name), c in val.items():
    m = re.search(r"(gemm_dma_kernel|conv3_dma_kernel|igemm_kernel|conv3x_kernel|rowchain_kernel)<[^>]*>|conv3x_kernelILi\d+|rowchain_kernelILi\d+|attn_kernel|depth_attn|gn_|layernorm|splitk_reduce|sparse_conv", name)
    k = m.group(0) if m else "other"
    fam[k][0] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    fam[k][1] += c.get("GRBM_GUI_ACTIVE", 0.0)
    fam[k][2] += 1
tb = sum(v[0] for v in fam.values())
ta = sum(v[1] for v in fam.values())
print(f"all {sum(v[2] for v in fam.values())} dispatches: MFMA busy = {tb:.3e} / (1024 x {ta / 8:.3e}) = {100 * tb / (1024 * ta / 8):.1f} % of the kernel-active cycles")
for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    if v[1] > 0:
        print(f"  {k:40s} {v[2]:6d} dispatches  active share {100 * v[1] / ta:5.1f} %  MFMA busy {100 * v[0] / (1024 * v[1] / 8):5.1f} %")
