"""Per-kernel averages of a rocprofv3 --pmc run: python tools/exp/pmc_table.py <dir> <name filter>; derived shares of
SQ_BUSY_CU_CYCLES / SQ_WAVE_CYCLES as the MI355X guide defines them."""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
        acc[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), c in sorted(acc.items()):
    a = {k: sum(v) / len(v) for k, v in c.items()}
    line = f"{name:44s} grid {grid:>9s} n={len(next(iter(c.values())))}"
    if "SQ_BUSY_CU_CYCLES" in a and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
        line += f"  mfma_busy/cu_busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / a['SQ_BUSY_CU_CYCLES'] / 4:.3f}"  # 4 SIMDs per CU
    if "SQ_WAVE_CYCLES" in a:
        for k in sorted(k for k in a if k.startswith(("SQ_WAIT", "SQ_ACTIVE_INST", "SQ_INST_CYCLES"))):
            if k in a:
                line += f"  {k[3:]} {a[k] / a['SQ_WAVE_CYCLES']:.3f}"
    if "SQ_LDS_IDX_ACTIVE" in a and "SQ_LDS_BANK_CONFLICT" in a:
        line += f"  lds_conflict/active {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.3f}"
    print(line)
    print("     " + "  ".join(f"{k}={v:.3e}" for k, v in sorted(a.items())))
