"""Average kernel duration per (kernel name, grid size) from a rocprofv3 --kernel-trace csv:  python tools/exp/ktrace_by_grid.py <dir> <filter>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        acc[(name, r.get("Grid_Size", r.get("Grid_Size_X", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items()):
    v.sort()
    print(f"{name:48s} grid {grid:>9s} n={len(v):3d} median {v[len(v) // 2]:8.1f} us  min {v[0]:8.1f}  p10 {v[len(v) // 10]:8.1f}")
