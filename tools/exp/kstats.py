"""Print the top rows of a rocprofv3 --stats kernel_stats.csv found under a directory:  python tools/exp/kstats.py <dir> [rows]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:10.1f} us')
