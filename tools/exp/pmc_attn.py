import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "attn" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][28:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, "launches", len(v), "avg", sum(v) / len(v) * 1024 * 2 / 1e6, "MB (x2 corrected)")
