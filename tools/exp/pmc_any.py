import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][sys.argv[3] == "full" and 0 or 28:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k[0][:48], k[1], "avg", sum(v) / len(v))
