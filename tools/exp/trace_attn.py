import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "attn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    if i % 5 == 4:
        print(r["Kernel_Name"][:48], r.get("Grid_Size_X", r.get("Grid_Size", "")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
