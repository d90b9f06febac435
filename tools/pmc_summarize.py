"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as the MI355X guide prescribes)
of tools/pmc_run.py into profiles/<name>.csv.  FETCH_SIZE is in KB and on gfx950 counts 64 B per 128-B
request for wide streaming reads, hence the x2; WRITE_SIZE is in KB.
   python tools/pmc_summarize.py <fetch_dir> <write_dir> <out.csv>"""
import collections
import csv
import glob
import sys

ALG_MB = {  # algorithmic HBM bytes per launch of the shapes in tools/pmc_run.py
    "attn_fwd_kernel": 4 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_bwd_dq_kernel": 5 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_bwd_dkdv_kernel": 6 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "lora_rows_kernelILi4ELb1": (2 * 98304 * 640 + 4 * 98304 * 5120) / 1e6,
    "lora_rows_kernelILi4ELb0": (2 * 98304 * 5120 + 4 * 98304 * 640) / 1e6,
    "lora_wgrad_partial": 2 * 98304 * (640 + 5120) / 1e6,
    "linear_kernel": (2 * 24 * 128 * 128 * 640 + 18 * 320 * 320) / 1e6,
}


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for k in ALG_MB:
            if k in r["Kernel_Name"]:
                acc[k] += float(r["Counter_Value"])
                n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, n


def main():
    fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    with open(sys.argv[3], "w") as out:
        out.write("kernel,launches,FETCH_SIZE_raw_KB_per_launch,fetch_MB_corrected_x2,write_MB,traffic_MB,algorithmic_MB\n")
        for k in ALG_MB:
            if k not in fetch:
                continue
            fmb = fetch[k] * 1024 * 2 / 1e6
            wmb = write.get(k, 0.0) * 1024 / 1e6
            out.write(f"{k},{n[k]},{fetch[k]:.0f},{fmb:.1f},{wmb:.1f},{fmb + wmb:.1f},{ALG_MB[k]:.1f}\n")
    print(open(sys.argv[3]).read())


if __name__ == "__main__":
    main()
