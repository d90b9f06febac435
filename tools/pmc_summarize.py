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
    "attn_fwd3_kernel": 4 * 24 * 10 * 4096 * 64 * 2 / 1e6,  # round 4: head_dim 64 / 96 forward on the LDS-DMA operand path
    "attn_bwd_dq_kernel": 5 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_bwd_dkdv_kernel": 6 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_bwd_dq3_kernel": 5 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_bwd_dkdv3_kernel": 6 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    # round 3, one-pass backward: q, k, v, dO read + dq, dk, dv written (the running-dQ scratch and the Q / dO re-reads per
    # key block come on top: that is what this row is for); its row-constant pre-pass reads O and dO
    "attn_bwd1_kernel": 7 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "attn_delta_kernel": 2 * 24 * 10 * 4096 * 64 * 2 / 1e6,
    "lora_rows_kernelILi4ELb1": (2 * 98304 * 640 + 4 * 98304 * 5120) / 1e6,
    "lora_rows_kernelILi4ELb0": (2 * 98304 * 5120 + 4 * 98304 * 640) / 1e6,
    "lora_wgrad_partial": 2 * 98304 * (640 + 5120) / 1e6,
    "linear_kernel": (2 * 24 * 128 * 128 * 640 + 18 * 320 * 320) / 1e6,
    # round 2: the second-generation GEMM.  Its two plain instantiations are told apart by launch order in tools/pmc_run.py
    # (first shape 98304 x 5120 x 640, then 24576 x 10240 x 1280: both pick the 256x256 tile); conv mode = C8<..., true>
    "lin8_kernel": None,
    # weight-gradient kernel: linear mode = wg8_kernel<false> (ILb0E), convolution mode = <true>; their fp32 partial tiles
    # (write, then read by wg8_reduce_kernel) are on top of the algorithmic operand bytes
    "wg8_kernelILb0E": 2 * (16384 * 1280 * 2 + 1280 * 1280) / 1e6,
    "wg8_kernelILb1E": 2 * (2 * 16 * 64 * 64 * 640 + 9 * 640 * 640) / 1e6,
    "wg8_kernel<false>": 2 * (16384 * 1280 * 2 + 1280 * 1280) / 1e6,
    "wg8_kernel<true>": 2 * (2 * 16 * 64 * 64 * 640 + 9 * 640 * 640) / 1e6,
    "wg8_reduce_kernel": None,
}
LIN8_SHAPES = [("lin8_kernel", "M=98304 N=5120 K=640 (+bias + rank-4 extension)", 2 * (98304 * 640 + 5120 * 640 + 98304 * 5120 + 4 * (98304 + 5120)) / 1e6),
               ("lin8_kernel_b", "M=24576 N=10240 K=1280 (+bias + rank-4 extension)", 2 * (24576 * 1280 + 10240 * 1280 + 24576 * 10240 + 4 * (24576 + 10240)) / 1e6),
               ("lin8_conv", "conv3x3 320->320 @128x128 B=24", (2 * 24 * 128 * 128 * 640 + 18 * 320 * 320) / 1e6),
               # round 4: NF4 weight dequantised once inside the launch (x + y + packed codes + absmax; the bf16 image — written
               # once, read once per XCD through L2 — is what the measured traffic shows on top)
               ("lin8_nf4", "NF4 M=17408 N=3840 K=2304 (+ rank-4 extension)", (2 * 17408 * (2304 + 3840) + 3840 * 2304 // 2 + 3840 * 2304 // 16) / 1e6)]


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc, n = collections.Counter(), collections.Counter()
    lin8_seen = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        if "lin8_kernel" in name:  # plain launches: 3 of the first shape, then 3 of the second; conv launches: Lb1 in the template
            import re
            m = re.search(r"C8<\d+, \d+, \d+, (true|false)(?:, (\d))?>, (true|false)>", name)  # C8<BM, BN, WAVES_M, CONV[, STRIDE]>, NF4
            nf4 = (m.group(3) == "true") if m else bool(re.search(r"false>\s*>?, true>", name) or "ELb0EEELb1E" in name)
            if nf4:
                acc["lin8_nf4"] += float(r["Counter_Value"])
                n["lin8_nf4"] += 1
                continue
            conv = (m.group(1) == "true") if m else ("ELb1EE" in name or ", true>" in name)
            key = "lin8_conv" if conv else ("lin8_kernel" if lin8_seen < 3 else "lin8_kernel_b")
            lin8_seen += 0 if conv else 1
            acc[key] += float(r["Counter_Value"])
            n[key] += 1
            continue
        for k in ALG_MB:
            if k in name:
                acc[k] += float(r["Counter_Value"])
                n[k] += 1
    return {k: acc[k] / n[k] for k in acc}, n


def main():
    fetch, n = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    alg = {k: v for k, v in ALG_MB.items() if v is not None}
    shape = {}
    for k, sh, mb in LIN8_SHAPES:
        alg[k], shape[k] = mb, sh
    with open(sys.argv[3], "w") as out:
        out.write("kernel,launches,FETCH_SIZE_raw_KB_per_launch,fetch_MB_corrected_x2,write_MB,traffic_MB,algorithmic_MB,shape\n")
        for k in alg:
            if k not in fetch:
                continue
            fmb = fetch[k] * 1024 * 2 / 1e6
            wmb = write.get(k, 0.0) * 1024 / 1e6
            out.write(f"{k},{n[k]},{fetch[k]:.0f},{fmb:.1f},{wmb:.1f},{fmb + wmb:.1f},{alg[k]:.1f},{shape.get(k, 'see tools/pmc_run.py')}\n")
    print(open(sys.argv[3]).read())


if __name__ == "__main__":
    main()
