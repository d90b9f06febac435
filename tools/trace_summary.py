"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time of the LAST training step (between
the last two fused-optimizer launches).  python tools/trace_summary.py <dir-or-csv> [topN]"""
import collections
import csv
import glob
import re
import sys


def main():
    src = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    f = src if src.endswith(".csv") else glob.glob(src + "/**/*_kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    opt = [i for i, r in enumerate(rows) if "FusedOptimizer" in r["Kernel_Name"] or "multi_tensor_apply" in r["Kernel_Name"]]
    bounds = []
    for i in opt:
        if not bounds or i - bounds[-1] > 50:
            bounds.append(i)
        else:
            bounds[-1] = i
    a, b = bounds[-2] + 1, bounds[-1] + 1
    step = rows[a:b]
    wall = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e6
    cat, calls = collections.Counter(), collections.Counter()
    for r in step:
        n = r["Kernel_Name"]
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        m = re.search(r"MT\d+x\d+x\d+", n)
        n = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", n)
        key = n[:56] + (" " + m.group(0) if m else "")
        cat[key] += d
        calls[key] += 1
    tot = sum(cat.values())
    print(f"last step: wall {wall:.2f} ms, kernel sum {tot/1e6:.2f} ms, {len(step)} launches")

    def cls(n):
        if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
            return "gemm(hipblaslt)"
        if "transpose_kernel" in n:
            return "other"  # (vft_transpose_bf16: our own copy kernel, not a library convolution helper)
        if "igemm" in n or "Im2d2Col" in n or "Col2Im" in n or "SubTensorOp" in n or "transpose" in n.lower() or "ck::" in n or "naive_conv" in n:
            return "conv(miopen/ck)"
        if "wg8_" in n:
            return "weight grad(ours, wg8)"
        if "lin8_kernel" in n:
            import re
            m = re.search(r"C8<\d+, \d+, \d+, (true|false)(?:, (\d))?>, (true|false)>", n)  # C8<BM, BN, WAVES_M, CONV[, STRIDE]>, NF4
            if m:
                if m.group(3) == "true":
                    return "NF4 gemm(ours, lin8: dequantised once per launch)"
                return "conv3x3(ours, lin8)" if m.group(1) == "true" else "linear gemm(ours, lin8)"
            if re.search(r"false>\s*>?, true>", n) or "ELb0EEELb1E" in n:  # (mangled / older spellings)
                return "NF4 gemm(ours, lin8: dequantised once per launch)"
            return "conv3x3(ours, lin8)" if "ELb1EE" in n or ", true>" in n else "linear gemm(ours, lin8)"
        if "linear_kernel" in n:
            return "conv3x3/linear(ours, gen 1)"
        if "attn_" in n:
            return "attention(ours)"
        if "geglu" in n:
            return "norm/act(ours)"
        if "lora_" in n:
            return "lora(ours)"
        if "gemm_nt" in n or "nf4" in n:
            return "gemm/nf4(ours)"
        if any(k in n for k in ("norm_", "gn_", "geglu", "swiglu", "gate_res", "qk_norm")):
            return "norm/act(ours)"
        if "elementwise" in n or "Elementwise" in n:
            return "elementwise(aten)"
        return "other"

    agg = collections.Counter()
    for k, v in cat.items():
        agg[cls(k)] += v
    for k, v in agg.most_common():
        print(f"  {k:24s} {v/1e6:8.2f} ms {100*v/tot:5.1f}%")
    print("name,calls,total_ms,avg_us")
    for k, v in cat.most_common(top):
        print(f"{k},{calls[k]},{v/1e6:.3f},{v/calls[k]/1e3:.1f}")


if __name__ == "__main__":
    main()
