"""Measure the lin8 tile configuration (256x256 / 256x128 / 128x128) per problem size and write
vision_ft_amd/tuning_tables/gfx950_lin8_tiles.csv (run on an MI355X).

    python tools/tune_lin8.py [--workloads sdxl_lora,aura_lora,...] [--iters 8]

Step 1 runs one eager forward+backward of each bench workload with `kernels.SHAPE_LOG` on, which records every
(M, N, K) the fused GEMM and every (B, H, W, Cin, Cout) the implicit-GEMM convolution is called with (forward and data
gradient).  Step 2 times each problem under `vft_linear_variant(1|2|3)` on random operands (HIP events, the median of
`iters` launches after a warm-up) and keeps the fastest; rows where the analytic choice is already the best are written
too, so that the table documents the measurement.  The library reads the table at load time (`_C.load_tile_table`)."""
import argparse
import gc
import os
import sys

os.environ["VFT_NO_TILE_TABLE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from vision_ft_amd import _C, kernels  # noqa: E402

BATCH = {"sdxl_lora": 24, "sdxl_full": 24, "sdxl_buckets": 16, "aura_lora": 6, "lumina_lora": 4, "lumina_qlora": 4}  # bench.py defaults
BUCKETS = [(640, 1536), (768, 1280), (768, 1408), (896, 1152), (1024, 1024), (1152, 896), (1280, 768), (1408, 768), (1536, 640)]


def record(workload: str, dev) -> set:
    model = bench.build(workload, dev)
    shapes: set = set()
    for wh in (BUCKETS if workload == "sdxl_buckets" else [(1024, 1024)]):
        step = bench.make_step(workload, model, dev, BATCH[workload], seed=1, wh=wh)
        kernels.SHAPE_LOG = shapes
        step().backward()
        kernels.SHAPE_LOG = None
        torch.cuda.synchronize()
    del model, step
    gc.collect()
    torch.cuda.empty_cache()
    return shapes


def time_ms(fn, iters: int) -> float:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="sdxl_lora,aura_lora,lumina_lora,sdxl_full,sdxl_buckets")
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--out", default=_C.TILE_TABLE)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch for every listed workload (default: bench.py's)")
    ap.add_argument("--merge", action="store_true", help="keep the rows of an existing table for problems not measured now")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    shapes: set = set()
    for w in args.workloads.split(","):
        if args.batch:
            BATCH[w] = args.batch
        got = record(w, dev)
        print(f"{w}: {len(got)} problem sizes", flush=True)
        shapes |= got
    lib = _C.lib()
    rows = []
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
    for sh in sorted(shapes):
        if sh[0] == "lin":
            _, M, N, K = sh
            if K % 64 or N % 8:
                continue
            x, w = rn(M, K), rn(N, K)
            call = lambda: kernels.linear_fused(x, w)  # noqa: E731
            flop, key = 2.0 * M * N * K, (M, N, K, 0)
        else:
            _, B, H, W, Cin, Cout = sh
            if Cin % 64:
                continue
            x = rn(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
            w = rn(Cout, Cin, 3, 3).contiguous(memory_format=torch.channels_last)
            call = lambda: kernels._conv3x3_call(x, w, None, B, H, W, Cin, Cout, "conv3x3_fwd")  # noqa: E731
            flop, key = 18.0 * B * H * W * Cin * Cout, (B * H * W, Cout, 9 * Cin, 1)
        tf = []
        cfgs = (1, 2, 3, 11, 12, 13, 0)  # 11..13: whole rounds on tile 1..3, the remaining rows as a second launch
        for cfg in cfgs:
            lib.vft_linear_variant(cfg)
            tf.append(flop / time_ms(call, args.iters) / 1e9)
        lib.vft_linear_variant(0)
        best = cfgs[max(range(6), key=lambda i: tf[i] * (1.0 if i < 3 else 0.985))]  # a split must win by 1.5 %
        rows.append((*key, best, *tf))
        print(f"{sh}: 256x256 {tf[0]:7.1f}  256x128 {tf[1]:7.1f}  128x128 {tf[2]:7.1f}  split {tf[3]:7.1f} {tf[4]:7.1f} {tf[5]:7.1f}"
              f"  analytic {tf[6]:7.1f} TF/s -> cfg {best}", flush=True)
        del x, w
    seen, uniq = set(), []
    for r in sorted(rows, key=lambda r: (r[3], r[0], r[1], r[2])):  # conv problems of transposed buckets share (M, N, K)
        if r[:4] not in seen:
            seen.add(r[:4])
            uniq.append(r)
    rows = uniq
    if args.merge and os.path.exists(args.out):
        for line in open(args.out):
            if line[0].isdigit():
                v = line.strip().split(",")
                key = tuple(int(t_) for t_ in v[:4])
                if key not in seen:
                    seen.add(key)
                    rows.append((*key, int(v[4]), *[float(t_) for t_ in v[5:12]]))
        rows.sort(key=lambda r: (r[3], r[0], r[1], r[2]))
    with open(args.out, "w") as f:
        f.write("# lin8 tile per problem, measured on MI355X by tools/tune_lin8.py (TFLOP/s per tile, then the analytic choice's)\n")
        f.write("M,N,K,conv,cfg,tf_256x256,tf_256x128,tf_128x128,tf_split1,tf_split2,tf_split3,tf_analytic\n")
        for r in rows:
            f.write(",".join(str(v) if isinstance(v, int) else f"{v:.1f}" for v in r) + "\n")
    gain = sum(max(r[5:11]) / r[11] for r in rows) / max(len(rows), 1)
    print(f"wrote {len(rows)} rows to {args.out}; mean best/analytic = {gain:.3f}")


if __name__ == "__main__":
    main()
