"""The three GEMMs of a TRAINABLE projection (BASELINE config 2, SDXL full fine-tune, B = 24) on our kernels against the
table-tuned library: forward (lin8 vs F.linear), data gradient (lin8 on a transposed weight copy vs dy @ W), weight gradient
(wg8 vs dy^T @ x).  One process, HIP events, median of 9."""
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels, tuning  # noqa: E402

dev = torch.device("cuda:0")
tuning.enable(tune=False)
# (tokens, out_features, in_features, count per step)
SHAPES = [(98304, 640, 640, 4 * 12), (98304, 5120, 640, 12), (98304, 640, 2560, 12),
          (24576, 1280, 1280, 4 * 58), (24576, 10240, 1280, 58), (24576, 1280, 5120, 58)]


def med(fn, reps=9):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return statistics.median(ts)


tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
for M, N, K, cnt in SHAPES:
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K**-0.5).to(torch.bfloat16)
    b = torch.randn(N, generator=g, device=dev).to(torch.bfloat16)
    dy = (torch.randn(M, N, generator=g, device=dev) * 0.01).to(torch.bfloat16)
    wt = w.t().contiguous()
    out = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    fl = 2.0 * M * N * K
    r = {}
    r["fwd"] = (med(lambda: kernels.linear_fused(x, w, b)), med(lambda: F.linear(x, w, b)))
    r["dgrad"] = (med(lambda: kernels.linear_fused(dy, wt, None)), med(lambda: dy @ w))
    r["wgrad"] = (med(lambda: kernels.linear_wgrad(dy, x, out=out)), med(lambda: torch.mm(dy.t(), x)))
    line = f"M={M:6d} N={N:5d} K={K:5d} x{cnt:3d}:"
    for k_, (ours, lib) in r.items():
        line += f"  {k_} ours {ours * 1e3:6.0f} us {fl / ours / 1e9:5.0f} TF/s | lib {lib * 1e3:6.0f} us {fl / lib / 1e9:5.0f}"
        tot[k_][0] += ours * cnt
        tot[k_][1] += lib * cnt
    print(line, flush=True)
tr = med(lambda: w.t().contiguous())
print(f"transpose of a {N}x{K} weight: {tr * 1e3:.0f} us")
for k_, (o, l) in tot.items():
    print(f"per step, {k_}: ours {o:.1f} ms, library {l:.1f} ms")
