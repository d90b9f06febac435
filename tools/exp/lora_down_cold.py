"""vft_lora_down on COLD inputs (the in-step case: a layer's input was written long ago and is not in the 256 MB Infinity
Cache): cycle through enough input buffers that every launch reads from HBM.  python tools/exp/lora_down_cold.py"""
import sys

import torch

sys.path.insert(0, ".")
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
NAMES = {0: "auto", 3: "32 rows", 2: "64 rows"}
for M, K, r in [(98304, 640, 4), (98304, 2560, 4), (24576, 1280, 4), (24576, 5120, 4), (26160, 3072, 16), (26160, 8192, 16), (17408, 2304, 4), (17408, 6144, 4)]:
    nbuf = max(2, int(700e6 / (M * K * 2)) + 1)
    xs = [bf(M, K) for _ in range(nbuf)]
    A = bf(r, K)
    line = f"M={M:6d} K={K:5d} r={r:2d} ({nbuf} buffers):"
    ref = None
    for var in (0, 3, 2):
        _C.lib().vft_lora_down_variant(var)
        for i in range(nbuf):
            kernels.lora_down(xs[i], A, 0.25)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3 * nbuf
        e0.record()
        for i in range(n):
            out = kernels.lora_down(xs[i % nbuf], A, 0.25)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        o = kernels.lora_down(xs[0], A, 0.25).float()
        if ref is None:
            ref = o
        d = float((o - ref).abs().max() / ref.abs().max())
        line += f"  {NAMES[var]} {t * 1e6:6.1f} us {2.0 * M * K / t / 1e9:5.0f} GB/s (d {d:.0e})"
    _C.lib().vft_lora_down_variant(0)
    print(line, flush=True)
    del xs
