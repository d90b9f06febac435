"""Time vft_linear_fwd (lin8) on a few shapes; VFT_LIN8_NT_MB selects the non-temporal-store threshold.
    VFT_LIN8_NT_MB=-1 python tools/exp/lin_nt.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
for (M, N, K) in [(98304, 5120, 640), (98304, 1920, 640), (98304, 640, 640), (24576, 10240, 1280), (24576, 1280, 5120), (24576, 1280, 1280)]:
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = torch.randn(N, K, device=dev).to(torch.bfloat16)
    for _ in range(3):
        y = kernels.linear_fused(x, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = kernels.linear_fused(x, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"M={M} N={N} K={K}: {us:8.1f} us {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)
