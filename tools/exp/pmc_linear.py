"""Experiment: PMC counters of the LDS-DMA GEMM kernel at two shapes (run under rocprofv3 --pmc ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from vision_ft_amd import kernels

dev = torch.device("cuda")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)
for (M, N, K) in [(16384, 10240, 1280), (65536, 640, 2560)]:
    x, w, b = bf(M, K), bf(N, K) * 0.05, bf(N)
    for _ in range(3):
        kernels.linear_fused(x, w, b)
torch.cuda.synchronize()
