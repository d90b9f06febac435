"""Experiment: GPU time of the cross-attention backward (Sk=77) kernels at the SDXL bench shapes (run under rocprofv3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vision_ft_amd import kernels

dev = torch.device("cuda")
for (B, H, Sq, Sk) in [(16, 10, 4096, 77), (16, 20, 1024, 77), (16, 10, 4096, 4096), (16, 20, 1024, 1024)]:
    q = torch.randn(B, Sq, H, 64, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    k = torch.randn(B, Sk, H, 64, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    v = torch.randn(B, Sk, H, 64, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    for _ in range(5):
        o = kernels.flash_attention(q, k, v)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
