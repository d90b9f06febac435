"""Experiment: per-kernel GPU time of attention fwd / dq / dkdv at the Lumina2 (D=96, GQA 24/8) and AuraFlow (D=256)
bench shapes (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vision_ft_amd import kernels

dev = torch.device("cuda")
for (B, Hq, Hkv, S, D) in [(4, 24, 8, 4352, 96), (4, 12, 12, 4360, 256)]:
    q = torch.randn(B, S, Hq, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    k = torch.randn(B, S, Hkv, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    v = torch.randn(B, S, Hkv, D, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3).requires_grad_(True)
    for _ in range(2):
        o = kernels.flash_attention(q, k, v)
        o.backward(torch.ones_like(o))
    torch.cuda.synchronize()
