"""Lumina2 self-attention (head_dim 96, GQA 24 / 8 heads, 4352 tokens) backward: read-ahead kernels (variant 0) against the
round-1 kernels (variant 1), same process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
B, Hq, Hkv, S, D = 4, 24, 8, 4352, 96
q = bf(B, S, Hq, D).permute(0, 2, 1, 3).requires_grad_(True)
k = bf(B, S, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True)
v = bf(B, S, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True)
do = bf(B, S, Hq, D).permute(0, 2, 1, 3)
res = {}
for variant in (1, 0, 1, 0):
    _C.lib().vft_attn_variant(variant)
    for _ in range(2):
        q.grad = k.grad = v.grad = None
        kernels.flash_attention(q, k, v).backward(do)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = kernels.flash_attention(q, k, v)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        q.grad = k.grad = v.grad = None
        out.backward(do, retain_graph=True)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    fl = 10.0 * B * Hq * S * S * D
    print(f"variant {variant}: backward {ms * 1e3:8.1f} us  {fl / ms / 1e9:6.0f} TF/s (algorithmic 10 B H S^2 D)", flush=True)
    res[variant] = (q.grad.clone(), k.grad.clone(), v.grad.clone())
for i, n in enumerate("qkv"):
    d = (res[0][i].float() - res[1][i].float()).norm() / res[1][i].float().norm()
    print(f"d{n}: variant 0 vs 1 rel diff {d.item():.2e}")
_C.lib().vft_attn_variant(0)
