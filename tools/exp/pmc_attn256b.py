"""PMC driver: AuraFlow-shaped attention backward (B=6, H=12, S=4360, D=256) twice."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
B, H, S, D = 6, 12, 4360, 256
q, k, v = (bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(3))
do = bf(B, S, H, D).permute(0, 2, 1, 3)
for _ in range(2):
    kernels.flash_attention(q, k, v).backward(do)
torch.cuda.synchronize()
