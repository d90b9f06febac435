"""A few launches of the weight-gradient kernel at two full fine-tune shapes (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
dy, x = bf(16384, 1280) * 0.01, bf(16384, 1280)
for _ in range(3):
    kernels.linear_wgrad(dy, x)
xq = bf(16, 1280, 32, 32).contiguous(memory_format=torch.channels_last)
gq = (bf(16, 1280, 32, 32) * 0.01).contiguous(memory_format=torch.channels_last)
for _ in range(3):
    kernels.conv3x3_wgrad(gq, xq)
torch.cuda.synchronize()
