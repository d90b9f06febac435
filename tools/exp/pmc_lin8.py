"""PMC driver: lin8 at SDXL / AuraFlow shapes (forward with bias + rank-4/16 extension), a few launches each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
for (M, N, K, r) in [(24576, 1280, 1280, 4), (24576, 10240, 1280, 4), (24576, 1280, 5120, 4), (98304, 640, 640, 4), (98304, 5120, 640, 4), (26160, 3072, 3072, 16), (26160, 8192, 3072, 16)]:
    x, w, b = bf(M, K), bf(N, K) * 0.05, bf(N)
    t, B2 = bf(M, r), bf(N, r)
    for _ in range(3):
        kernels.linear_fused(x, w, b, ext_x=t, ext_w=B2)
    torch.cuda.synchronize()
    del x, w, b, t, B2
