"""PMC driver: AuraFlow-shaped attention (B=4, H=12, S=4360, D=256), forward with the 4-wave kernel (variant 1) and
the 8-wave kernel (variant 0), then one backward.  rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -- python3 tools/exp/pmc_attn256.py"""
import sys

import torch

sys.path.insert(0, ".")
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
B, H, S, D = 4, 12, 4360, 256
q = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
k = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
v = bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True)
do = bf(B, S, H, D).permute(0, 2, 1, 3)
for var in (1, 0):
    _C.lib().vft_attn_variant(var)
    for _ in range(3):
        o = kernels.flash_attention(q, k, v)
    o.backward(do)
torch.cuda.synchronize()
