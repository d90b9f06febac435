"""PMC driver: SDXL self-attention backward at head_dim 64, one-pass kernel (variant 0) and the dq + dkdv pair (variant 3).
cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
   SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d <out> -- python3 tools/exp/pmc_attn64.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
shapes = [(24, 10, 4096, 64), (24, 20, 1024, 64)] if len(sys.argv) < 2 else [tuple(int(x) for x in sys.argv[1].split(","))]
for B, H, S, D in shapes:
    q, k, v = (bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(3))
    do = bf(B, S, H, D).permute(0, 2, 1, 3)
    for var in tuple(int(x) for x in os.environ.get("VFT_PMC_VARIANTS", "0,3").split(",")):
        _C.lib().vft_attn_variant(var)
        for _ in range(2):
            kernels.flash_attention(q, k, v).backward(do)
    _C.lib().vft_attn_variant(0)
torch.cuda.synchronize()
