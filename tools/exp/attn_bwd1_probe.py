"""Where does attn_bwd1_kernel's time go?  Timing-only runs with parts of the kernel disabled (VFT_ATTN_DEBUG bits, WRONG
results; the library must be built with the probe compiled in: `make -C vision_ft_amd/csrc clean && make -C vision_ft_amd/csrc
EXTRA=-DVFT_ATTN_PROBE`, the shipped build ignores the variable): 1 = no running-dQ traffic (scratch plane neither read nor written), 2 = dQ product cut to one of its eight MFMA
steps, 8 = no Q / dO / statistics staging traffic (empty descriptors: the DMAs deliver zeros).  One process, alternating.

    python tools/exp/attn_bwd1_probe.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")


def bf(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)


for (B, H, S, D) in ((24, 10, 4096, 64), (24, 20, 1024, 64)):
    q, k, v = (bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(3))
    do = bf(B, S, H, D).permute(0, 2, 1, 3)
    times = {}
    for i in range(9):
        for dbg in tuple(int(v) for v in os.environ.get("VFT_PROBE_SET", "0,1,2,8,9,11").split(",")):
            os.environ["VFT_ATTN_DEBUG"] = str(dbg)
            o = kernels.flash_attention(q, k, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            o.backward(do)
            e1.record()
            torch.cuda.synchronize()
            q.grad = k.grad = v.grad = None
            if i >= 2:
                times.setdefault(dbg, []).append(e0.elapsed_time(e1))
    os.environ["VFT_ATTN_DEBUG"] = "0"
    print(f"B{B} H{H} S{S} D{D}: " + "  ".join(f"dbg{d}: {statistics.median(t):.3f} ms" for d, t in times.items()), flush=True)
