"""Cycles per segment of attn_bwd1_kernel's tile loop (diagnostic build: -DVFT_ATTN_PROBE -DVFT_ATTN_STAMP, library path in
VFT_LIB_PATH; the shipped build has no stamps).  Shares only — the stamps' own waits forbid overlaps the real kernel has.

    VFT_LIB_PATH=$PWD/tools/exp/libvft_stamp.so python tools/exp/attn_bwd1_stamps.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["R+M1+T (0)", "V+W (0)", "M2 (0)", "M1+T (1)", "V+W (1)", "M2 (1)", "B1", "stage", "dQ product", "xchg+B2", "RMW", "-"]


def bf(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)


lib = _C.lib()
for (B, H, S, D) in ((24, 10, 4096, 64), (24, 20, 1024, 64)):
    q, k, v = (bf(B, S, H, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(3))
    do = bf(B, S, H, D).permute(0, 2, 1, 3)
    for dbg in (0, 9):
        os.environ["VFT_ATTN_DEBUG"] = str(dbg)
        for _ in range(3):
            o = kernels.flash_attention(q, k, v)
            o.backward(do)
            q.grad = k.grad = v.grad = None
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (8 * 16))()
        rc = lib.vft_attn_probe_stamps(buf)
        assert rc == 0, rc
        tiles = (S // 256) * (S // 64)
        print(f"B{B} H{H} S{S} dbg{dbg}: cycles per tile and segment, waves 0-7 of workgroup 0 ({tiles} tiles)")
        for kseg in range(11):
            row = [buf[w * 16 + kseg] / tiles for w in range(8)]
            print(f"  {NAMES[kseg]:<14}" + " ".join(f"{x:7.0f}" for x in row))
        tot = [sum(buf[w * 16 + kseg] for kseg in range(11)) / tiles for w in range(8)]
        print(f"  {'sum':<14}" + " ".join(f"{x:7.0f}" for x in tot), flush=True)
