"""Experiment: MIOpen 3x3 conv fwd + dgrad time, NCHW vs channels-last storage (bf16, frozen weight)."""
import sys
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda")
cases = [(16, 320, 320, 128), (16, 640, 640, 64), (16, 1280, 1280, 32), (16, 2560, 1280, 32), (16, 960, 320, 128), (16, 320, 640, 64)]
for cl in (False, True):
    for (B, Ci, Co, H) in cases:
        x = torch.randn(B, Ci, H, H, device=dev, dtype=torch.bfloat16).requires_grad_(True)
        w = torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02
        b = torch.zeros(Co, device=dev, dtype=torch.bfloat16)
        if cl:
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            w = w.contiguous(memory_format=torch.channels_last)
        y = F.conv2d(x, w, b, padding=1)
        gy = torch.randn_like(y)
        for _ in range(3):
            y = F.conv2d(x, w, b, padding=1)
            y.backward(gy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            y = F.conv2d(x, w, b, padding=1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n):
            y = F.conv2d(x, w, b, padding=1)
            y.backward(gy)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        fl = 2.0 * B * H * H * Ci * Co * 9
        tf = (t1 - t0) / n
        tb = (t2 - t1) / n - tf
        print(f"cl={cl} B={B} Ci={Ci} Co={Co} H={H}: fwd {tf*1e3:7.3f} ms {fl/tf/1e12:6.0f} TF/s | dgrad {tb*1e3:7.3f} ms {fl/tb/1e12:6.0f} TF/s"
              f" | y cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)

print("== ours (vft_conv3x3_fwd) ==")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels
for (B, Ci, Co, H) in cases:
    x = torch.randn(B, Ci, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, device=dev, dtype=torch.bfloat16) * 0.02).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(Co, device=dev, dtype=torch.bfloat16)
    y = kernels.conv3x3(x, w, b)
    gy = torch.randn_like(y)
    for _ in range(3):
        y = kernels.conv3x3(x, w, b)
        y.backward(gy)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        y = kernels.conv3x3(x, w, b)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n):
        y = kernels.conv3x3(x, w, b)
        y.backward(gy)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    fl = 2.0 * B * H * H * Ci * Co * 9
    tf = (t1 - t0) / n
    tb = (t2 - t1) / n - tf
    print(f"ours B={B} Ci={Ci} Co={Co} H={H}: fwd {tf*1e3:7.3f} ms {fl/tf/1e12:6.0f} TF/s | dgrad {tb*1e3:7.3f} ms {fl/tb/1e12:6.0f} TF/s", flush=True)
