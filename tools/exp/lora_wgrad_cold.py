"""LoRA weight-gradient partial + reduce kernels on COLD inputs, per workgroup target (vft_lora_wgrad_target).
   python tools/exp/lora_wgrad_cold.py"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
raw = ctypes.CDLL(_C.LIB_PATH)
for M, K, N, r in [(98304, 640, 640, 4), (98304, 640, 5120, 4), (98304, 2560, 640, 4), (24576, 1280, 1280, 4), (24576, 1280, 10240, 4),
                   (24576, 5120, 1280, 4), (26160, 3072, 3072, 16), (26160, 3072, 8192, 16), (26160, 8192, 3072, 16), (17408, 2304, 3840, 4)]:
    nbuf = max(2, int(700e6 / (M * (K + N) * 2)) + 1)
    xs, dys = [bf(M, K) for _ in range(nbuf)], [bf(M, N) for _ in range(nbuf)]
    u, tt = bf(M, r), bf(M, r)
    A, B = bf(r, K).requires_grad_(), bf(N, r).requires_grad_()
    line = f"M={M:6d} K={K:5d} N={N:5d} r={r:2d} ({nbuf} buffers):"
    for tgt in (0, 512, 1024, 2048, 4096):
        raw.vft_lora_wgrad_target(tgt)
        for i in range(nbuf):
            kernels._lora_wgrad_into(xs[i], u, dys[i], tt, [(A, B)], r, r, 1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 3 * nbuf
        e0.record()
        for i in range(n):
            kernels._lora_wgrad_into(xs[i % nbuf], u, dys[i % nbuf], tt, [(A, B)], r, r, 1.0)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / n * 1e-3
        line += f"  target {tgt if tgt else 'auto'}: {t * 1e6:6.1f} us {2.0 * M * (K + N) / t / 1e9:5.0f} GB/s"
    raw.vft_lora_wgrad_target(0)
    print(line, flush=True)
    del xs, dys
