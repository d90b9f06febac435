"""vft_linear_wgrad against torch's (library) dyᵀ·x on the SDXL full fine-tune's weight-gradient shapes (B = 16)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import kernels  # noqa: E402

SHAPES = [(65536, 640, 640), (65536, 5120, 640), (65536, 640, 2560), (16384, 1280, 1280), (16384, 10240, 1280),
          (16384, 1280, 5120), (65536, 1920, 640), (16384, 3840, 1280), (262144, 320, 320)]


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for M, R, Cc in SHAPES:
        dy = (torch.randn(M, R, generator=g, device=dev) * 0.01).to(torch.bfloat16)
        x = torch.randn(M, Cc, generator=g, device=dev).to(torch.bfloat16)
        out = torch.empty(R, Cc, dtype=torch.bfloat16, device=dev)
        t_ours = timeit(lambda: kernels.linear_wgrad(dy, x, out=out))
        t_lib = timeit(lambda: torch.mm(dy.t(), x))
        fl = 2.0 * M * R * Cc
        err = ((out.float() - (dy.float().t() @ x.float())).norm() / (dy.float().t() @ x.float()).norm()).item()
        print(f"M={M:7d} R={R:6d} Cc={Cc:5d}  ours {t_ours * 1e3:8.1f} us {fl / t_ours / 1e9:7.0f} TF/s   "
              f"library {t_lib * 1e3:8.1f} us {fl / t_lib / 1e9:7.0f} TF/s   rel err {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
