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


def conv():
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    for B, Cin, Cout, H in [(16, 320, 320, 128), (16, 640, 640, 64), (16, 1280, 1280, 32), (16, 960, 640, 64), (16, 2560, 1280, 32),
                            (16, 640, 320, 128)]:
        x = torch.randn(B, Cin, H, H, generator=g, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = (torch.randn(B, Cout, H, H, generator=g, device=dev) * 0.01).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 3, 3, generator=g, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        out = torch.empty(Cout, 9 * Cin, dtype=torch.bfloat16, device=dev)
        t_ours = timeit(lambda: kernels.conv3x3_wgrad(dy, x, out=out), it=10)
        lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])
        t_lib = timeit(lib, it=10)
        ref = lib()[1].permute(0, 2, 3, 1).reshape(Cout, -1).float()
        err = ((out.float() - ref).norm() / ref.norm()).item()
        fl = 2.0 * B * H * H * Cout * 9 * Cin
        print(f"conv B={B} {Cin}->{Cout} @{H}  ours {t_ours * 1e3:8.1f} us {fl / t_ours / 1e9:7.0f} TF/s   "
              f"MIOpen {t_lib * 1e3:8.1f} us {fl / t_lib / 1e9:7.0f} TF/s   rel diff {err:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "conv":
        conv()
    else:
        main()
