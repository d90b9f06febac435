"""A/B timing of vft_conv3x3_fwd / vft_linear_fwd for ONE library build (run once per build, same box, same process order):
   python tools/exp/lin8_ab.py <path to libvft_hip.so> [conv|lin|all]
Loads the library directly (no package import), so an older build with a different symbol set can be timed too."""
import ctypes
import sys

import torch

lib = ctypes.CDLL(sys.argv[1])
what = sys.argv[2] if len(sys.argv) > 2 else "all"
vp, ci, c64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.vft_conv3x3_fwd.argtypes = [vp] * 4 + [ci] * 5 + [vp]
lib.vft_linear_fwd.argtypes = [vp] * 7 + [ci] * 4 + [c64] * 4 + [vp]
dev = torch.device("cuda:0")
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


st = torch.cuda.current_stream().cuda_stream
tot = 0.0
if what in ("conv", "all"):
    B = 24
    for Cin, Cout, H in [(320, 320, 128), (640, 320, 128), (960, 320, 128), (320, 640, 64), (640, 640, 64), (1280, 640, 64), (1920, 640, 64),
                         (960, 640, 64), (640, 1280, 32), (1280, 1280, 32), (2560, 1280, 32), (1920, 1280, 32)]:
        x, w, y, b = bf(B, H, H, Cin), bf(Cout, 3, 3, Cin) * 0.02, torch.empty(B, H, H, Cout, device=dev, dtype=torch.bfloat16), bf(Cout)
        us = timeit(lambda: lib.vft_conv3x3_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), b.data_ptr(), B, H, H, Cin, Cout, st))
        tot += us
        print(f"conv {Cin:4d}->{Cout:4d} @{H:3d}^2  {us:8.1f} us  {18.0 * B * H * H * Cin * Cout / us / 1e6:7.1f} TF/s")
if what in ("lin", "all"):
    for M, N, K in [(98304, 640, 640), (98304, 1920, 640), (98304, 5120, 640), (98304, 640, 2560), (24576, 1280, 1280), (24576, 3840, 1280),
                    (24576, 10240, 1280), (24576, 1280, 5120), (26160, 3072, 3072), (26160, 8192, 3072), (26160, 3072, 8192), (17408, 3840, 2304)]:
        x, w, y = bf(M, K), bf(N, K) * 0.02, torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: lib.vft_linear_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), None, None, None, None, M, N, K, 0, K, K, N, 0, st))
        tot += us
        print(f"lin M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
print(f"total {tot:.1f} us")
