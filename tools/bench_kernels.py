"""Micro-benchmarks of the hand-written kernels at the headline shapes (run on the GPU box):
   python tools/bench_kernels.py [lora|attn|norm|all]
Prints achieved GB/s (HBM-bound kernels, algorithmic bytes) or TFLOP/s (attention)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from vision_ft_amd import kernels  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e-3


def bf(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)


def lora():
    print("== LoRA (bytes = algorithmic, see csrc/lora.hip) ==")
    for M, K, N, r in [(4096, 1280, 1280, 4), (4096, 1280, 10240, 4), (4096, 5120, 1280, 4), (16384, 640, 640, 4),
                       (16384, 640, 5120, 4), (16384, 2560, 640, 4), (308, 2048, 1280, 4), (8720, 3072, 3072, 16),
                       (8720, 3072, 8192, 16)]:
        x, y, dy, dx = bf(M, K), bf(M, N), bf(M, N), bf(M, K)
        A, B = bf(r, K), bf(N, r)
        t, Bt = kernels._lora_fwd_(x, y, A, B, 0.25)
        u = kernels._lora_bwd_dx_(dy, dx, Bt, A, 0.25)
        tf = timeit(lambda: kernels._lora_fwd_(x, y, A, B, 0.25))
        tb = timeit(lambda: kernels._lora_bwd_dx_(dy, dx, Bt, A, 0.25))
        tw = timeit(lambda: kernels._lora_wgrad2(x, u, dy, t, A, B, 0.25))
        bfwd, bbwd, bwg = 2 * M * K + 4 * M * N, 2 * M * N + 4 * M * K, 2 * M * (K + N)
        print(f"M={M:6d} K={K:5d} N={N:5d} r={r:2d} | fwd {tf*1e6:7.1f} us {bfwd/tf/1e9:7.0f} GB/s | bwd_dx {tb*1e6:7.1f} us "
              f"{bbwd/tb/1e9:7.0f} GB/s | wgrad2 {tw*1e6:7.1f} us {bwg/tw/1e9:7.0f} GB/s")


def lora_down():
    print("== vft_lora_down (t = x A^T) and vft_lora_wgrad2g at the bench shapes; GB/s of algorithmic bytes ==")
    shapes = [(98304, 640, 4), (98304, 640, 16), (98304, 1920, 16), (98304, 2560, 4), (98304, 5120, 4), (24576, 1280, 4),
              (24576, 1280, 16), (24576, 3840, 16), (24576, 5120, 4), (24576, 10240, 4), (17440, 3072, 16), (17440, 8192, 16)]
    from vision_ft_amd import _C
    for M, K, r in shapes:
        x, A = bf(M, K), bf(r, K)
        line = f"lora_down M={M:6d} K={K:5d} r={r:2d}:"
        outs = []
        for var in (1, 0, 2):
            _C.lib().vft_lora_down_variant(var)
            outs.append(kernels.lora_down(x, A, 0.25))
            t = timeit(lambda: kernels.lora_down(x, A, 0.25))
            line += f"  {(32, 16, 64)[var]} rows {t * 1e6:7.1f} us {2.0 * M * K / t / 1e9:6.0f} GB/s"
        _C.lib().vft_lora_down_variant(0)
        print(line, " equal" if torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) else " DIFFERENT", flush=True)
    for M, K, N, r in [(98304, 640, 640, 4), (98304, 640, 1920, 16), (98304, 640, 5120, 4), (98304, 2560, 640, 4),
                       (24576, 1280, 1280, 4), (24576, 1280, 3840, 16), (24576, 1280, 10240, 4), (24576, 5120, 1280, 4),
                       (17440, 3072, 3072, 16), (17440, 3072, 8192, 16), (17440, 8192, 3072, 16)]:
        x, dy, u, tt = bf(M, K), bf(M, N), bf(M, r), bf(M, r)
        A, B = bf(r, K).requires_grad_(), bf(N, r).requires_grad_()
        import ctypes
        raw = ctypes.CDLL(_C.LIB_PATH)
        line = f"wgrad2g  M={M:6d} K={K:5d} N={N:5d} r={r:2d}:"
        for tgt in (1024, 2048, 4096):
            raw.vft_lora_wgrad_target(tgt)
            t = timeit(lambda: kernels._lora_wgrad_into(x, u, dy, tt, [(A, B)], r, r, 1.0))
            line += f"  target {tgt}: {t * 1e6:7.1f} us {2.0 * M * (K + N) / t / 1e9:6.0f} GB/s"
        raw.vft_lora_wgrad_target(0)
        print(line, flush=True)


def attn():
    from vision_ft_amd import _C

    print("== attention (TFLOP/s: fwd 4BHSSD, bwd algorithmic 10BHSSD); v1 = 4-wave kernels, v0 = automatic (8-wave for D >= 128) ==")
    for B, H, Hkv, Sq, Sk, D in [(4, 10, 10, 4096, 4096, 64), (4, 20, 20, 1024, 1024, 64), (4, 10, 10, 4096, 77, 64),
                                 (4, 20, 20, 1024, 77, 64), (2, 12, 12, 4360, 4360, 256), (4, 12, 12, 4360, 4360, 256),
                                 (2, 24, 8, 4352, 4352, 96), (16, 16, 16, 2048, 2048, 128)]:
        q = bf(B, Sq, H, D).permute(0, 2, 1, 3).requires_grad_(True)
        k = bf(B, Sk, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True)
        v = bf(B, Sk, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True)
        do = bf(B, Sq, H, D).permute(0, 2, 1, 3)
        fl = 4.0 * B * H * Sq * Sk * D
        line = f"B={B} H={H}/{Hkv} Sq={Sq} Sk={Sk} D={D:3d}"
        outs = []
        for var in ((1, 0) if D != 96 else (0,)):
            _C.lib().vft_attn_variant(var)
            o = kernels.flash_attention(q, k, v)
            outs.append(o.detach().float())
            tf = timeit(lambda: kernels.flash_attention(q, k, v))

            def fb():
                kernels.flash_attention(q, k, v).backward(do)

            tfb = timeit(fb)
            line += f" | v{var} fwd {tf*1e6:8.1f} us {fl/tf/1e12:7.1f} TF/s  bwd {(tfb-tf)*1e6:8.1f} us {2.5*fl/(tfb-tf)/1e12:7.1f} TF/s"
        if len(outs) == 2:
            line += f" | rel diff {((outs[0]-outs[1]).norm()/outs[0].norm()).item():.1e}"
        _C.lib().vft_attn_variant(0)
        print(line, flush=True)


def norm():
    print("== norms / activations (GB/s, algorithmic bytes) ==")
    for rows, C in [(98304, 640), (24576, 1280), (17440, 3072), (17408, 2304)]:  # bench shapes (SDXL B=24, AuraFlow / Lumina2 B=4)
        x, w, b = bf(rows, C).requires_grad_(), bf(C), bf(C)
        dy = bf(rows, C)
        tf = timeit(lambda: kernels.layer_norm(x, w, b, 1e-5))
        y = kernels.layer_norm(x, w, b, 1e-5)
        tb = timeit(lambda: torch.autograd.grad(y, x, dy, retain_graph=True))
        print(f"layer_norm rows={rows} C={C}: fwd {tf*1e6:7.1f} us {4*rows*C/tf/1e9:6.0f} GB/s   bwd {tb*1e6:7.1f} us {6*rows*C/tb/1e9:6.0f} GB/s")
    for N, C, H, W in [(4, 320, 128, 128), (4, 640, 64, 64), (4, 1280, 32, 32), (4, 1920, 64, 64)]:
        x, w, b = bf(N, C, H, W), bf(C), bf(C)
        tf = timeit(lambda: kernels.group_norm(x, w, b, 32, 1e-5, True))
        print(f"group_norm+silu {N}x{C}x{H}x{W}: {tf*1e6:7.1f} us {6*x.numel()/tf/1e9:7.0f} GB/s")
    for M, F in [(16384, 2560), (4096, 5120)]:
        p = bf(M, 2 * F)
        tf = timeit(lambda: kernels.geglu(p))
        print(f"geglu M={M} F={F}: {tf*1e6:7.1f} us {6*M*F/tf/1e9:7.0f} GB/s")


def geglu_lora():
    print("== GEGLU: plain (on a materialised LoRA projection: lora_fwd pass + geglu) vs LoRA applied inside GEGLU ==")
    for M, C in [(65536, 640), (16384, 1280)]:
        Fh = 4 * C
        lin = torch.nn.Linear(C, 2 * Fh).to(dev, torch.bfloat16).requires_grad_(False)
        x = bf(M, C)
        A, B = bf(4, C).requires_grad_(True), bf(2 * Fh, 4).requires_grad_(True)
        proj = lin(x)
        t = kernels.lora_down(x, A.detach())
        gy = bf(M, Fh)
        pr = proj.clone().requires_grad_(True)
        tf = timeit(lambda: kernels.geglu(pr))
        tff = timeit(lambda: kernels._GeGLULoRA.apply(pr, t, B.detach(), 0.25))
        o1 = kernels.geglu(pr)
        tb = timeit(lambda: torch.autograd.grad(o1, pr, gy, retain_graph=True))
        o2 = kernels._GeGLULoRA.apply(pr, t, B.detach(), 0.25)
        tbf = timeit(lambda: torch.autograd.grad(o2, pr, gy, retain_graph=True))
        tl = timeit(lambda: kernels._lora_fwd_(x, proj, A.detach(), B.detach(), 0.25))
        print(f"M={M} C={C}: geglu fwd {tf*1e6:6.1f} us, fused fwd {tff*1e6:6.1f} us | bwd {tb*1e6:6.1f} us, fused bwd {tbf*1e6:6.1f} us"
              f" | lora_fwd pass it replaces {tl*1e6:6.1f} us", flush=True)


def gemm():
    import torch.nn.functional as F

    print("== GEMM NT bf16: ours vs hipBLASLt (F.linear), TFLOP/s; NF4-fused forward / dgrad ==")
    from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4

    for M, N, K in [(65536, 640, 640), (16384, 1280, 1280), (16384, 10240, 1280), (16384, 1280, 5120), (65536, 5120, 640),
                    (17408, 3840, 2304), (17408, 9216, 2304), (17408, 2304, 9216), (8720, 3072, 3072), (8720, 8192, 3072)]:
        x, w, b = bf(M, K), bf(N, K), bf(N)
        to = timeit(lambda: kernels.linear_fused(x, w, b), iters=10)
        tl = timeit(lambda: F.linear(x, w, b), iters=10)
        fl = 2.0 * M * N * K
        line = f"M={M:6d} N={N:5d} K={K:5d} | ours {to*1e6:8.1f} us {fl/to/1e12:7.1f} TF/s | hipBLASLt {tl*1e6:8.1f} us {fl/tl/1e12:7.1f} TF/s"
        if K % 64 == 0 and N % 64 == 0 and M <= 20000:
            q = quantize_nf4(w.float().cpu() * 0.02, double_quant=True)
            layer = NF4Linear(K, N, bias=False)
            layer.load_state_dict({"weight" + k: v for k, v in q.items()})
            layer = layer.to(dev)
            st = layer.quant_state()
            dy = bf(M, N)
            tq = timeit(lambda: kernels._nf4_gemm(x, st, None, False), iters=10)
            tqt = timeit(lambda: kernels._nf4_gemm(dy, st, None, True), iters=10)
            line += f" | nf4 fwd {fl/tq/1e12:6.1f} dgrad {fl/tqt/1e12:6.1f} TF/s"
        print(line)


def linear():
    import torch.nn.functional as F

    from vision_ft_amd import _C

    print("== fused linear (vft_linear_fwd): first-generation kernel (v-1), lin8 256x256 / 256x128 / 128x128 (v1-3), auto (v0) vs "
          "hipBLASLt F.linear; TFLOP/s; +ext = rank-4 extension and residual fused (auto) ==")
    shapes = [(65536, 640, 640), (65536, 5120, 640), (65536, 640, 2560), (16384, 1280, 1280), (16384, 10240, 1280),
              (16384, 1280, 5120), (98304, 640, 640), (98304, 1920, 640), (98304, 5120, 640), (98304, 640, 2560),
              (24576, 1280, 1280), (24576, 3840, 1280), (24576, 10240, 1280), (24576, 1280, 5120), (8720, 3072, 3072),
              (17440, 3072, 3072), (17440, 8192, 3072), (17440, 3072, 8192), (17408, 3840, 2304), (17408, 2304, 9216)]
    if len(sys.argv) > 2:
        shapes = shapes[: int(sys.argv[2])]
    for M, N, K in shapes:
        x, w, b = bf(M, K), bf(N, K) * 0.05, bf(N)
        ex, ew, res = bf(M, 4), bf(N, 4), bf(M, N)
        ref = F.linear(x, w, b)
        tl = timeit(lambda: F.linear(x, w, b), iters=10)
        fl = 2.0 * M * N * K
        line = f"M={M:6d} N={N:5d} K={K:5d} | hipBLASLt {fl/tl/1e12:6.1f}"
        for v in (-1, 1, 2, 3, 0):
            _C.lib().vft_linear_variant(v)
            y = kernels.linear_fused(x, w, b)
            err = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
            to = timeit(lambda: kernels.linear_fused(x, w, b), iters=10)
            line += f" | v{v} {fl/to/1e12:6.1f}" + (f" (err {err:.1e})" if err > 1e-3 else "")
        te = timeit(lambda: kernels.linear_fused(x, w, b, ext_x=ex, ext_w=ew, residual=res), iters=10)
        line += f" | +ext+res {te*1e6:7.1f} us vs plain {to*1e6:7.1f} us"
        print(line, flush=True)


def conv():
    from vision_ft_amd import _C

    print("== conv3x3 implicit GEMM: first-generation kernel vs lin8 configs; TFLOP/s ==")
    for B, H, W, Cin, Cout in [(24, 128, 128, 320, 320), (24, 128, 128, 640, 320), (24, 64, 64, 640, 640), (24, 64, 64, 1280, 640),
                               (24, 64, 64, 1920, 640), (24, 32, 32, 1280, 1280), (24, 32, 32, 2560, 1280), (24, 64, 64, 320, 640)]:
        x = bf(B, Cin, H, W).contiguous(memory_format=torch.channels_last)
        w = (bf(Cout, Cin, 3, 3) * 0.05).contiguous(memory_format=torch.channels_last)
        b = bf(Cout)
        fl = 18.0 * B * H * W * Cin * Cout
        _C.lib().vft_linear_variant(-1)
        ref = kernels._conv3x3_call(x, w, b, B, H, W, Cin, Cout, "conv")
        line = f"B={B} {H}x{W} {Cin:4d}->{Cout:4d}"
        for v in (-1, 1, 2, 3, 0):
            _C.lib().vft_linear_variant(v)
            y = kernels._conv3x3_call(x, w, b, B, H, W, Cin, Cout, "conv")
            err = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
            to = timeit(lambda: kernels._conv3x3_call(x, w, b, B, H, W, Cin, Cout, "conv"), iters=10)
            line += f" | v{v} {fl/to/1e12:6.1f}" + (f" (err {err:.1e})" if err > 1e-3 else "")
        print(line, flush=True)
    _C.lib().vft_linear_variant(0)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("lora", "all"):
        lora()
    if which in ("lora_down", "all"):
        lora_down()
    if which in ("attn", "all"):
        attn()
    if which in ("norm", "all"):
        norm()
    if which in ("geglu", "all"):
        geglu_lora()
    if which in ("gemm", "all"):
        gemm()
    if which in ("linear", "all"):
        linear()
    if which in ("conv", "all"):
        conv()
