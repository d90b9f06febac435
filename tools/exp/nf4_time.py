"""Timing of the NF4 GEMM forms at Lumina2's shapes (M = 17 408): python tools/exp/nf4_time.py"""
import sys

import torch

sys.path.insert(0, ".")
from vision_ft_amd import kernels  # noqa: E402
from vision_ft_amd.modules.quant import NF4Linear, quantize_nf4  # noqa: E402

dev = torch.device("cuda:0")


_spin = None


def timeit(fn, iters=30, warm=5):
    global _spin
    if _spin is None:
        _spin = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    for _ in range(40):  # ~100 ms of matrix work first: a timing that starts on an idle chip also measures its clock ramp
        _spin @ _spin
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 17408
for out_f, in_f in [(3840, 2304), (2304, 2304), (6144, 2304), (2304, 6144)]:
    w = torch.randn(out_f, in_f) * in_f**-0.5
    q = quantize_nf4(w, double_quant=True)
    layer = NF4Linear(in_f, out_f, bias=False)
    layer.load_state_dict({"weight" + k: v for k, v in q.items()}, strict=True)
    layer = layer.to(dev)
    qs = layer.quant_state()
    dense = layer.dequantize()
    dense_t = dense.t().contiguous()
    x = torch.randn(M, in_f, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, out_f, device=dev).to(torch.bfloat16)
    res = {}
    kernels.NF4_DEQUANT_ONCE = True
    res["once fwd"] = timeit(lambda: kernels._nf4_gemm(x, qs, None, False))
    res["once dgrad"] = timeit(lambda: kernels._nf4_gemm(dy, qs, None, True))
    # the same launch with the conversion / the barrier switched off, on a scratch that holds the REAL image from the calls above
    # (zero or constant operands would run the MFMAs at a higher clock: the chip is power-limited)
    import os
    for dbg, name in ((1, "no-conv"), (2, "no-barrier"), (3, "neither")):
        os.environ["VFT_NF4_DEBUG"] = str(dbg)
        res[f"once fwd {name}"] = timeit(lambda: kernels._nf4_gemm(x, qs, None, False))
    os.environ["VFT_NF4_DEBUG"] = "0"
    res["once fwd (again)"] = timeit(lambda: kernels._nf4_gemm(x, qs, None, False))
    res["once dgrad (again)"] = timeit(lambda: kernels._nf4_gemm(dy, qs, None, True))
    os.environ["VFT_NF4_DEBUG"] = "16"
    ws = kernels._nf4_workspace(dev, 1)
    for _ in range(3):
        kernels._nf4_gemm(x, qs, None, False)
    torch.cuda.synchronize()
    st = ws[4096:4096 + 256 * 32].view(torch.int64).view(256, 4).cpu().double()
    t0 = st[:, 0].min()
    tick = 0.01  # wall_clock64: 100 MHz
    print(f"   stamps (us after the first start): start max {float((st[:,0]-t0).max())*tick:.1f} | conversion done: min {float((st[:,1]-t0).min())*tick:.1f} median {float((st[:,1]-t0).median())*tick:.1f} max {float((st[:,1]-t0).max())*tick:.1f} | barrier passed: min {float((st[:,2]-t0).min())*tick:.1f} max {float((st[:,2]-t0).max())*tick:.1f}")
    ws[4096:16384].zero_()
    os.environ["VFT_NF4_DEBUG"] = "0"
    img = ws[16384:16384 + out_f * in_f * 2].view(torch.bfloat16).view(out_f, in_f)
    def two():
        kernels._C.lib().vft_nf4_dequant(img.data_ptr(), qs.packed.data_ptr(), qs.absmax.data_ptr(), int(qs.absmax_is_u8), qs.nested_absmax.data_ptr(),
                                         qs.nested_code.data_ptr(), qs.code.data_ptr(), qs.offset, qs.out_features, qs.in_features, torch.cuda.current_stream().cuda_stream)
        return kernels.linear_fused(x, img)
    res["two launches fwd"] = timeit(two)
    kernels.NF4_DEQUANT_ONCE = False
    res["inloop fwd"] = timeit(lambda: kernels._nf4_gemm(x, qs, None, False))
    res["bf16 fwd"] = timeit(lambda: kernels.linear_fused(x, dense))
    res["bf16 dgrad"] = timeit(lambda: kernels.linear_fused(dy, dense_t))
    res["dequant kernel"] = timeit(lambda: layer.dequantize())
    print(f"out {out_f} in {in_f}: " + "  ".join(f"{k} {v:7.1f}" for k, v in res.items()), flush=True)
