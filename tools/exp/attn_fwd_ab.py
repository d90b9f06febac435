"""Forward attention A/B at head_dim 64 / 96: vft_attn_variant 0 (round 4: attn_fwd3_kernel, LDS-DMA operand path) against 1
(the round-1 4-wave kernel), alternating inside every repetition.  python tools/exp/attn_fwd_ab.py [reps]"""
import statistics
import sys

import torch

sys.path.insert(0, ".")
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 9
bf = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)  # noqa: E731
for name, (B, Hq, Hkv, Sq, Sk, D, lens) in {
        "sdxl self 128^2": (24, 10, 10, 4096, 4096, 64, None), "sdxl self 64^2": (24, 20, 20, 1024, 1024, 64, None),
        "sdxl cross 128^2": (24, 10, 10, 4096, 77, 64, None), "sdxl cross 64^2": (24, 20, 20, 1024, 77, 64, None),
        "lumina": (4, 24, 8, 4352, 4352, 96, [4352, 4200, 4300, 4128]), "lumina refiner": (4, 24, 8, 256, 256, 96, [256, 100, 200, 32])}.items():
    q = bf(B, Sq, Hq, D).permute(0, 2, 1, 3)
    k, v = bf(B, Sk, Hkv, D).permute(0, 2, 1, 3), bf(B, Sk, Hkv, D).permute(0, 2, 1, 3)
    kv = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
    outs, times = {}, {0: [], 1: []}
    with torch.no_grad():
        for i in range(reps + 2):
            for var in (0, 1):
                _C.lib().vft_attn_variant(var)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                o = kernels.flash_attention(q, k, v, kv_len=kv)
                e1.record()
                torch.cuda.synchronize()
                outs[var] = o
                if i >= 2:
                    times[var].append(e0.elapsed_time(e1))
    _C.lib().vft_attn_variant(0)
    t0, t1 = statistics.median(times[0]), statistics.median(times[1])
    unit = 4.0 * B * Hq * Sq * Sk * D / 1e9
    d = (outs[0].float() - outs[1].float()).norm() / outs[1].float().norm()
    print(f"{name:18s} new {t0 * 1e3:8.1f} us {unit / t0:7.1f} TF/s | old {t1 * 1e3:8.1f} us {unit / t1:7.1f} TF/s | new vs old rel.diff {float(d):.2e}", flush=True)
