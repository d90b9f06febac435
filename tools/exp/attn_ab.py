"""A/B timing of the attention kernels on one GPU, one process (box-to-box variance rules out comparisons across calls):
forward and backward at the bench shapes, `vft_attn_variant` 0 (automatic: one-pass backward at head_dim 64) against 3
(the dq + dkdv pair; at head_dim >= 128 variant 0 is the stored-dS form), HIP events around each call, median of `reps` after a warm-up.  Prints algorithmic TFLOP/s
(4 B H Sq Sk D forward, 10 B H Sq Sk D backward).

    python tools/exp/attn_ab.py [sdxl|aura|lumina|all] [reps]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vision_ft_amd import _C, kernels  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = {
    "sdxl": [(24, 10, 10, 4096, 4096, 64), (24, 20, 20, 1024, 1024, 64), (24, 10, 10, 4096, 77, 64), (24, 20, 20, 1024, 77, 64)],
    "aura": [(6, 12, 12, 4360, 4360, 256)],
    "lumina": [(4, 24, 8, 4352, 4352, 96)],
}


def bf(*s):
    return torch.randn(*s, device=dev, dtype=torch.float32).to(torch.bfloat16)


def time_case(B, Hq, Hkv, Sq, Sk, D, variants, reps):
    """Median forward / backward ms per variant; the variants alternate inside every repetition (run one after the other,
    the first would be measured on a colder / differently clocked chip: the same forward kernel read 5 % apart that way)."""
    q = bf(B, Sq, Hq, D).permute(0, 2, 1, 3).requires_grad_(True)
    k, v = (bf(B, Sk, Hkv, D).permute(0, 2, 1, 3).requires_grad_(True) for _ in range(2))
    do = bf(B, Sq, Hq, D).permute(0, 2, 1, 3)
    times = {}
    try:
        for i in range(reps + 2):
            for variant in variants:
                if _C.lib().vft_attn_variant(variant) != 0:  # an older library (VFT_LIB_PATH) without this variant
                    continue
                e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                e[0].record()
                o = kernels.flash_attention(q, k, v)
                e[1].record()
                o.backward(do)
                e[2].record()
                torch.cuda.synchronize()
                q.grad = k.grad = v.grad = None
                if i >= 2:
                    f, b_ = times.setdefault(variant, ([], []))
                    f.append(e[0].elapsed_time(e[1]))
                    b_.append(e[1].elapsed_time(e[2]))
    finally:
        _C.lib().vft_attn_variant(0)
    return {var: (statistics.median(f), statistics.median(b_)) for var, (f, b_) in times.items()}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
    for name, shapes in SHAPES.items():
        if which not in ("all", name):
            continue
        for (B, Hq, Hkv, Sq, Sk, D) in shapes:
            unit = B * Hq * Sq * Sk * D / 1e9
            row = f"{name:7s} B{B} Hq{Hq} Hkv{Hkv} Sq{Sq} Sk{Sk} D{D}:"
            variants = tuple(int(v) for v in os.environ["VFT_AB_VARIANTS"].split(",")) if os.environ.get("VFT_AB_VARIANTS") else ((0, 3) if D != 96 else (0,))
            for variant, (f, b_) in time_case(B, Hq, Hkv, Sq, Sk, D, variants, reps).items():
                row += f"  [variant {variant}] fwd {f:7.3f} ms {4 * unit / f:7.1f} TF/s | bwd {b_:7.3f} ms {10 * unit / b_:7.1f} TF/s"
            print(row, flush=True)


if __name__ == "__main__":
    main()
