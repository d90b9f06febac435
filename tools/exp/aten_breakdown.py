"""Which framework (ATen / library) launches does one training step make, and from where?

    python tools/exp/aten_breakdown.py --workload sdxl_full --batch 16

Runs two eager warm-up steps, then one eager step under torch.profiler; prints the operator table sorted by device
time and, for the heaviest non-hand-written operators, the Python call sites.  Diagnostic only.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from vision_ft_amd.trainer.dp import DataParallelGrads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sdxl_full")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--top", type=int, default=40)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    model = bench.build(args.workload, device)
    params = [p for p in model.parameters() if p.requires_grad]
    dp = DataParallelGrads(params)
    opt = torch.optim.AdamW(params, lr=1e-3, fused=True)
    fwd = bench.make_step(args.workload, model, device, args.batch, seed=42)

    def step():
        loss = fwd()
        loss.backward()
        opt.step()
        dp.zero_grad()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    ka = prof.key_averages()
    print(ka.table(sort_by="self_cuda_time_total", row_limit=args.top, max_name_column_width=60))
    print("=" * 100)
    print("by input shape:")
    kshape = sorted(prof.key_averages(group_by_input_shape=True),
                    key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
    n = 0
    for e in kshape:
        t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if not e.key.startswith("aten::") or t <= 0:
            continue
        print(f"{e.key:28s} n={e.count:5d} dev={t / 1000:8.2f} ms  {str(e.input_shapes)[:150]}")
        n += 1
        if n >= 40:
            break
    print("=" * 100)
    ks = prof.key_averages(group_by_stack_n=6)
    rows = sorted(ks, key=lambda e: -getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)))
    shown = 0
    for e in rows:
        if not e.key.startswith("aten::"):
            continue
        t = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if t <= 0:
            continue
        print(f"{e.key:40s} n={e.count:5d} dev={t / 1000:8.2f} ms")
        for fr in e.stack[:6]:
            if "site-packages/torch" in fr or "<built-in" in fr:
                continue
            print("      ", fr[-140:])
        shown += 1
        if shown >= 45:
            break


if __name__ == "__main__":
    main()
