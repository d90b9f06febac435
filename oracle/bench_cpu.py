"""CPU baseline leg of bench.py: times the oracle's restatement of BASELINE config 1 (SDXL UNet,
LoRA rank 4 on ["attn1","attn2",".ff."], batch 1, 512x512 -> latents 1x4x64x64, AdamW lr 1e-3,
autocast cpu bf16) on the host cores of the box it runs on.  TEST/BENCH INFRASTRUCTURE ONLY — a
reported baseline, never the product path."""
from __future__ import annotations

import time

import torch

from . import ops
from . import sdxl as osdxl


def _random_params(shapes: dict[str, tuple], lora_targets: list[str], rank: int) -> dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(42)
    P: dict[str, torch.Tensor] = {}
    for name, shape in shapes.items():
        if len(shape) == 1:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = (torch.rand(shape, generator=g) * 2 - 1) / fan_in**0.5
        P[name] = t
    for n in lora_targets:
        w = P.pop(f"{n}.weight")
        P[f"{n}.linear.weight"] = w
        if f"{n}.bias" in P:
            P[f"{n}.linear.bias"] = P.pop(f"{n}.bias")
        P[f"{n}.lora_down.weight"] = ((torch.rand(rank, w.shape[1], generator=g) * 2 - 1) / w.shape[1] ** 0.5).to(torch.bfloat16)
        P[f"{n}.lora_up.weight"] = torch.zeros(w.shape[0], rank, dtype=torch.bfloat16)
        P[f"{n}.alpha"] = torch.tensor(1.0, dtype=torch.bfloat16)
    return P


def time_sdxl_lora_step(shapes: dict[str, tuple], lora_targets: list[str], threads: int | list[int], warmup: int = 1,
                        steps: int = 2, rank: int = 4) -> dict:
    """`threads` may be a list: one warm-up step, then ONE timed step per thread count (same tensors, same optimizer state
    carried on; the sweep stops early once a count is more than 3x slower than the best so far — on a 128-core host the
    128-thread step alone took 50 s), then `steps - 1` more at the fastest count; the result is the mean over that count's steps.  A B = 1,
    1 024-token step does not scale to every core of a 128-core host (23.4 s/step on 128 threads against
    10.8 s on 8), so the reported baseline is the best of a small sweep, with the sweep in the result."""
    sweep = [threads] if isinstance(threads, int) else list(threads)
    torch.set_num_threads(sweep[0])
    cfg = osdxl.SDXLCfg()
    P = _random_params(shapes, lora_targets, rank)
    params = [v.requires_grad_(True) for k, v in P.items() if ".lora_down." in k or ".lora_up." in k]
    opt = torch.optim.AdamW(params, lr=1e-3)
    g = torch.Generator().manual_seed(42)
    lat = torch.randn(1, 4, 64, 64, generator=g)
    ctx = torch.randn(1, 77, 2048, generator=g)
    pooled = torch.randn(1, 1280, generator=g)
    size = torch.tensor([[512, 512]])
    crop = torch.tensor([[0, 0]])
    times: dict[int, list[float]] = {n: [] for n in sweep}
    schedule = [sweep[0]] * warmup + sweep
    i = -1
    while True:
        i += 1
        if i >= len(schedule):
            best = min(sweep, key=lambda n: sum(times[n]) / len(times[n]))
            if len(times[best]) >= steps:
                break
            schedule.append(best)
        torch.set_num_threads(schedule[i])
        t0 = time.perf_counter()
        t = torch.randint(0, 1000, (1,), generator=g, dtype=torch.int32)
        noise = torch.randn(lat.shape, generator=g)
        noisy = ops.ddpm_noise(lat, t, noise)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pred = osdxl.unet_forward(P, cfg, noisy, t, ctx, pooled, size, size, crop)
            loss = ops.ddpm_loss(noise, pred)
        loss.backward()
        opt.step()
        opt.zero_grad()
        if i >= warmup:
            times[schedule[i]].append(time.perf_counter() - t0)
            done = [n for n in sweep if times[n]]
            fastest = min(sum(times[n]) / len(times[n]) for n in done)
            if i + 1 < len(schedule) and schedule[i + 1] not in done and times[schedule[i]][-1] > 3.0 * fastest:
                # oversubscribed already: the larger counts still queued would only cost minutes of a bounded sample
                skipped = [n for n in schedule[i + 1:] if n not in done]
                schedule = schedule[: i + 1]
                sweep = [n for n in sweep if n not in skipped]
    s = sum(times[best]) / len(times[best])
    return {"s_per_step": s, "images_per_s": 1.0 / s, "loss": float(loss), "threads": best,
            "sweep_s_per_step": {n: (round(sum(v) / len(v), 2) if v else "skipped") for n, v in times.items()}}
