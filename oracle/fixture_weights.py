"""Deterministic, RNG-stream-independent synthetic weights for fixtures and benchmarks.

Values come from numpy's PCG64 bit generator (stream stable across numpy versions) seeded by a
CRC of the parameter name, so the generator script (run next to the reference) and the tests (run
anywhere) build identical state dicts without storing them.  The golden files carry a checksum of
the weights that were used, which the tests verify.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def fill_like(name: str, shape, seed: int = 0, scale: float | None = None) -> torch.Tensor:
    """fp32 tensor of `shape`, uniform in [-a, a] with a = scale or 1/sqrt(fan_in)-like."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    rng = np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed]))
    u = rng.random(n, dtype=np.float64) * 2.0 - 1.0
    if scale is None:
        if len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            scale = 1.0 / np.sqrt(fan_in)
        else:
            scale = 0.1
    return torch.from_numpy((u * scale).astype(np.float32)).reshape(shape)


def synth_state_dict(shapes: dict[str, tuple], seed: int = 0, dtype=torch.float32) -> dict[str, torch.Tensor]:
    """Name-aware fill: norm weights ~ 1 + noise, biases small, everything else fan-in scaled."""
    out = {}
    for name, shape in shapes.items():
        if len(shape) == 0:
            out[name] = torch.tensor(1.0, dtype=dtype)  # LoRA alpha
            continue
        leaf = name.rsplit(".", 1)[-1]
        is_norm_w = leaf == "weight" and len(shape) == 1
        if is_norm_w:
            t = 1.0 + fill_like(name, shape, seed, 0.1)
        elif leaf == "bias":
            t = fill_like(name, shape, seed, 0.05)
        elif "lora_up" in name:
            t = fill_like(name, shape, seed, 0.05)  # non-zero so the adapter path is exercised
        elif name in ("positional_encoding", "register_tokens"):
            t = fill_like(name, shape, seed, 0.1)
        else:
            t = fill_like(name, shape, seed)
        out[name] = t.to(dtype)
    return out


def checksum(sd: dict[str, torch.Tensor]) -> float:
    tot = 0.0
    for k in sorted(sd):
        tot += float(sd[k].double().abs().sum())
    return tot
