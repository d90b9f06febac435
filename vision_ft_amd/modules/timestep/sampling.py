"""Train-time timestep samplers (reference: src/modules/timestep/sampling.py:10-272)."""
from __future__ import annotations

import math
from typing import Callable, Literal

import numpy as np
import torch


def get_lin_function(x1: float = 256, y1: float = 0.5, x2: float = 4096, y2: float = 1.15) -> Callable[[float], float]:
    slope = (y2 - y1) / (x2 - x1)
    intercept = y1 - slope * x1
    return lambda x: slope * x + intercept


def time_shift(mu: float, sigma: float, t: torch.Tensor) -> torch.Tensor:
    return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)


def _logit_normal(batch: int, device, scale: float) -> torch.Tensor:
    return (torch.randn(batch, device=device) * scale).sigmoid()


def sigmoid_randn(latents_shape: torch.Size, device: torch.device, sigmoid_scale: float = 1.0) -> torch.Tensor:
    return _logit_normal(latents_shape[0], device, sigmoid_scale)


def shift_sigmoid_randn(latents_shape, device, discrete_flow_shift: float = 3.1825, sigmoid_scale: float = 1.0):
    t = _logit_normal(latents_shape[0], device, sigmoid_scale)
    return (t * discrete_flow_shift) / (1 + (discrete_flow_shift - 1) * t)


def flux_shift_randn(latents_shape, device, sigmoid_scale: float = 1.0):
    _b, _c, h, w = latents_shape
    t = _logit_normal(latents_shape[0], device, sigmoid_scale)
    mu = get_lin_function(y1=0.5, y2=1.15)((h // 2) * (w // 2))
    return time_shift(mu, 1.0, t)


def scale_shift_sigmoid_randn(latents_shape, device, std: float = 0.8, mean: float = -0.8, **kwargs):
    return (torch.randn(latents_shape[0], device=device) * std + mean).sigmoid()


def uniform_rand(latents_shape: torch.Size, device: torch.device) -> torch.Tensor:
    return torch.rand(latents_shape[0], device=device)


def shift_uniform_rand(latents_shape, device, shift: float = 6.0):
    t = torch.rand(latents_shape[0], device=device)
    return (t * shift) / (1 + (shift - 1) * t)


def _create_fraction(denominators: list[int]) -> np.ndarray:
    return np.array(list({i / d for d in denominators for i in range(d + 1)}), dtype=np.float32)


def fraction_uniform_rand(latents_shape, device, divisible: list[int] = list(range(20, 30))) -> torch.Tensor:
    assert len(divisible) > 0, "divisible must not be empty"
    picks = np.random.choice(_create_fraction(divisible), size=latents_shape[0], replace=True)  # numpy global RNG, as the reference
    return torch.from_numpy(picks).to(device=device, dtype=torch.float32)


def shift_fraction_uniform_rand(latents_shape, device, shift: float = 6.0, divisible: list[int] = list(range(20, 30))):
    t = fraction_uniform_rand(latents_shape, device, divisible)
    return (t * shift) / (1 + (shift - 1) * t)


TimestepSamplingType = Literal["shift_sigmoid", "flux_shift", "sigmoid", "uniform", "scale_shift_sigmoid"]


def sample_timestep(latents_shape, device, sampling_type: TimestepSamplingType = "sigmoid", **kwargs):
    table = {
        "shift_sigmoid": shift_sigmoid_randn,
        "flux_shift": flux_shift_randn,
        "sigmoid": sigmoid_randn,
        "scale_shift_sigmoid": scale_shift_sigmoid_randn,
    }
    if sampling_type == "uniform":
        return uniform_rand(latents_shape, device)
    if sampling_type not in table:
        raise ValueError(f"Invalid sampling type: {sampling_type}")
    return table[sampling_type](latents_shape, device, **kwargs)


def uniform_randint(latents_shape: torch.Size, device: torch.device, min_timesteps: int = 0, max_timesteps: int = 1000):
    return torch.randint(min_timesteps, max_timesteps, (latents_shape[0],), device=device, dtype=torch.int32)
