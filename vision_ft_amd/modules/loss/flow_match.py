"""Rectified-flow noising and velocity loss (reference: src/modules/loss/flow_match.py:13-116)."""
from __future__ import annotations

from typing import Literal, NamedTuple

import torch
import torch.nn.functional as F  # noqa: F401

from .reduce import mse_mean


class NoisedLatents(NamedTuple):
    noisy_latents: torch.Tensor
    random_noise: torch.Tensor


def _bcast(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return t.view([like.size(0)] + [1] * (like.dim() - 1))


def prepare_noised_latents(latents: torch.Tensor, timestep: torch.Tensor, max_sigma: float = 1.0,
                           random_noise: torch.Tensor | None = None) -> NoisedLatents:
    """x_t = (1 - t) x + t eps  (t = 0 clean, t = 1 noise)."""
    t = _bcast(timestep, latents)
    if random_noise is None:
        random_noise = torch.normal(mean=0.0, std=max_sigma, size=latents.shape, dtype=latents.dtype, device=latents.device)
    return NoisedLatents((1 - t) * latents + t * random_noise, random_noise)


def prepare_scaled_noised_latents(latents, timestep, noise_scale: float = 1.0, clean_at_zero: bool = False):
    noise = torch.randn_like(latents) * noise_scale
    t = _bcast(timestep, latents)
    noisy = (1 - t) * latents + t * noise if clean_at_zero else t * latents + (1 - t) * noise
    return NoisedLatents(noisy, noise)


def get_flow_match_target_velocity(latents: torch.Tensor, random_noise: torch.Tensor) -> torch.Tensor:
    return random_noise - latents


def loss_with_predicted_velocity(latents: torch.Tensor, random_noise: torch.Tensor, predicted_velocity: torch.Tensor) -> torch.Tensor:
    return mse_mean(predicted_velocity, random_noise - latents)


ModelPredictionType = Literal["noise", "velocity", "image"]


def convert_x0_to_velocity(x0, noisy_latents, timestep, eps: float = 1e-5, clean_at_zero: bool = False):
    t = _bcast(timestep, x0)
    if clean_at_zero:
        return (noisy_latents - x0) / t.clamp_min(eps)
    return (x0 - noisy_latents) / (1 - t).clamp_min(eps)
