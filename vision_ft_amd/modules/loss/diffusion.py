"""DDPM epsilon-prediction noising and loss (reference: src/modules/loss/diffusion.py:15-68).

The alpha-bar table is a per-device constant here (the reference rebuilds it on every call, :23-34);
values are identical (same fp32 linspace -> square -> cumprod)."""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn.functional as F  # noqa: F401

from .reduce import mse_mean


class NoisedLatents(NamedTuple):
    noisy_latents: torch.Tensor
    random_noise: torch.Tensor


_TABLES: dict = {}


def alphas_cumprod(device, beta_start: float = 0.00085, beta_end: float = 0.012, num_train_timesteps: int = 1000) -> torch.Tensor:
    key = (str(device), beta_start, beta_end, num_train_timesteps)
    if key not in _TABLES:
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32, device=device) ** 2
        _TABLES[key] = torch.cumprod(1.0 - betas, dim=0)
    return _TABLES[key]


def prepare_noised_latents(
    latents: torch.Tensor,
    timestep: torch.Tensor,
    max_sigma: float = 1.0,
    beta_start: float = 0.00085,
    beta_end: float = 0.012,
    num_train_timesteps: int = 1000,
    random_noise: torch.Tensor | None = None,
) -> NoisedLatents:
    abar = alphas_cumprod(latents.device, beta_start, beta_end, num_train_timesteps)[timestep.long()]
    shape = [-1] + [1] * (latents.dim() - 1)
    if random_noise is None:
        random_noise = torch.randn_like(latents) * max_sigma
    noisy = (abar**0.5).view(shape) * latents + ((1 - abar) ** 0.5).view(shape) * random_noise
    return NoisedLatents(noisy, random_noise)


def loss_with_predicted_noise(latents: torch.Tensor, random_noise: torch.Tensor, predicted_noise: torch.Tensor) -> torch.Tensor:
    # fp32 like the reference under autocast (mse_loss is an autocast-to-fp32 op)
    return mse_mean(predicted_noise, random_noise)
