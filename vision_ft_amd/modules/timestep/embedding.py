"""Sinusoidal timestep embedding (reference: src/modules/timestep/embedding.py:10-62). fp32, tiny."""
import math

import torch


def get_timestep_embedding(
    timesteps: torch.Tensor,
    embedding_dim: int,
    flip_sin_to_cos: bool = False,
    downscale_freq_shift: float = 1,
    scale: float = 1,
    max_period: int = 10000,
) -> torch.Tensor:
    assert timesteps.dim() == 1, "Timesteps should be a 1d-array"
    half = embedding_dim // 2
    freqs = torch.arange(half, dtype=torch.float32, device=timesteps.device) * (-math.log(max_period))
    freqs = torch.exp(freqs / (half - downscale_freq_shift))
    arg = scale * (timesteps.float().unsqueeze(1) * freqs.unsqueeze(0))
    first, second = (arg.cos(), arg.sin()) if flip_sin_to_cos else (arg.sin(), arg.cos())
    emb = torch.cat((first, second), dim=-1)
    if embedding_dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb
