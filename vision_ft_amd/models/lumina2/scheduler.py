"""Lumina-Image-2.0 flow-match schedule: the train-time part (reference: src/models/lumina2/scheduler.py:9-72).

Only what the training step calls is kept: the resolution-dependent logit-normal ("lognorm") timestep sampler.
The inference half (`get_timesteps` / `get_sigmas` / `step`) belongs to `generate()` and is out of scope."""
from __future__ import annotations

import math

import torch

from ...modules.timestep.sampling import get_lin_function, sigmoid_randn


class Scheduler:
    shift: float = 6.0
    num_train_timesteps: int = 1000
    base_shift: float = 0.5
    max_shift: float = 1.15
    base_image_seq_len: int = 256
    max_image_seq_len: int = 4096

    def sample_sigmoid_randn(self, latents_shape: torch.Size, device: torch.device, patch_size: int = 2,
                             sigma: float = 1.0) -> torch.Tensor:
        """scheduler.py:48-72: t ~ sigmoid(N(0,1)), shifted by mu(seq_len) on Lumina2's reversed time axis
        (0 = noise, 1 = clean)."""
        _b, _c, height, width = latents_shape
        t = sigmoid_randn(latents_shape, device)
        seq_len = (height // patch_size) * (width // patch_size)
        mu = get_lin_function(x1=self.base_image_seq_len, y1=self.base_shift, x2=self.max_image_seq_len,
                              y2=self.max_shift)(seq_len)
        t = 1 - t
        t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** sigma)
        return 1 - t
