"""SDXL model holder: the denoiser plus its checkpoint loader (reference: src/models/sdxl/pipeline.py:25-102).
VAE, text encoders, sampler and ``generate()`` are outside the hot path and are not part of this class."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..checkpoint import load_denoiser_checkpoint
from .config import SDXLConfig
from .denoiser import Denoiser
from .util import convert_from_original_key


class SDXLModel(nn.Module):
    denoiser: Denoiser
    denoiser_class: type[Denoiser] = Denoiser  # substitution hook, as in the reference (:27)

    def __init__(self, config: SDXLConfig):
        super().__init__()
        self.config = config
        self.denoiser = self.denoiser_class(config.denoiser)

    @classmethod
    def from_config(cls, config: SDXLConfig) -> "SDXLModel":
        return cls(config)

    def _from_checkpoint(self, strict: bool = True) -> None:
        load_denoiser_checkpoint(self, self.config.checkpoint_path, convert_from_original_key, strict=strict)

    @classmethod
    def from_checkpoint(cls, config: SDXLConfig) -> "SDXLModel":
        with torch.device("meta"):
            model = cls.from_config(config)
        model._from_checkpoint()
        return model

    def forward(self, *args, **kwargs):
        return self.denoiser(*args, **kwargs)
