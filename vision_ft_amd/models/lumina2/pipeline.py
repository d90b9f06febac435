"""Lumina-Image-2.0 model holder: denoiser + checkpoint loader (reference: src/models/lumina2/pipeline.py:22-97)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..checkpoint import load_denoiser_checkpoint
from .config import Lumina2Config
from .denoiser import Denoiser
from .scheduler import Scheduler
from .util import convert_from_original_key


class Lumina2(nn.Module):
    denoiser: Denoiser
    denoiser_class: type[Denoiser] = Denoiser

    def __init__(self, config: Lumina2Config):
        super().__init__()
        self.config = config
        self.denoiser = self.denoiser_class(config.denoiser)
        self.scheduler = Scheduler()

    @classmethod
    def from_config(cls, config: Lumina2Config) -> "Lumina2":
        return cls(config)

    def _from_checkpoint(self, strict: bool = True) -> None:
        load_denoiser_checkpoint(self, self.config.checkpoint_path, convert_from_original_key, strict=strict)

    @classmethod
    def from_checkpoint(cls, config: Lumina2Config) -> "Lumina2":
        with torch.device("meta"):
            model = cls.from_config(config)
        model._from_checkpoint()
        return model

    def forward(self, *args, **kwargs):
        return self.denoiser(*args, **kwargs)
