"""AuraFlow model holder: denoiser + checkpoint loader (reference: src/models/auraflow/pipeline.py:35-136).
Original AuraFlow single-file checkpoints keep the MMDiT under ``model.`` (``DENOISER_TENSOR_PREFIX``,
src/models/auraflow/denoiser.py:32); ComfyUI exports use ``diffusion_model.``."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..checkpoint import load_denoiser_checkpoint
from .config import AuraFlowConig
from .denoiser import Denoiser

DENOISER_TENSOR_PREFIX = "model."
VAE_TENSOR_PREFIX = "vae."
TEXT_ENCODER_TENSOR_PREFIX = "text_encoders.pile_t5xl.transformer."  # src/models/auraflow/text_encoder.py:50


def convert_from_original_key(key: str) -> str:
    # same (unanchored, every-occurrence) replacements and order as the reference (:49-54)
    key = key.replace("diffusion_model.", "denoiser.")
    key = key.replace(DENOISER_TENSOR_PREFIX, "denoiser.")
    key = key.replace(VAE_TENSOR_PREFIX, "vae.")
    return key.replace(TEXT_ENCODER_TENSOR_PREFIX, "text_encoder.model.")


def convert_to_original_key(key: str) -> str:
    key = key.replace("denoiser.", DENOISER_TENSOR_PREFIX)
    key = key.replace("vae.", VAE_TENSOR_PREFIX)
    return key.replace("text_encoder.model.", TEXT_ENCODER_TENSOR_PREFIX)


def convert_to_comfy_key(key: str) -> str:
    key = key.replace("denoiser.", "diffusion_model.")
    key = key.replace("vae.", VAE_TENSOR_PREFIX)
    return key.replace("text_encoder.model.", TEXT_ENCODER_TENSOR_PREFIX)


class AuraFlowModel(nn.Module):
    denoiser: Denoiser
    denoiser_class: type[Denoiser] = Denoiser

    def __init__(self, config: AuraFlowConig):
        super().__init__()
        self.config = config
        self.denoiser = self.denoiser_class(config.denoiser)

    @classmethod
    def from_config(cls, config: AuraFlowConig) -> "AuraFlowModel":
        return cls(config)

    def _load_original_weights(self, strict: bool = True) -> None:
        load_denoiser_checkpoint(self, self.config.checkpoint_path, convert_from_original_key, strict=strict)

    _from_checkpoint = _load_original_weights

    @classmethod
    def from_original_checkpoint(cls, config: AuraFlowConig) -> "AuraFlowModel":
        with torch.device("meta"):
            model = cls.from_config(config)
        model._load_original_weights()
        return model

    def forward(self, *args, **kwargs):
        return self.denoiser(*args, **kwargs)
