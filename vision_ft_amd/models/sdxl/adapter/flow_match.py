"""SDXL trained with a flow-matching objective (reference: src/models/sdxl/adapter/flow_match.py:14-25).

Only the training-side surface is part of the hot path: the config fields the flow-match task reads
(`model_prediction`, `noise_scale`, `clean_at_zero`, `timestep_eps`) and the model holder.  `generate()` and its
Euler sampler (:48-190) belong to inference and are out of scope."""
from __future__ import annotations

import torch

from ....modules.loss.flow_match import ModelPredictionType
from ..config import SDXLConfig
from ..pipeline import SDXLModel


class SDXLFlowMatchConfig(SDXLConfig):
    model_prediction: ModelPredictionType = "velocity"
    noise_scale: float = 1.0
    clean_at_zero: bool = False
    timestep_eps: float = 1e-5


class SDXLFlowMatch(SDXLModel):
    config: SDXLFlowMatchConfig

    def prepare_timesteps(self, num_inference_steps: int, device: torch.device):
        """1000 -> 1 timesteps and their sigmas with a trailing zero (reference :28-46); used by previews."""
        timesteps = torch.linspace(1000.0, 1.0, num_inference_steps, device=device, dtype=torch.float32)
        sigmas = torch.cat([timesteps / 1000.0, torch.zeros(1, device=device)])
        return timesteps, sigmas
