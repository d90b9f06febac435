"""Normalisation layers with fp32 statistics (reference: src/modules/norm.py:9-27)."""
import torch
import torch.nn as nn

from .. import kernels


class FP32LayerNorm(nn.LayerNorm):
    """LayerNorm computed in fp32 and cast back to the input dtype."""

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return kernels.layer_norm(hidden_states, self.weight, self.bias, self.eps)


class FP32RMSNorm(nn.RMSNorm):
    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return kernels.rms_norm(hidden_states, self.weight, self.eps if self.eps is not None else 1e-6)
