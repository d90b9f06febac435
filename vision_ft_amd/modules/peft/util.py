"""Adapter layer protocol (reference: src/modules/peft/util.py:10-49)."""
from abc import ABC, abstractmethod

import torch
import torch.nn as nn


class PeftLayer(ABC, nn.Module):
    adapter_param_names: list[str]
    adapter_weight_names: list[str]
    enabled: bool

    @abstractmethod
    def init_weights(self) -> None: ...

    def set_enabled(self, enabled: bool) -> None:
        self.enabled = enabled

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor: ...

    @classmethod
    @abstractmethod
    def from_weights(cls, adapter_weights: dict[str, torch.Tensor], original_layer: nn.Module) -> "PeftLayer": ...

    @abstractmethod
    def load_weights(self, adapter_weights: dict[str, torch.Tensor | None]) -> None: ...
