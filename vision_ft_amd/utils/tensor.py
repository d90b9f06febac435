"""Small tensor/module helpers (reference: src/utils/tensor.py:131-176)."""
from __future__ import annotations

import torch
import torch.nn as nn


def remove_orig_mod_prefix(name: str) -> str:
    return name.replace("_orig_mod.", "", 1)


def swap_seq_len_and_num_heads(*args: torch.Tensor) -> tuple[torch.Tensor, ...]:
    """(B, S, H, D) <-> (B, H, S, D) views."""
    return tuple(a.transpose(1, 2) for a in args)


def set_requires_grad(model: nn.Module, requires_grad: bool = True) -> None:
    for p in model.parameters():
        p.requires_grad = requires_grad
    if hasattr(model, "set_requires_grad"):
        model.set_requires_grad(requires_grad)


def module_dtype(module: nn.Module) -> torch.dtype:
    """Compute dtype of a (possibly adapter-wrapped or quantised) layer: dtype of its first
    floating-point parameter."""
    for p in module.parameters():
        if p.is_floating_point():
            return p.dtype
    return torch.bfloat16
