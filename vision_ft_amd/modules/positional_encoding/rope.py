"""Flux-style multi-axis RoPE tables (reference: src/modules/positional_encoding/rope.py:5-162).

Tables are built on the host in fp64 exactly as the reference does (angles in fp64, cos/sin cast to
fp32, stacked (S, D/2, 2)); the rotation itself is fused with the per-head QK-norm in
``kernels.qk_norm_rope`` (interleaved pairs, fp32 math — reference :83-111)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import kernels


def image_position_indices(height: int, width: int, rope_dim: int = 3, y_index: int = 1, x_index: int = 2) -> torch.Tensor:
    hp, wp = height // 2, width // 2
    pos = torch.zeros(hp, wp, rope_dim)
    pos[..., y_index] = torch.arange(hp, dtype=pos.dtype).view(hp, 1)
    pos[..., x_index] = torch.arange(wp, dtype=pos.dtype).view(1, wp)
    return pos.view(hp * wp, rope_dim)


def get_rope_frequencies(position_indices: torch.Tensor, dim_sizes: list[int], theta: float) -> torch.Tensor:
    if len(dim_sizes) != position_indices.shape[-1]:
        raise AssertionError("dim_sizes must have the same length as position_indices.shape[-1]")
    tables = []
    for axis, dim in enumerate(dim_sizes):
        assert dim % 2 == 0, "dim must be even"
        exponent = torch.arange(0, dim, 2, dtype=torch.float64, device=position_indices.device) / dim
        angles = torch.outer(position_indices[..., axis].double(), theta ** (-exponent))
        tables.append(torch.stack((angles.cos(), angles.sin()), dim=-1).float())
    return torch.cat(tables, dim=-2)


def applye_rope_frequencies(inputs: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """(B, H, S, D) x (S, D/2, 2) -> rotated, same dtype.  (Name kept from the reference, typo and all.)"""
    return kernels.rope_apply(inputs, freqs)


def apply_rope_qk(q: torch.Tensor, k: torch.Tensor, rope_freqs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return applye_rope_frequencies(q, rope_freqs), applye_rope_frequencies(k, rope_freqs)


class RoPEFrequency(nn.Module):
    def __init__(self, dim_sizes: list[int], theta: float):
        super().__init__()
        self.dim_sizes = dim_sizes
        self.theta = theta

    def get_image_position_indices(self, height: int, width: int, y_index: int = 1, x_index: int = 2) -> torch.Tensor:
        return image_position_indices(height, width, len(self.dim_sizes), y_index, x_index)

    def get_text_position_indices(self, seq_len: int) -> torch.Tensor:
        return torch.zeros(seq_len, len(self.dim_sizes))

    def forward(self, position_indecis: torch.Tensor) -> torch.Tensor:
        return get_rope_frequencies(position_indecis, self.dim_sizes, self.theta)
