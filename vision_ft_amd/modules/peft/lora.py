"""LoRA adapter layers.

Interface and state-dict layout follow the reference (src/modules/peft/lora.py:11-309):
``<name>.linear.{weight,bias}`` (frozen base), ``<name>.lora_down.weight`` (rank, in),
``<name>.lora_up.weight`` (out, rank), ``<name>.alpha`` (0-d, frozen).  Forward:
``linear(x) + lora_up(lora_down(dropout(x))) * (alpha / rank)``  (:92-104).

MI355X path: the low-rank branch is two fused HIP kernels (``vft_lora_down`` — one pass over x —
and ``vft_lora_up_add`` — adds s·t·Bᵀ into the base GEMM's output in place) instead of two extra
GEMM launches plus a scale and an add; the backward produces dA, dB and the low-rank part of dx
in two passes (see kernels.lora_linear).  A base layer stored in NF4 (modules/quant/nf4.py)
runs the dequant-fused GEMM.
"""
from __future__ import annotations

import math
from typing import Literal

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import kernels
from ...utils.dtype import str_to_dtype
from .config import PeftConfigMixin
from .util import PeftLayer


class LoRAConfig(PeftConfigMixin):
    type: Literal["lora"] = "lora"
    rank: int
    alpha: float = 1.0
    dropout: float = 0.0
    use_bias: bool = False


def _alpha_param(value: float, dtype: torch.dtype, device=None) -> nn.Parameter:
    return nn.Parameter(torch.tensor(value, dtype=dtype, device=device), requires_grad=False)


class LoRALinear(PeftLayer):
    adapter_param_names = ["lora_up", "lora_down", "alpha"]
    adapter_weight_names = ["lora_up.weight", "lora_up.bias", "lora_down.weight", "alpha"]

    def __init__(self, config: LoRAConfig, original_linear: nn.Linear) -> None:
        super().__init__()
        self.config = config
        self.rank = config.rank
        self.enabled = True
        dtype = str_to_dtype(config.dtype)
        dev = original_linear.weight.device
        self.lora_down = nn.Linear(original_linear.in_features, config.rank, bias=False, dtype=dtype, device=dev)
        self.lora_up = nn.Linear(config.rank, original_linear.out_features, bias=config.use_bias, dtype=dtype, device=dev)
        self.dropout = nn.Dropout(config.dropout) if config.dropout > 0 else nn.Identity()
        self.alpha = _alpha_param(config.alpha, dtype, dev)
        self.linear = original_linear
        for p in self.linear.parameters():
            p.requires_grad_(False)
        self.init_weights()

    def init_weights(self) -> None:
        if self.lora_down.weight.is_meta:
            return
        # kaiming-uniform(down), zeros(up): the adapter starts as an exact no-op (reference :70-77)
        nn.init.kaiming_uniform_(self.lora_down.weight)
        nn.init.zeros_(self.lora_up.weight)
        if self.lora_up.bias is not None:
            nn.init.zeros_(self.lora_up.bias)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None) -> torch.Tensor:
        """`residual` (MI355X extension, optional): the enclosing block's skip connection, added in the same
        kernel pass as the low-rank update: returns residual + layer(x)."""
        if not self.enabled:
            y = self.linear(x)
            return y if residual is None else residual + y
        drop = self.dropout(x)
        if drop is x and self.lora_up.bias is None:
            return kernels.lora_linear(x, self.linear, self.lora_down.weight, self.lora_up.weight, self._scale(), residual,
                                       pack=self.__dict__.get("_vft_pack"))
        # dropout / adapter-bias variants are not on the benchmarked path: unfused but same math
        base = self.linear(x)
        lo = self.lora_up(self.lora_down(drop))
        y = base + lo * (self.alpha / self.rank)
        return y if residual is None else residual + y

    def training_dropout_active(self) -> bool:
        """True when the adapter input goes through an active dropout (or the up-projection has a bias): the
        fused kernels do not cover those variants."""
        d = self.dropout
        return (isinstance(d, nn.Dropout) and d.p > 0 and self.training) or self.lora_up.bias is not None

    def _scale(self) -> float:
        # alpha is a frozen 0-d parameter; cache its host value so the hot loop never syncs
        s = getattr(self, "_scale_cache", None)
        if s is None:
            s = float(self.alpha.detach().float().cpu()) / self.rank
            self._scale_cache = s
        return s

    def train(self, mode: bool = True) -> "LoRALinear":
        self.training = mode
        self.lora_down.train(mode)
        self.lora_up.train(mode)
        self.dropout.train(mode)
        self.linear.train(False)
        return self

    def requires_grad_(self, requires_grad: bool = True) -> "LoRALinear":
        self.lora_down.requires_grad_(requires_grad)
        self.lora_up.requires_grad_(requires_grad)
        self.linear.weight.requires_grad_(False)
        return self

    @classmethod
    def from_weights(cls, adapter_weights: dict[str, torch.Tensor], original_layer: nn.Linear) -> "LoRALinear":
        rank = adapter_weights["lora_down.weight"].shape[0]
        module = cls(LoRAConfig(rank=rank, alpha=adapter_weights["alpha"].item()), original_layer)
        module.load_weights(adapter_weights)
        return module

    def load_weights(self, adapter_weights: dict[str, torch.Tensor | None]) -> None:
        dev = self.linear.weight.device
        if (w := adapter_weights.get("lora_down.weight")) is not None:
            self.lora_down.weight = nn.Parameter(w.to(dev))
        if (w := adapter_weights.get("lora_up.weight")) is not None:
            self.lora_up.weight = nn.Parameter(w.to(dev))
        if (w := adapter_weights.get("lora_up.bias")) is not None:
            self.lora_up.bias = nn.Parameter(w.to(dev))
        if (w := adapter_weights.get("alpha")) is not None:
            self.alpha = nn.Parameter(w.to(dev), requires_grad=False)
            self._scale_cache = None


class LoRAConv2d(PeftLayer):
    """Conv2d adapter: down = same kernel/stride/padding conv to `rank` channels, up = 1x1 conv
    (reference :167-309).  Not targeted by any shipped config.  On the GPU the frozen BASE convolution runs the
    implicit-GEMM HIP kernel whenever its geometry allows (3x3 / stride 1 / padding 1 or 1x1, channels % 32 == 0 —
    `kernels.conv3x3` / `conv1x1`, forward and data gradient), whatever class the wrapped layer has; the rank-wide
    branch (4-16 output channels: below the kernel's 32-channel granule) stays on the library convolutions.
    Pinned by tests/golden/lora_conv.safetensors (reference-generated) on CPU and GPU."""

    adapter_param_names = ["lora_up", "lora_down", "alpha"]
    adapter_weight_names = ["lora_up.weight", "lora_up.bias", "lora_down.weight", "alpha"]

    def __init__(self, config: LoRAConfig, original_conv2d: nn.Conv2d) -> None:
        super().__init__()
        self.config = config
        self.rank = config.rank
        self.enabled = True
        dtype = str_to_dtype(config.dtype)
        c = original_conv2d
        self.lora_down = nn.Conv2d(c.in_channels, config.rank, kernel_size=c.kernel_size, stride=c.stride,
                                   padding=c.padding, bias=False, dtype=dtype)
        self.lora_up = nn.Conv2d(config.rank, c.out_channels, kernel_size=1, stride=1, bias=config.use_bias, dtype=dtype)
        self.dropout = nn.Dropout(config.dropout) if config.dropout > 0 else nn.Identity()
        self.alpha = _alpha_param(config.alpha, dtype)
        self.conv2d = c
        for p in self.conv2d.parameters():
            p.requires_grad_(False)
        self.init_weights()

    def init_weights(self) -> None:
        nn.init.kaiming_uniform_(self.lora_down.weight)
        nn.init.zeros_(self.lora_up.weight)

    def _base(self, x: torch.Tensor) -> torch.Tensor:
        c = self.conv2d
        if x.is_cuda and c.padding_mode == "zeros" and not isinstance(c.padding, str):
            if kernels.conv3x3_eligible(x, c.weight, c.stride, c.padding, c.dilation, c.groups):
                return kernels.conv3x3(x, c.weight, c.bias)
            if kernels.conv1x1_eligible(x, c.weight, c.stride, c.padding, c.dilation, c.groups):
                return kernels.conv1x1(x, c.weight, c.bias)
        return c(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self._base(x)
        if not self.enabled:
            return y
        return y + self.lora_up(self.lora_down(self.dropout(x))) * (self.alpha / self.rank)

    def train(self, mode: bool = True) -> "LoRAConv2d":
        super().train(mode)
        self.conv2d.train(False)
        return self

    def requires_grad_(self, requires_grad: bool = True) -> "LoRAConv2d":
        self.lora_down.requires_grad_(requires_grad)
        self.lora_up.requires_grad_(requires_grad)
        self.conv2d.requires_grad_(False)
        return self

    @classmethod
    def from_weights(cls, adapter_weights: dict[str, torch.Tensor], original_layer: nn.Conv2d) -> "LoRAConv2d":
        rank = adapter_weights["lora_down.weight"].shape[0]
        module = cls(LoRAConfig(rank=rank, alpha=adapter_weights["alpha"].item()), original_layer)
        module.load_weights(adapter_weights)
        return module

    def load_weights(self, adapter_weights: dict[str, torch.Tensor | None]) -> None:
        dev = self.conv2d.weight.device
        for key, target in (("lora_down.weight", (self.lora_down, "weight")), ("lora_up.weight", (self.lora_up, "weight")),
                            ("lora_up.bias", (self.lora_up, "bias"))):
            if (w := adapter_weights.get(key)) is not None:
                setattr(target[0], target[1], nn.Parameter(w.to(dev)))
        if (w := adapter_weights.get("alpha")) is not None:
            self.alpha = nn.Parameter(w.to(dev), requires_grad=False)
