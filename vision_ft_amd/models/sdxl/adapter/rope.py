"""SDXL with 2-D rotary position embedding on the UNet's attention (reference: src/models/sdxl/adapter/rope.py).

Same classes, constructor arguments, substitution hooks and state-dict keys as the reference
(`SelfAttentionWithRoPE` / `CrossAttentionWithRoPE` :196-316, `TransformerWithRoPE` :357-447,
`DenoiserWithRoPE` :459-536, `SDXLWithRoPEModel` :543-568): the RoPE modules own no parameters, so a plain SDXL
checkpoint loads into them unchanged.

MI355X execution: the rotation is the interleaved-pair (complex) multiply the Lumina2 / AuraFlow paths already run as
one HIP kernel (`kernels.qk_norm_rope`, kind "none": fp32 math on bf16 operands, forward and backward), applied to
the strided (B,S,H,D) views of the projection outputs; attention then reads the rotated q/k in the same layout.
cos/sin tables are built once per (height, width) / context length on the host exactly as the reference builds its
complex `freqs` (:21-38: float64 theta, fp32 radians, `torch.polar`) and kept on the device; they are the same for
every sample, so one (S, D/2) table serves the batch (the reference repeats it per sample)."""
from __future__ import annotations

import math
from collections import defaultdict
from contextlib import contextmanager
from typing import Literal

import torch
import torch.nn as nn

from .... import kernels
from ....modules.attention import AttentionImplementation, scaled_dot_product_attention
from ..config import DenoiserConfig, SDXLConfig
from ..denoiser import (CrossAttention, Denoiser, SelfAttention, TransformerBlock, _merge_heads, _proj, _proj_residual,
                        _split_heads)
from ..pipeline import SDXLModel

ORIGIN_POSITION = Literal["top_left", "center"]


def _get_rope_freqs(position_ids: torch.Tensor, dim: int, rope_theta: float = 10000.0) -> torch.Tensor:
    """complex64 e^{i * pos * theta_k}; reference :21-38 (float64 exponent, product rounded to fp32, torch.polar)."""
    assert dim % 2 == 0, "dim must be even"
    exponent = torch.arange(0, dim, 2, dtype=torch.float64, device=position_ids.device) / dim
    theta = 1.0 / (rope_theta**exponent)
    radians = (position_ids * theta).float()
    return torch.polar(torch.ones_like(radians), radians).to(torch.complex64)


def apply_rope(inputs: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """inputs (B,H,S,D), freqs complex (B,S,D/2) or (S,D/2): rotate interleaved pairs in fp32 (reference :41-54).
    On the GPU this is one fused kernel pass (forward and backward); the result keeps the (B,H,S,D) view convention."""
    cos, sin = freqs.real.contiguous(), freqs.imag.contiguous()  # the kernel takes (S,D/2) or per-sample (B,S,D/2) tables
    return _rotate(inputs, cos, sin)


def _rotate(x_bhsd: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    out = kernels.qk_norm_rope(x_bhsd.permute(0, 2, 1, 3), "none", None, 0.0, cos, sin)  # (B,S,H,D) contiguous
    return out.permute(0, 2, 1, 3)


class RoPEEmbedder(nn.Module):
    """Position ids and frequency tables; reference :60-190.  No parameters.  The cache keeps one cos/sin pair per
    (height, width) and per context length on the device they were first asked for."""

    def __init__(self, rope_dims: list[int], rope_theta: float = 10000.0, origin_position: ORIGIN_POSITION = "top_left"):
        super().__init__()
        self.rope_dims = rope_dims
        self.rope_theta = rope_theta
        self.origin_position: ORIGIN_POSITION = origin_position
        self.image_freqs: dict[int, dict[int, torch.Tensor]] = defaultdict(dict)
        self.context_freqs: dict[int, torch.Tensor] = {}
        self._tables: dict = {}

    def get_image_position_ids(self, batch_size: int, height: int, width: int) -> torch.Tensor:
        ids = torch.zeros(batch_size, height, width, len(self.rope_dims), dtype=torch.int64)
        y_ids = torch.arange(height, dtype=torch.int64).view(1, -1, 1).repeat(batch_size, 1, width)
        if self.origin_position == "center":
            y_ids = y_ids - math.ceil(height // 2)  # (0, 0) moves to the centre; `ceil(h // 2)` as in the reference (:93)
        ids[:, :, :, 0] = y_ids
        x_ids = torch.arange(width, dtype=torch.int64).view(1, 1, -1).repeat(batch_size, height, 1)
        if self.origin_position == "center":
            x_ids = x_ids - math.ceil(width // 2)
        ids[:, :, :, 1] = x_ids
        return ids.view(batch_size, height * width, len(self.rope_dims))

    def get_context_position_ids(self, batch_size: int, length: int) -> torch.Tensor:
        return torch.arange(length, dtype=torch.int64).view(1, -1, 1).repeat(batch_size, 1, 2)  # (i, i)

    def get_rope_freqs(self, position_ids: torch.Tensor) -> torch.Tensor:
        parts = [_get_rope_freqs(ids, dim, rope_theta=self.rope_theta)
                 for ids, dim in zip(position_ids.chunk(2, dim=-1), self.rope_dims, strict=True)]
        return torch.cat(parts, dim=-1)

    @torch.no_grad()
    def get_image_freqs(self, batch_size: int, height: int, width: int, device: torch.device) -> torch.Tensor:
        """complex64 (B, H*W, sum(rope_dims)/2), the reference's return value (:137-163)."""
        return self.get_rope_freqs(self.get_image_position_ids(batch_size, height, width)).detach().to(device)

    @torch.no_grad()
    def get_context_freqs(self, batch_size: int, length: int, device: torch.device) -> torch.Tensor:
        return self.get_rope_freqs(self.get_context_position_ids(batch_size, length)).detach().to(device)

    # ---- device-resident fp32 tables for the HIP kernel (same for every sample) -------------------------
    @torch.no_grad()
    def _table(self, key: tuple, ids_fn, device: torch.device) -> tuple[torch.Tensor, torch.Tensor]:
        key = key + (tuple(self.rope_dims), float(self.rope_theta), self.origin_position, str(device))
        hit = self._tables.get(key)
        if hit is None:
            f = self.get_rope_freqs(ids_fn())[0]
            hit = (f.real.contiguous().to(device), f.imag.contiguous().to(device))
            self._tables[key] = hit
        return hit

    def image_table(self, height: int, width: int, device: torch.device):
        return self._table(("img", height, width), lambda: self.get_image_position_ids(1, height, width), device)

    def context_table(self, length: int, device: torch.device):
        return self._table(("ctx", length), lambda: self.get_context_position_ids(1, length), device)


class _WithRoPE:
    rope_enabled: bool = True

    def set_rope_enabled(self, enabled: bool):
        self.rope_enabled = enabled


def _as_table(freqs) -> tuple[torch.Tensor, torch.Tensor]:
    """Accept the reference's complex (B,S,D/2) tensor or our (cos, sin) fp32 pair."""
    if isinstance(freqs, tuple):
        return freqs
    return freqs.real.contiguous(), freqs.imag.contiguous()


class SelfAttentionWithRoPE(SelfAttention, _WithRoPE):
    def __init__(self, num_heads: int, head_dim: int, dropout: float = 0.0,
                 attn_implementation: AttentionImplementation = "hip"):
        super().__init__(num_heads=num_heads, head_dim=head_dim, dropout=dropout, attn_implementation=attn_implementation)

    def forward(self, hidden_states: torch.Tensor, mask: torch.Tensor | None = None, image_freqs=None, *args,
                residual: torch.Tensor | None = None, **kwargs):
        wqkv = self._fused_qkv_weight() if hidden_states.is_cuda and hidden_states.dim() == 3 else None
        if wqkv is not None:
            q, k, v = kernels.lora_qkv(hidden_states, wqkv,
                                       [(p.lora_down.weight, p.lora_up.weight) for p in (self.to_q, self.to_k, self.to_v)],
                                       self.to_q._scale(), self.num_heads, pack=self.__dict__.get("_vft_qkv_pack"))
        else:
            q = _split_heads(_proj(self.to_q, hidden_states), self.num_heads, self.head_dim)
            k = _split_heads(_proj(self.to_k, hidden_states), self.num_heads, self.head_dim)
            v = _split_heads(_proj(self.to_v, hidden_states), self.num_heads, self.head_dim)
        if self.rope_enabled:
            assert image_freqs is not None
            cos, sin = _as_table(image_freqs)
            q = _rotate(q, cos, sin)
            k = _rotate(k, cos, sin)
        attn = scaled_dot_product_attention(q, k, v, mask=mask, backend=self.attn_implementation)
        return self.to_out[1](_proj_residual(self.to_out[0], _merge_heads(attn), residual))


class CrossAttentionWithRoPE(CrossAttention, _WithRoPE):
    def forward(self, query: torch.Tensor, context: torch.Tensor, mask: torch.Tensor | None = None, image_freqs=None,
                context_freqs=None, *args, residual: torch.Tensor | None = None, **kwargs) -> torch.Tensor:
        q = _split_heads(_proj(self.to_q, query), self.num_heads, self.head_dim)
        pk, pv = self._project_kv(context)
        k = _split_heads(pk, self.num_heads, self.head_dim)
        v = _split_heads(pv, self.num_heads, self.head_dim)
        if self.rope_enabled:
            assert image_freqs is not None and context_freqs is not None
            q = _rotate(q, *_as_table(image_freqs))
            k = _rotate(k, *_as_table(context_freqs))
        attn = scaled_dot_product_attention(q, k, v, mask=mask, backend=self.attn_implementation)
        return self.to_out[1](_proj_residual(self.to_out[0], _merge_heads(attn), residual))


class MigrationScale(nn.Module):
    """Learned blend ratio used while migrating a checkpoint to RoPE (reference :319-354)."""

    def __init__(self, init_ratio: float = 0.0, log_scale: bool = False):
        super().__init__()
        self.init_ratio = init_ratio
        self.log_scale = log_scale
        self.scale = nn.Parameter(torch.tensor(init_ratio, dtype=torch.float32), requires_grad=True)

    def init_weights(self):
        v = torch.tensor(self.init_ratio, dtype=torch.float32)
        self.scale.data = torch.exp(v) if self.log_scale else v

    def get_scale(self) -> torch.Tensor:
        return torch.log(self.scale) if self.log_scale else self.scale


class TransformerWithRoPE(TransformerBlock, _WithRoPE):
    self_attention_class: type[SelfAttention] = SelfAttentionWithRoPE
    cross_attention_class: type[CrossAttention] = CrossAttentionWithRoPE

    def __init__(self, hidden_dim: int, num_heads: int, head_dim: int, context_dim: int,
                 attn_implementation: AttentionImplementation = "hip", rope_dims: list[int] = [32, 32],
                 rope_theta: float = 10000.0, origin_position: ORIGIN_POSITION = "top_left"):
        super().__init__(hidden_dim=hidden_dim, num_heads=num_heads, head_dim=head_dim, context_dim=context_dim,
                         attn_implementation=attn_implementation)
        self.rope_embedder = RoPEEmbedder(rope_dims=rope_dims, rope_theta=rope_theta, origin_position=origin_position)

    def forward(self, hidden_states: torch.Tensor, context: torch.Tensor, time_embedding: torch.Tensor,
                height: int | None = None, width: int | None = None, cross_attention_kwargs: dict | None = None,
                *args, **kwargs) -> torch.Tensor:
        image_freqs = context_freqs = None
        if self.rope_enabled:
            assert height is not None and width is not None, "height and width must be provided for RoPE"
            dev = hidden_states.device
            image_freqs = self.rope_embedder.image_table(height, width, dev)
            context_freqs = self.rope_embedder.context_table(context.size(1), dev)
        # pre-LN residual x3 as in TransformerBlock.forward: the skip adds ride in the projections / norm backward
        x = hidden_states
        n, x = self.norm1.with_skip(x)
        x = self.attn1(n, image_freqs=image_freqs, residual=x)
        n, x = self.norm2.with_skip(x)
        x = self.attn2(n, context=context, time_embedding=time_embedding, image_freqs=image_freqs,
                       context_freqs=context_freqs, residual=x, **(cross_attention_kwargs or {}))
        n, x = self.norm3.with_skip(x)
        return self.ff(n, residual=x)


class DenoiserConfigWithRoPE(DenoiserConfig):
    rope_enabled: bool = True
    migrating: bool = False
    rope_dims: list[int] = [32, 32]
    rope_theta: float = 10000.0
    origin_position: ORIGIN_POSITION = "center"


class DenoiserWithRoPE(Denoiser):
    transformer_block_class: type[TransformerBlock] = TransformerWithRoPE

    def __init__(self, config: DenoiserConfigWithRoPE):
        super().__init__(config)
        self.rope_enabled = config.rope_enabled
        self.set_rope_enabled(config.rope_enabled)

    def set_rope_enabled(self, enabled: bool):
        self.rope_enabled = enabled
        for layer in self.modules():
            if isinstance(layer, _WithRoPE):
                layer.set_rope_enabled(enabled)


class SDXLWithRoPEConfig(SDXLConfig):
    denoiser: DenoiserConfigWithRoPE = DenoiserConfigWithRoPE()


class SDXLWithRoPEModel(SDXLModel):
    denoiser: DenoiserWithRoPE
    denoiser_class: type[Denoiser] = DenoiserWithRoPE
    config: SDXLWithRoPEConfig

    def __init__(self, config: SDXLWithRoPEConfig):
        super().__init__(config)
        self.apply_rope_config()

    def apply_rope_config(self):
        """The UNet builds its blocks with the constructor defaults; the model then pushes the configured
        rope_dims / theta / origin into every embedder (reference :553-560)."""
        for module in self.modules():
            if isinstance(module, _WithRoPE):
                module.set_rope_enabled(self.config.denoiser.rope_enabled)
            elif isinstance(module, RoPEEmbedder):
                module.rope_dims = self.config.denoiser.rope_dims
                module.rope_theta = self.config.denoiser.rope_theta
                module.origin_position = self.config.denoiser.origin_position


@contextmanager
def while_rope_enabled(model: SDXLWithRoPEModel):
    original_state = model.denoiser.rope_enabled
    model.denoiser.set_rope_enabled(True)
    try:
        yield
    finally:
        model.denoiser.set_rope_enabled(original_state)


@contextmanager
def while_rope_disabled(model: SDXLWithRoPEModel):
    original_state = model.denoiser.rope_enabled
    model.denoiser.set_rope_enabled(False)
    try:
        yield
    finally:
        model.denoiser.set_rope_enabled(original_state)
