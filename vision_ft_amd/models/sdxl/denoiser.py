"""SDXL UNet denoiser for MI355X.

Operator surface of the reference's src/models/sdxl/denoiser.py: same class names, constructor
arguments, substitution hooks (``TransformerBlock.self_attention_class`` / ``cross_attention_class``,
``UNet.transformer_block_class``), forward signature (:1180-1190) and — critically — the same
parameter names and shapes, so reference checkpoints load with ``strict=True`` and LoRA targeting
strings ("attn1", "attn2", ".ff.") select the same 700 Linears.

What differs is the execution: attention runs the HIP flash kernel on strided (B,S,H,D) views of
the projection outputs (no head permute copies), GroupNorm+SiLU / LayerNorm / GEGLU are fused HIP
kernels with fp32 statistics, and activations stay bf16 end to end (the reference reaches the same
cast points through ``accelerator.autocast()``).
"""
from __future__ import annotations

import inspect
import os
from typing import Literal, NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from ... import kernels
from ...modules.attention import AttentionImplementation, scaled_dot_product_attention
from ...modules.peft.lora import LoRALinear
from ...modules.timestep.embedding import get_timestep_embedding
from ...utils.tensor import module_dtype
from .config import DOWN_BLOCK_NAME, MID_BLOCK_NAME, UP_BLOCK_NAME, DenoiserConfig


class MLPEmbedder(nn.Sequential):
    def __init__(self, hidden_dim: int, time_embed_dim: int):
        super().__init__(nn.Linear(hidden_dim, time_embed_dim), nn.SiLU(), nn.Linear(time_embed_dim, time_embed_dim))


def _split_heads(t: torch.Tensor, heads: int, head_dim: int) -> torch.Tensor:
    # (B, S, H*D) -> (B, H, S, D) as a strided view; the attention kernel reads it in place
    b, s, _ = t.shape
    return t.view(b, s, heads, head_dim).permute(0, 2, 1, 3)


def _proj_residual(layer: nn.Module, h: torch.Tensor, residual: torch.Tensor | None) -> torch.Tensor:
    """residual + layer(h); a LoRA-wrapped projection folds the add into its own kernel pass, a plain nn.Linear runs on
    the fused projection GEMM (bias and the add in its epilogue)."""
    if isinstance(layer, LoRALinear):
        return layer(h) if residual is None else layer(h, residual=residual)
    if type(layer) is nn.Linear and kernels.linear_eligible(h, layer.weight):
        return kernels.linear(h, layer.weight, layer.bias, residual)
    if type(layer) is nn.Linear and kernels.trained_linear_eligible(h, layer.weight):
        return kernels.linear_trained(h, layer.weight, layer.bias, residual)  # full fine-tune: all three GEMMs on own kernels
    return layer(h) if residual is None else residual + layer(h)


def _proj(layer: nn.Module, h: torch.Tensor) -> torch.Tensor:
    return _proj_residual(layer, h, None)


def _merge_heads(t: torch.Tensor) -> torch.Tensor:
    b, h, s, d = t.shape
    return t.permute(0, 2, 1, 3).reshape(b, s, h * d)  # free: the kernel writes (B,S,H,D) memory


class SelfAttention(nn.Module):
    def __init__(self, num_heads: int, head_dim: int, dropout: float, attn_implementation: AttentionImplementation = "hip"):
        super().__init__()
        self.inner_dim = num_heads * head_dim
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.dropout = dropout
        self.attn_implementation: AttentionImplementation = attn_implementation
        self.to_q = nn.Linear(self.inner_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(self.inner_dim, self.inner_dim, bias=False)
        self.to_v = nn.Linear(self.inner_dim, self.inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(self.inner_dim, self.inner_dim), nn.Dropout(dropout))

    def _fused_qkv_weight(self) -> torch.Tensor | None:
        """[Wq; Wk; Wv] of the frozen bases when q/k/v are LoRA-wrapped bf16 Linears of one rank (else None):
        one base GEMM then serves the three projections (kernels.lora_qkv).  Cached; rebuilt if a base changes."""
        ps = (self.to_q, self.to_k, self.to_v)
        if not all(isinstance(p, LoRALinear) and p.enabled and not p.training_dropout_active()
                   and type(p.linear) is nn.Linear and p.linear.bias is None and p.linear.weight.dtype == torch.bfloat16
                   and not p.linear.weight.requires_grad and p.rank == ps[0].rank and p._scale() == ps[0]._scale() for p in ps):
            return None
        key = tuple((p.linear.weight.data_ptr(), p.linear.weight._version) for p in ps)
        if getattr(self, "_wqkv_key", None) != key:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._wqkv = torch.cat([p.linear.weight.detach() for p in ps], dim=0).contiguous()
            self._wqkv_key = key
        return self._wqkv

    def forward(self, hidden_states: torch.Tensor, mask: torch.Tensor | None = None, residual: torch.Tensor | None = None):
        wqkv = self._fused_qkv_weight() if hidden_states.is_cuda and hidden_states.dim() == 3 else None
        if wqkv is not None:
            q, k, v = kernels.lora_qkv(hidden_states, wqkv,
                                       [(p.lora_down.weight, p.lora_up.weight) for p in (self.to_q, self.to_k, self.to_v)],
                                       self.to_q._scale(), self.num_heads, pack=self.__dict__.get("_vft_qkv_pack"))
        else:
            q = _split_heads(_proj(self.to_q, hidden_states), self.num_heads, self.head_dim)
            k = _split_heads(_proj(self.to_k, hidden_states), self.num_heads, self.head_dim)
            v = _split_heads(_proj(self.to_v, hidden_states), self.num_heads, self.head_dim)
        attn = scaled_dot_product_attention(q, k, v, mask=mask, backend=self.attn_implementation)
        return self.to_out[1](_proj_residual(self.to_out[0], _merge_heads(attn), residual))


class CrossAttention(nn.Module):
    def __init__(self, query_dim: int, context_dim: int, num_heads: int, head_dim: int, dropout: float,
                 attn_implementation: AttentionImplementation = "hip"):
        super().__init__()
        self.query_dim = query_dim
        self.context_dim = context_dim
        self.inner_dim = num_heads * head_dim
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.dropout = dropout
        self.attn_implementation: AttentionImplementation = attn_implementation
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.to_k = nn.Linear(context_dim, self.inner_dim, bias=False)
        self.to_v = nn.Linear(context_dim, self.inner_dim, bias=False)
        self.to_out = nn.Sequential(nn.Linear(self.inner_dim, query_dim), nn.Dropout(dropout))

    def _fused_kv_weight(self) -> torch.Tensor | None:
        """[Wk; Wv] of the frozen bases when to_k / to_v are LoRA-wrapped bf16 Linears of one rank (else None): the two
        context projections then run as one GEMM (kernels.lora_group).  Cached; rebuilt if a base changes."""
        ps = (self.to_k, self.to_v)
        if not all(isinstance(p, LoRALinear) and p.enabled and not p.training_dropout_active()
                   and type(p.linear) is nn.Linear and p.linear.bias is None and p.linear.weight.dtype == torch.bfloat16
                   and not p.linear.weight.requires_grad and p.rank == ps[0].rank and p._scale() == ps[0]._scale() for p in ps):
            return None
        key = tuple((p.linear.weight.data_ptr(), p.linear.weight._version) for p in ps)
        if getattr(self, "_wkv_key", None) != key:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._wkv = torch.cat([p.linear.weight.detach() for p in ps], dim=0).contiguous()
            self._wkv_key = key
        return self._wkv

    def _project_kv(self, context: torch.Tensor):
        wkv = self._fused_kv_weight() if FUSE_CROSS_KV and context.is_cuda and context.dim() == 3 else None
        if wkv is not None:
            kv = kernels.lora_group(context, wkv, [(p.lora_down.weight, p.lora_up.weight) for p in (self.to_k, self.to_v)],
                                    self.to_k._scale(), pack=self.__dict__.get("_vft_kv_pack"))
            if kv is not None:
                return kv
        return _proj(self.to_k, context), _proj(self.to_v, context)

    def forward(self, query: torch.Tensor, context: torch.Tensor, mask: torch.Tensor | None = None,
                time_embedding: torch.Tensor | None = None, *args, residual: torch.Tensor | None = None, **kwargs) -> torch.Tensor:
        q = _split_heads(_proj(self.to_q, query), self.num_heads, self.head_dim)
        pk, pv = self._project_kv(context)
        k = _split_heads(pk, self.num_heads, self.head_dim)
        v = _split_heads(pv, self.num_heads, self.head_dim)
        attn = scaled_dot_product_attention(q, k, v, mask=mask, backend=self.attn_implementation)
        return self.to_out[1](_proj_residual(self.to_out[0], _merge_heads(attn), residual))


# The FeedForward's two LoRA-wrapped projections as one call (kernels.lora_geglu_down): both adapters inside their
# projection's GEMM, a plain GEGLU kernel in between, the block's skip connection in the output projection's epilogue.
FUSE_LORA_INTO_GEGLU = True
FUSE_CROSS_KV = os.environ.get("VFT_FUSE_KV", "1") != "0"  # A/B switch: to_k | to_v of the cross-attention context as one GEMM


class GeGLU(nn.Module):
    def __init__(self, in_dim: int, out_dim: int):
        super().__init__()
        self.proj = nn.Linear(in_dim, out_dim * 2)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        p = self.proj
        if (FUSE_LORA_INTO_GEGLU and isinstance(p, LoRALinear) and p.enabled and not p.training_dropout_active()
                and kernels.lora_geglu_eligible(hidden_states, p.linear, p.lora_down.weight)):
            return kernels.lora_linear_geglu(hidden_states, p.linear, p.lora_down.weight, p.lora_up.weight, p._scale())
        return kernels.geglu(_proj(p, hidden_states))


class FeedForward(nn.Module):
    def __init__(self, hidden_dim: int, multiplier: float = 4, dropout: float = 0.0):
        super().__init__()
        self.intermediate_dim = int(hidden_dim * multiplier)
        self.net = nn.Sequential(GeGLU(hidden_dim, self.intermediate_dim), nn.Dropout(dropout),
                                 nn.Linear(self.intermediate_dim, hidden_dim, bias=True))

    def forward(self, hidden_states: torch.Tensor, residual: torch.Tensor | None = None) -> torch.Tensor:
        up, down = self.net[0].proj, self.net[2]
        if (FUSE_LORA_INTO_GEGLU and isinstance(up, LoRALinear) and isinstance(down, LoRALinear) and up.enabled and down.enabled
                and not up.training_dropout_active() and not down.training_dropout_active()
                and not (isinstance(self.net[1], nn.Dropout) and self.net[1].p > 0 and self.training)
                and kernels.lora_geglu_eligible(hidden_states, up.linear, up.lora_down.weight)
                and type(down.linear) is nn.Linear and down.linear.weight.dtype == torch.bfloat16
                and down.rank <= (4 if up.rank <= 4 else 8)):
            return kernels.lora_geglu_down(hidden_states, up.linear, up.lora_down.weight, up.lora_up.weight, up._scale(),
                                           down.linear, down.lora_down.weight, down.lora_up.weight, down._scale(), residual,
                                           up_pack=up.__dict__.get("_vft_pack"), down_pack=down.__dict__.get("_vft_pack"))
        return _proj_residual(self.net[2], self.net[1](self.net[0](hidden_states)), residual)


class _LN(nn.LayerNorm):
    """nn.LayerNorm parameters, fused fp32-statistics kernel."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.layer_norm(x, self.weight, self.bias, self.eps)

    def with_skip(self, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """(norm(x), x): route the second output into the branch's residual so the backward adds the skip
        gradient inside the norm-backward kernel."""
        return kernels.layer_norm_skip(x, self.weight, self.bias, self.eps)


class TransformerBlock(nn.Module):
    self_attention_class: type[SelfAttention] = SelfAttention
    cross_attention_class: type[CrossAttention] = CrossAttention

    def __init__(self, hidden_dim: int, num_heads: int, head_dim: int, context_dim: int = 2048,
                 attn_implementation: AttentionImplementation = "hip"):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.context_dim = context_dim
        self.attn_implementation: AttentionImplementation = attn_implementation
        self.attn1 = self.self_attention_class(num_heads=num_heads, head_dim=head_dim, dropout=0.0,
                                               attn_implementation=attn_implementation)
        self.ff = FeedForward(hidden_dim=hidden_dim, dropout=0.0)
        self.attn2 = self.cross_attention_class(query_dim=hidden_dim, context_dim=context_dim, num_heads=num_heads,
                                                head_dim=head_dim, dropout=0.0, attn_implementation=attn_implementation)
        self.norm1 = _LN(hidden_dim)
        self.norm2 = _LN(hidden_dim)
        self.norm3 = _LN(hidden_dim)
        # substituted attention classes (self_attention_class / cross_attention_class hooks) may not know `residual`
        self._res1 = "residual" in inspect.signature(self.attn1.forward).parameters
        self._res2 = "residual" in inspect.signature(self.attn2.forward).parameters

    def forward(self, hidden_states: torch.Tensor, context: torch.Tensor, time_embedding: torch.Tensor,
                cross_attention_kwargs: dict | None = None, *args, **kwargs) -> torch.Tensor:
        # pre-LN residual x3 (reference :273-284); each skip add rides in the branch's last projection (forward)
        # and in the norm's backward kernel (backward) instead of separate elementwise passes
        x = hidden_states
        n, x = self.norm1.with_skip(x)
        x = self.attn1(n, residual=x) if self._res1 else x + self.attn1(n)
        n, x = self.norm2.with_skip(x)
        ca = cross_attention_kwargs or {}
        x = (self.attn2(n, context=context, time_embedding=time_embedding, residual=x, **ca) if self._res2
             else x + self.attn2(n, context=context, time_embedding=time_embedding, **ca))
        n, x = self.norm3.with_skip(x)
        x = self.ff(n, residual=x)
        return x


class _Conv2d(nn.Conv2d):
    """nn.Conv2d parameters; every convolution of the UNet runs the implicit-GEMM HIP kernel (forward, data and weight
    gradient): 3x3 / stride 1 / padding 1 and 1x1 directly, the 4-channel stem / head on zero-padded channels, the
    stride-2 Downsample through the kernel's stride-2 gather (round 5; MIOpen only for geometries none of these take)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if kernels.conv3x3_eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return kernels.conv3x3(x, self.weight, self.bias)
        if kernels.conv1x1_eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return kernels.conv1x1(x, self.weight, self.bias)
        if self.padding_mode == "zeros" and not isinstance(self.padding, str):
            if kernels.conv3x3_narrow_eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return kernels.conv3x3_narrow(x, self.weight, self.bias)
            if kernels.conv3x3_s2_eligible(x, self.weight, self.stride, self.padding, self.dilation, self.groups):
                return kernels.conv3x3_s2(x, self.weight, self.bias)
        if self.bias is not None and self.bias.requires_grad and x.is_cuda and torch.is_grad_enabled() and self.padding_mode == "zeros":
            # a TRAINABLE bias of a convolution that stays on MIOpen (stem, head, stride-2): the convolution runs without it
            # and the bias is added by the op whose backward does not use ATen's long reduction (kernels.add_channel_bias)
            return kernels.add_channel_bias(F.conv2d(x, self.weight, None, self.stride, self.padding, self.dilation, self.groups), self.bias)
        return super().forward(x)


class _GN(nn.GroupNorm):
    """nn.GroupNorm parameters; fused GroupNorm(+SiLU) kernel."""

    fuse_silu: bool = False

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return kernels.group_norm(x, self.weight, self.bias, self.num_groups, self.eps, silu=self.fuse_silu)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels: int, num_heads: int, head_dim: int, context_dims: list[int] = [2048],
                 attn_implementation: AttentionImplementation = "hip",
                 transformer_block_class: type[TransformerBlock] = TransformerBlock):
        super().__init__()
        self.in_channels = in_channels
        self.inner_dim = num_heads * head_dim
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.context_dims = context_dims
        self.attn_implementation = attn_implementation
        self.norm = _GN(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, self.inner_dim, bias=True)
        self.transformer_blocks = nn.ModuleList([
            transformer_block_class(hidden_dim=self.inner_dim, num_heads=num_heads, head_dim=head_dim,
                                    context_dim=cd, attn_implementation=attn_implementation)
            for cd in context_dims
        ])
        self.proj_out = nn.Linear(self.inner_dim, in_channels, bias=True)

    def forward(self, hidden_states: torch.Tensor, context: torch.Tensor | None = None,
                time_embedding: torch.Tensor | None = None, transformer_args: dict | None = {},
                cross_attention_kwargs: dict | None = None) -> torch.Tensor:
        b, c, h, w = hidden_states.shape
        tokens = self.norm(hidden_states).permute(0, 2, 3, 1).reshape(b, h * w, c)
        tokens = _proj(self.proj_in, tokens)
        extra = dict(transformer_args or {}, height=h, width=w)
        for block in self.transformer_blocks:
            tokens = block(tokens, context=context, time_embedding=time_embedding,
                           cross_attention_kwargs=cross_attention_kwargs, **extra)
        # + hidden_states rides in proj_out's epilogue: on channels-last storage the skip's token view is free
        skip = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, c)
        tokens = _proj_residual(self.proj_out, tokens, skip)
        return tokens.reshape(b, h, w, self.inner_dim).permute(0, 3, 1, 2)


class Downsample(nn.Module):
    def __init__(self, hidden_dim: int, out_channels: int, use_resample: bool, padding: int = 1,
                 kernel_size: int = 3, stride: int = 2):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.out_channels = out_channels
        self.use_resample = use_resample
        self.op = (_Conv2d(hidden_dim, out_channels, kernel_size=kernel_size, stride=stride, padding=padding)
                   if use_resample else nn.AvgPool2d(kernel_size=stride, stride=stride))

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        assert hidden_states.size(1) == self.hidden_dim
        return self.op(hidden_states)


class Upsample(nn.Module):
    def __init__(self, hidden_dim: int, out_channels: int, use_resample: bool, padding: int = 1,
                 kernel_size: int = 3, scale_factor: int = 2):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.out_channels = out_channels
        self.scale_factor = scale_factor
        self.conv = (_Conv2d(hidden_dim, out_channels, kernel_size=kernel_size, padding=padding)
                     if use_resample else nn.Identity())

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        assert hidden_states.size(1) == self.hidden_dim
        return self.conv(F.interpolate(hidden_states, scale_factor=self.scale_factor, mode="nearest"))


UPDOWN_TYPE = Literal["up", "down", "none"]


def updown_layer(hidden_dim: int, out_channels: int, updown_type: UPDOWN_TYPE, use_resample: bool = False,
                 kernel_size: int = 3, padding: int = 1, stride: int = 2):
    if updown_type == "down":
        return Downsample(hidden_dim, out_channels, use_resample=use_resample, padding=padding,
                          kernel_size=kernel_size, stride=stride)
    if updown_type == "up":
        return Upsample(hidden_dim, out_channels, use_resample=use_resample, padding=padding,
                        kernel_size=kernel_size, scale_factor=2)
    if updown_type == "none":
        return nn.Identity()
    raise ValueError(f"Invalid updown_type: {updown_type}")


class ResidualBlock(nn.Module):
    """GN32+SiLU -> conv3x3 -> (+ Linear(SiLU(emb))) -> GN32+SiLU -> conv3x3 -> + skip(x).

    Sequential indices are the reference's (in_layers.{0,2}, emb_layers.1, out_layers.{0,3}) so the
    state dict matches; the SiLU members are kept as placeholders and the activation is fused into
    the GroupNorm kernel."""

    def __init__(self, hidden_dim: int, embedding_dim: int, dropout: float, out_channels: int,
                 updown_type: UPDOWN_TYPE, kernel_size: int = 3, num_norm_groups: int = 32):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.embedding_dim = embedding_dim
        self.dropout = dropout
        self.out_channels = out_channels
        self.updown_type = updown_type
        pad = kernel_size // 2
        gn_in = _GN(num_norm_groups, hidden_dim, eps=1e-5, affine=True)
        gn_in.fuse_silu = True
        self.in_layers = nn.Sequential(gn_in, nn.SiLU(), _Conv2d(hidden_dim, out_channels, kernel_size, padding=pad))
        self.h_upd = updown_layer(hidden_dim, out_channels, updown_type, use_resample=False)
        self.x_upd = updown_layer(hidden_dim, out_channels, updown_type, use_resample=False)
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(embedding_dim, out_channels))
        gn_out = _GN(num_norm_groups, out_channels, eps=1e-5, affine=True)
        gn_out.fuse_silu = True
        self.out_layers = nn.Sequential(gn_out, nn.SiLU(), nn.Dropout(dropout),
                                        _Conv2d(out_channels, out_channels, kernel_size, padding=pad))
        self.skip_connection = (_Conv2d(hidden_dim, out_channels, kernel_size=1, padding=0)
                                if hidden_dim != out_channels else nn.Identity())

    def forward(self, hidden_states: torch.Tensor, embedding: torch.Tensor) -> torch.Tensor:
        residual = hidden_states
        h = self.in_layers[0](hidden_states)  # GroupNorm + SiLU (fused)
        if self.updown_type in ("up", "down"):
            h = self.h_upd(h)
            residual = self.x_upd(residual)
        h = self.in_layers[2](h)
        emb = self.emb_layers[1](F.silu(embedding))
        h = kernels.add_channel_bias(h, emb)  # h + emb[:, :, None, None]; the (H, W) sum of its gradient on vft_colsum
        h = self.out_layers[0](h)  # GroupNorm + SiLU (fused)
        h = self.out_layers[3](self.out_layers[2](h))
        skip = self.skip_connection
        if isinstance(skip, nn.Conv2d) and kernels.conv1x1_eligible(residual, skip.weight, skip.stride, skip.padding,
                                                                    skip.dilation, skip.groups):
            return kernels.conv1x1(residual, skip.weight, skip.bias, add=h)  # the add rides in the GEMM's epilogue
        return h + skip(residual)


def _forward_layer(layer: nn.Module, *args, gradient_checkpointing: bool = False, use_reentrant: bool = False):
    if gradient_checkpointing and torch.is_grad_enabled():
        return checkpoint.checkpoint(layer, *args, use_reentrant=use_reentrant)
    return layer(*args)


def _run_stage(layers: nn.ModuleList, x, context, global_embedding, time_embedding, transformer_args,
               cross_attention_kwargs, ckpt: bool):
    for layer in layers:
        if isinstance(layer, ResidualBlock):
            x = _forward_layer(layer, x, global_embedding, gradient_checkpointing=ckpt)
        elif isinstance(layer, SpatialTransformer):
            x = _forward_layer(layer, x, context, time_embedding, transformer_args, cross_attention_kwargs,
                               gradient_checkpointing=ckpt)
        elif isinstance(layer, (Downsample, Upsample, nn.Conv2d)):
            x = _forward_layer(layer, x, gradient_checkpointing=ckpt)
        else:
            raise ValueError(f"Invalid layer: {layer}")
    return x


class DownBlocksOutput(NamedTuple):
    hidden_states: torch.Tensor
    skip_connections: list[torch.Tensor]


class DownBlocks(nn.Module):
    def __init__(self, in_channels: int = 4, block_out_channels: list[int] = [320, 640, 1280],
                 down_blocks: list[DOWN_BLOCK_NAME] = ["DownBlock2D", "TransformerDownBlock2D", "TransformerDownBlock2D"],
                 num_transformers_per_block: list[int] = [1, 2, 10], layers_per_block: int = 2,
                 time_embed_dim: int = 320 * 4, dropout: float = 0.0, conv_resample: bool = True,
                 num_head_channels: int = 64, context_dim: int = 2048,
                 attn_implementation: AttentionImplementation = "hip",
                 transformer_block_class: type[TransformerBlock] = TransformerBlock):
        super().__init__()
        self.blocks = nn.ModuleList()
        width = in_channels
        last = len(down_blocks) - 1
        for i, (kind, out_ch, n_tf) in enumerate(zip(down_blocks, block_out_channels, num_transformers_per_block, strict=True)):
            if kind == "DownBlock2D":
                self.blocks.append(nn.ModuleList([_Conv2d(in_channels, block_out_channels[0], kernel_size=3, padding=1)]))
                width = out_ch
                for _ in range(layers_per_block):
                    self.blocks.append(nn.ModuleList([ResidualBlock(width, time_embed_dim, dropout, out_channels=out_ch, updown_type="none")]))
            elif kind == "TransformerDownBlock2D":
                for _ in range(layers_per_block):
                    self.blocks.append(nn.ModuleList([
                        ResidualBlock(width, time_embed_dim, dropout, out_channels=out_ch, updown_type="none"),
                        SpatialTransformer(in_channels=out_ch, num_heads=out_ch // num_head_channels, head_dim=num_head_channels,
                                           context_dims=[context_dim] * n_tf, attn_implementation=attn_implementation,
                                           transformer_block_class=transformer_block_class),
                    ]))
                    width = out_ch
            else:
                raise ValueError(f"Invalid block: {kind}")
            if i != last:
                self.blocks.append(nn.ModuleList([Downsample(out_ch, out_ch, use_resample=conv_resample)]))
        self.gradient_checkpointing = False

    def forward(self, hidden_states, context, global_embedding, time_embedding, transformer_args: dict | None = {},
                cross_attention_kwargs: dict | None = {}) -> DownBlocksOutput:
        skips: list[torch.Tensor] = []
        for stage in self.blocks:
            hidden_states = _run_stage(stage, hidden_states, context, global_embedding, time_embedding,
                                       transformer_args, cross_attention_kwargs, self.gradient_checkpointing)
            skips.append(hidden_states)
        return DownBlocksOutput(hidden_states, skips)


class MidBlock(nn.Module):
    def __init__(self, hidden_dim: int = 1280, time_embed_dim: int = 1280, mid_block_type: MID_BLOCK_NAME = "TransformerMidBlock2D",
                 num_transformers: int = 10, dropout: float = 0.0, num_head_channels: int = 64, context_dim: int = 2048,
                 attn_implementation: AttentionImplementation = "hip",
                 transformer_block_class: type[TransformerBlock] = TransformerBlock):
        super().__init__()
        self.blocks = nn.ModuleList([ResidualBlock(hidden_dim, time_embed_dim, dropout, out_channels=hidden_dim, updown_type="none")])
        if mid_block_type == "TransformerMidBlock2D":
            self.blocks.append(SpatialTransformer(in_channels=hidden_dim, num_heads=hidden_dim // num_head_channels,
                                                  head_dim=num_head_channels, context_dims=[context_dim] * num_transformers,
                                                  attn_implementation=attn_implementation,
                                                  transformer_block_class=transformer_block_class))
        self.blocks.append(ResidualBlock(hidden_dim, time_embed_dim, dropout, out_channels=hidden_dim, updown_type="none"))
        self.gradient_checkpointing = False

    def forward(self, hidden_states, context, global_embedding, time_embedding, transformer_args: dict | None = {},
                cross_attention_kwargs: dict | None = {}) -> torch.Tensor:
        return _run_stage(self.blocks, hidden_states, context, global_embedding, time_embedding, transformer_args,
                          cross_attention_kwargs, self.gradient_checkpointing)


class UpBlocks(nn.Module):
    def __init__(self, in_channels: int = 1280, block_out_channels: list[int] = [1280, 640, 320],
                 down_skip_channels: list[int] = [320, 320, 320, 320, 640, 640, 640, 1280, 1280],
                 up_blocks: list[UP_BLOCK_NAME] = ["TransformerUpBlock2D", "TransformerUpBlock2D", "UpBlock2D"],
                 num_transformers_per_block: list[int] = [0, 2, 10], layers_per_block: int = 2,
                 time_embed_dim: int = 320 * 4, dropout: float = 0.0, conv_resample: bool = True,
                 num_head_channels: int = 64, context_dim: int = 2048,
                 attn_implementation: AttentionImplementation = "hip",
                 transformer_block_class: type[TransformerBlock] = TransformerBlock):
        super().__init__()
        self.blocks = nn.ModuleList()
        skips = list(down_skip_channels)
        width = in_channels
        last = len(up_blocks) - 1
        for i, (kind, out_ch, n_tf) in enumerate(zip(up_blocks, block_out_channels, num_transformers_per_block, strict=True)):
            stage_tail: nn.ModuleList | None = None
            for _ in range(layers_per_block):
                members: list[nn.Module] = [ResidualBlock(width + skips.pop(), time_embed_dim, dropout, out_channels=out_ch, updown_type="none")]
                width = out_ch
                if kind == "TransformerUpBlock2D":
                    members.append(SpatialTransformer(in_channels=out_ch, num_heads=out_ch // num_head_channels,
                                                      head_dim=num_head_channels, context_dims=[context_dim] * n_tf,
                                                      attn_implementation=attn_implementation,
                                                      transformer_block_class=transformer_block_class))
                elif kind != "UpBlock2D":
                    raise ValueError(f"Invalid block: {kind}")
                stage_tail = nn.ModuleList(members)
                self.blocks.append(stage_tail)
            if i != last and stage_tail is not None:
                stage_tail.append(Upsample(out_ch, out_ch, use_resample=conv_resample))
        self.gradient_checkpointing = False

    def forward(self, hidden_states, context, global_embedding, time_embedding, skip_connections: list[torch.Tensor],
                transformer_args: dict | None = {}, cross_attention_kwargs: dict | None = {}) -> torch.Tensor:
        for stage in self.blocks:
            hidden_states = torch.cat([hidden_states, skip_connections.pop()], dim=1)
            hidden_states = _run_stage(stage, hidden_states, context, global_embedding, time_embedding,
                                       transformer_args, cross_attention_kwargs, self.gradient_checkpointing)
        return hidden_states


class UNet(nn.Module):
    transformer_block_class: type[TransformerBlock] = TransformerBlock

    def __init__(self, in_channels: int = 4, out_channels: int = 4, hidden_dim: int = 320, dropout: float = 0.0,
                 conv_resample: bool = True, num_head_channels: int = 64, context_dim: int = 2048,
                 global_cond_dim: int = 2816, additional_cond_dim: int = 256,
                 block_out_channels: list[int] = [320, 640, 1280], num_transformers_per_block: list[int] = [1, 2, 10],
                 layers_per_block: int = 2,
                 down_blocks: list[DOWN_BLOCK_NAME] = ["DownBlock2D", "TransformerDownBlock2D", "TransformerDownBlock2D"],
                 mid_block: MID_BLOCK_NAME = "TransformerMidBlock2D",
                 up_blocks: list[UP_BLOCK_NAME] = ["TransformerUpBlock2D", "TransformerUpBlock2D", "UpBlock2D"],
                 attn_implementation: AttentionImplementation = "hip"):
        super().__init__()
        self.in_channels = in_channels
        self.hidden_dim = hidden_dim
        self.out_channels = out_channels
        self.dropout = dropout
        self.conv_resample = conv_resample
        self.num_head_channels = num_head_channels
        time_embed_dim = hidden_dim * 4
        self.time_embed_dim = time_embed_dim
        self.additional_cond_dim = additional_cond_dim

        self.time_embed = MLPEmbedder(hidden_dim, time_embed_dim)
        self.label_emb = nn.Sequential(MLPEmbedder(global_cond_dim, time_embed_dim))

        common = dict(num_head_channels=num_head_channels, time_embed_dim=time_embed_dim, dropout=dropout,
                      context_dim=context_dim, transformer_block_class=self.transformer_block_class)
        # NB (reference quirk, kept for checkpoint compatibility): UNet.__init__ does not forward
        # num_transformers_per_block to the down path (sdxl/denoiser.py:1060-1072), so the down blocks
        # always use DownBlocks' default [1, 2, 10]; only the mid and up paths follow the config.
        self.input_blocks = DownBlocks(in_channels=in_channels, block_out_channels=block_out_channels, down_blocks=down_blocks,
                                       layers_per_block=layers_per_block, conv_resample=conv_resample,
                                       attn_implementation=attn_implementation, **common)
        self.middle_block = MidBlock(hidden_dim=block_out_channels[-1], mid_block_type=mid_block,
                                     num_transformers=num_transformers_per_block[-1], attn_implementation=attn_implementation,
                                     **common)
        # widths of the tensors pushed on the skip stack by the down path
        skip_channels: list[int] = []
        for i, (kind, ch) in enumerate(zip(down_blocks, block_out_channels, strict=True)):
            skip_channels += [ch] * (3 if kind == "DownBlock2D" else 2)
            if i != len(down_blocks) - 1:
                skip_channels.append(ch)
        self.output_blocks = UpBlocks(in_channels=block_out_channels[-1], block_out_channels=block_out_channels[::-1],
                                      down_skip_channels=skip_channels, up_blocks=up_blocks,
                                      num_transformers_per_block=num_transformers_per_block[::-1],
                                      layers_per_block=layers_per_block + 1, conv_resample=conv_resample,
                                      attn_implementation=attn_implementation, **common)
        gn = _GN(32, hidden_dim, eps=1e-5, affine=True)
        gn.fuse_silu = True
        self.out = nn.Sequential(gn, nn.SiLU(), _Conv2d(hidden_dim, out_channels, kernel_size=3, padding=1))

    def get_timestep_embedding(self, timestep: torch.Tensor, dim: int) -> torch.Tensor:
        return get_timestep_embedding(timestep, embedding_dim=dim, flip_sin_to_cos=True, downscale_freq_shift=0.0)

    def to_channels_last(self) -> "UNet":
        """Store every convolution weight (and its gradient view, if one exists) channels-last.

        Activations then stay NHWC through the whole network: MIOpen's implicit-GEMM convolutions run
        without layout transposes, GroupNorm uses the NHWC kernel, and the (N,C,H,W)->(N,HW,C) token
        reshapes around each SpatialTransformer are views instead of copies.  Shapes, state-dict keys
        and values are unchanged (memory format only).  Called once, lazily, by the first CUDA forward."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                w = m.weight
                if w.is_contiguous(memory_format=torch.channels_last):
                    continue
                g = w.grad
                w.data = w.data.contiguous(memory_format=torch.channels_last)
                if g is not None:  # e.g. a view into the trainer's flat gradient buffer: re-stride in place
                    vals = g.clone()
                    w.grad = g.as_strided(w.size(), w.stride())
                    w.grad.copy_(vals)
        self._nhwc = True
        return self

    def prepare_global_condition(self, timestep, text_pooler_output, original_size, target_size, crop_coords, dtype):
        t_emb = self.time_embed(self.get_timestep_embedding(timestep, self.hidden_dim).to(dtype))
        b = text_pooler_output.size(0)
        sizes = torch.cat([original_size, crop_coords, target_size], dim=1).flatten()
        size_emb = self.get_timestep_embedding(sizes, self.additional_cond_dim).reshape(b, -1)
        cond = torch.cat([text_pooler_output.float(), size_emb], dim=1).to(dtype=dtype)
        return t_emb, self.label_emb(cond) + t_emb

    def forward(self, latents: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                encoder_pooler_output: torch.Tensor, original_size: torch.Tensor, target_size: torch.Tensor,
                crop_coords_top_left: torch.Tensor, cross_attention_kwargs: dict | None = None) -> torch.Tensor:
        dtype = module_dtype(self.out[2])
        latents = latents.to(dtype)
        if latents.is_cuda:
            # MI355X layout: the whole UNet runs channels-last (see to_channels_last)
            if not getattr(self, "_nhwc", False):
                self.to_channels_last()
            latents = latents.contiguous(memory_format=torch.channels_last)
        context = encoder_hidden_states.to(dtype)
        t_emb, g_emb = self.prepare_global_condition(timestep, encoder_pooler_output, original_size, target_size,
                                                     crop_coords_top_left, dtype)
        x, skips = self.input_blocks(latents, context=context, global_embedding=g_emb, time_embedding=t_emb,
                                     cross_attention_kwargs=cross_attention_kwargs)
        x = self.middle_block(x, context=context, global_embedding=g_emb, time_embedding=t_emb,
                              cross_attention_kwargs=cross_attention_kwargs)
        x = self.output_blocks(x, context=context, global_embedding=g_emb, time_embedding=t_emb,
                               skip_connections=skips, cross_attention_kwargs=cross_attention_kwargs)
        return self.out[2](self.out[0](x))


class Denoiser(UNet):
    def __init__(self, config: DenoiserConfig):
        super().__init__(in_channels=config.in_channels, out_channels=config.out_channels, hidden_dim=config.hidden_dim,
                         dropout=0.0, conv_resample=config.conv_resample, num_head_channels=config.num_head_channels,
                         context_dim=config.context_dim, block_out_channels=config.block_out_channels,
                         num_transformers_per_block=config.num_transformers_per_block,
                         layers_per_block=config.layers_per_block, down_blocks=config.down_blocks,
                         mid_block=config.mid_block, up_blocks=config.up_blocks,
                         attn_implementation=config.attention_backend)
        self.config = config

    @property
    def device(self):
        return next(self.parameters()).device

    def set_gradient_checkpointing(self, gradient_checkpointing: bool):
        self.gradient_checkpointing = gradient_checkpointing
        for part in (self.input_blocks, self.middle_block, self.output_blocks):
            part.gradient_checkpointing = gradient_checkpointing
