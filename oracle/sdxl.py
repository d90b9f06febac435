"""SDXL UNet oracle (CPU, functional).  TEST INFRASTRUCTURE ONLY.

Restates src/models/sdxl/denoiser.py as a pure function of a flat parameter dict that uses the
reference's state-dict key names (so a reference ``Denoiser.state_dict()`` — with or without
LoRA-wrapped Linears — can be passed straight in).  Run under
``torch.autocast("cpu", dtype=torch.bfloat16)`` like the reference trainer.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from . import ops
from .attention import sdpa_ref as sdpa


@dataclass
class SDXLCfg:
    """Mirror of DenoiserConfig, src/models/sdxl/config.py:22-55 (field names kept)."""

    in_channels: int = 4
    out_channels: int = 4
    hidden_dim: int = 320
    channel_multipiler: list = field(default_factory=lambda: [1, 2, 4])
    conv_resample: bool = True
    num_head_channels: int = 64
    context_dim: int = 2048
    global_cond_dim: int = 2816
    additional_condition_dim: int = 256
    block_out_channels: list = field(default_factory=lambda: [320, 640, 1280])
    num_transformers_per_block: list = field(default_factory=lambda: [1, 2, 10])
    layers_per_block: int = 2
    down_blocks: list = field(default_factory=lambda: ["DownBlock2D", "TransformerDownBlock2D", "TransformerDownBlock2D"])
    mid_block: str = "TransformerMidBlock2D"
    up_blocks: list = field(default_factory=lambda: ["TransformerUpBlock2D", "TransformerUpBlock2D", "UpBlock2D"])


def _attention(P, pre, x, ctx, heads, head_dim, fq=None, fk=None):
    """SelfAttention / CrossAttention forward, denoiser.py:57-94 and :135-172; with `fq`/`fk` (complex RoPE tables for
    the query / key positions) the *WithRoPE variants of adapter/rope.py:196-316."""
    q = ops.linear(P, f"{pre}.to_q", x)
    k = ops.linear(P, f"{pre}.to_k", ctx)
    v = ops.linear(P, f"{pre}.to_v", ctx)
    b, s, _ = q.shape
    sk = k.shape[1]
    q = q.reshape(b, s, heads, head_dim).permute(0, 2, 1, 3)
    k = k.reshape(b, sk, heads, head_dim).permute(0, 2, 1, 3)
    v = v.reshape(b, sk, heads, head_dim).permute(0, 2, 1, 3)
    if fq is not None:
        q = ops.sdxl_apply_rope(q, fq)
        k = ops.sdxl_apply_rope(k, fk)
    a = sdpa(q, k, v)
    a = a.permute(0, 2, 1, 3).reshape(b, s, heads * head_dim)
    return ops.linear(P, f"{pre}.to_out.0", a)


def _transformer_block(P, pre, x, ctx, heads, head_dim, f_img=None, f_ctx=None):
    """TransformerBlock.forward, denoiser.py:254-280; GEGLU FF :183-204.  TransformerWithRoPE (adapter/rope.py:398-447)
    rotates attn1's q,k by the image table and attn2's q / k by the image / context tables."""
    c = x.shape[-1]
    h = F.layer_norm(x, (c,), P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"])
    x = x + _attention(P, f"{pre}.attn1", h, h, heads, head_dim, f_img, f_img)
    h = F.layer_norm(x, (c,), P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"])
    x = x + _attention(P, f"{pre}.attn2", h, ctx, heads, head_dim, f_img, f_ctx)
    h = F.layer_norm(x, (c,), P[f"{pre}.norm3.weight"], P[f"{pre}.norm3.bias"])
    hs, gate = ops.linear(P, f"{pre}.ff.net.0.proj", h).chunk(2, dim=-1)
    x = x + ops.linear(P, f"{pre}.ff.net.2", hs * F.gelu(gate))
    return x


def _spatial_transformer(P, pre, x, ctx, head_dim, n_blocks, rope=None):
    """SpatialTransformer.forward, denoiser.py:320-365 (GN eps 1e-6, NCHW<->NLC); it hands (height, width) to its
    blocks (:342-346), from which the RoPE blocks build their tables."""
    b, c, hh, ww = x.shape
    heads = c // head_dim
    f_img = f_ctx = None
    if rope is not None:
        f_img = ops.sdxl_rope_freqs(ops.sdxl_image_position_ids(hh, ww, rope["origin_position"]), rope["rope_dims"], rope["rope_theta"])
        f_ctx = ops.sdxl_rope_freqs(ops.sdxl_context_position_ids(ctx.shape[1]), rope["rope_dims"], rope["rope_theta"])
    res = x
    h = F.group_norm(x, 32, P[f"{pre}.norm.weight"], P[f"{pre}.norm.bias"], eps=1e-6)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    h = ops.linear(P, f"{pre}.proj_in", h)
    for i in range(n_blocks):
        h = _transformer_block(P, f"{pre}.transformer_blocks.{i}", h, ctx, heads, head_dim, f_img, f_ctx)
    h = ops.linear(P, f"{pre}.proj_out", h)
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return h + res


def _res_block(P, pre, x, emb):
    """ResidualBlock.forward, denoiser.py:559-586 (updown 'none' — the only kind the UNet builds)."""
    h = F.group_norm(x, 32, P[f"{pre}.in_layers.0.weight"], P[f"{pre}.in_layers.0.bias"], eps=1e-5)
    h = ops.conv2d(P, f"{pre}.in_layers.2", F.silu(h), padding=1)
    e = ops.linear(P, f"{pre}.emb_layers.1", F.silu(emb))
    h = h + e.reshape(e.shape[0], e.shape[1], 1, 1)
    h = F.group_norm(h, 32, P[f"{pre}.out_layers.0.weight"], P[f"{pre}.out_layers.0.bias"], eps=1e-5)
    h = ops.conv2d(P, f"{pre}.out_layers.3", F.silu(h), padding=1)
    if f"{pre}.skip_connection.weight" in P:
        x = ops.conv2d(P, f"{pre}.skip_connection", x)
    return h + x


def _mlp_embed(P, pre, x):
    """MLPEmbedder, denoiser.py:19-25"""
    return ops.linear(P, f"{pre}.2", F.silu(ops.linear(P, f"{pre}.0", x)))


def global_condition(P, cfg: SDXLCfg, timestep, pooled, original_size, target_size, crop, dtype):
    """UNet.prepare_global_condition, denoiser.py:1125-1178"""
    t_emb = ops.timestep_embedding(timestep, cfg.hidden_dim, flip_sin_to_cos=True, downscale_freq_shift=0.0)
    t_emb = _mlp_embed(P, "time_embed", t_emb)
    b = pooled.size(0)
    add = torch.cat([original_size, crop, target_size], dim=1).flatten()
    add = ops.timestep_embedding(add, cfg.additional_condition_dim, flip_sin_to_cos=True, downscale_freq_shift=0.0)
    add = add.reshape(b, -1)
    g = torch.cat([pooled, add], dim=1).to(dtype=dtype)
    g = _mlp_embed(P, "label_emb.0", g)
    return t_emb, g + t_emb


def unet_forward(P, cfg: SDXLCfg, latents, timestep, encoder_hidden_states, encoder_pooler_output,
                 original_size, target_size, crop_coords_top_left, rope: dict | None = None):
    """UNet.forward, denoiser.py:1180-1233; block wiring from DownBlocks/MidBlock/UpBlocks
    constructors (:609-722, :776-826, :860-956).  `rope` = dict(rope_dims, rope_theta, origin_position) runs
    DenoiserWithRoPE (adapter/rope.py:459-536) instead."""
    ctx = encoder_hidden_states
    _t, g = global_condition(P, cfg, timestep, encoder_pooler_output, original_size, target_size,
                             crop_coords_top_left, latents.dtype)
    hd = cfg.num_head_channels
    x = latents
    skips = []
    idx = 0
    nstage = len(cfg.down_blocks)
    # ---- down ----
    # the reference does not forward num_transformers_per_block to DownBlocks (denoiser.py:1060-1072):
    # the down path always runs the constructor default [1, 2, 10]
    for i, (kind, _ch, nt) in enumerate(zip(cfg.down_blocks, cfg.block_out_channels, [1, 2, 10])):
        if kind == "DownBlock2D":
            x = ops.conv2d(P, f"input_blocks.blocks.{idx}.0", x, padding=1)
            skips.append(x)
            idx += 1
            for _ in range(cfg.layers_per_block):
                x = _res_block(P, f"input_blocks.blocks.{idx}.0", x, g)
                skips.append(x)
                idx += 1
        else:
            for _ in range(cfg.layers_per_block):
                x = _res_block(P, f"input_blocks.blocks.{idx}.0", x, g)
                x = _spatial_transformer(P, f"input_blocks.blocks.{idx}.1", x, ctx, hd, nt, rope)
                skips.append(x)
                idx += 1
        if i != nstage - 1:
            x = ops.conv2d(P, f"input_blocks.blocks.{idx}.0.op", x, stride=2, padding=1)
            skips.append(x)
            idx += 1
    # ---- mid ----
    x = _res_block(P, "middle_block.blocks.0", x, g)
    x = _spatial_transformer(P, "middle_block.blocks.1", x, ctx, hd, cfg.num_transformers_per_block[-1], rope)
    x = _res_block(P, "middle_block.blocks.2", x, g)
    # ---- up ----
    idx = 0
    rev_nt = cfg.num_transformers_per_block[::-1]
    for i, (kind, nt) in enumerate(zip(cfg.up_blocks, rev_nt)):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            pre = f"output_blocks.blocks.{idx}"
            x = _res_block(P, f"{pre}.0", x, g)
            nxt = 1
            if kind == "TransformerUpBlock2D":
                x = _spatial_transformer(P, f"{pre}.1", x, ctx, hd, nt, rope)
                nxt = 2
            if j == cfg.layers_per_block and i != len(cfg.up_blocks) - 1:
                x = F.interpolate(x, scale_factor=2, mode="nearest")
                x = ops.conv2d(P, f"{pre}.{nxt}.conv", x, padding=1)
            idx += 1
    # ---- out ----
    x = F.group_norm(x, 32, P["out.0.weight"], P["out.0.bias"], eps=1e-5)
    return ops.conv2d(P, "out.2", F.silu(x), padding=1)
