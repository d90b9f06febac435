"""Lumina-Image-2.0 NextDiT oracle (CPU, functional).  TEST INFRASTRUCTURE ONLY.

Restates src/models/lumina2/denoiser.py as a pure function of a flat parameter dict keyed like the
reference's state dict, including the ragged-batch assembly (per-sample patchify, position ids,
padding, caption/image concatenation).  Run under ``torch.autocast("cpu", dtype=torch.bfloat16)``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from . import ops
from .attention import sdpa_ref as sdpa


@dataclass
class LuminaCfg:
    """Mirror of DenoiserConfig, src/models/lumina2/config.py:9-30."""

    in_channels: int = 16
    out_channels: int = 16
    hidden_dim: int = 2304
    caption_dim: int = 2304
    timestep_embed_dim: int = 256
    norm_eps: float = 1e-5
    depth: int = 26
    num_heads: int = 24
    num_kv_heads: int = 8
    refiner_depth: int = 2
    multiple_of: int = 256
    axes_dims: list = field(default_factory=lambda: [32, 32, 32])
    axes_lens: list = field(default_factory=lambda: [300, 512, 512])
    theta: int = 10_000
    patch_size: int = 2

    @property
    def head_dim(self) -> int:
        return self.hidden_dim // self.num_heads


def ffn_dim(cfg: LuminaCfg) -> int:
    """FeedForward.intermediate_dim, denoiser.py:186-188 (intermediate = 4*hidden)."""
    m = cfg.multiple_of
    return m * ((cfg.hidden_dim * 4 + m - 1) // m)


def _rms(P, name, x, eps):
    return F.rms_norm(x, (x.shape[-1],), P[f"{name}.weight"], eps)


def _self_attention(P, pre, x, freqs_cis, mask, cfg: LuminaCfg):
    """SelfAttention.forward, denoiser.py:114-171"""
    b, s, _ = x.shape
    H, Hkv, D = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    q, k, v = torch.split(ops.linear(P, f"{pre}.qkv", x), [H * D, Hkv * D, Hkv * D], dim=-1)
    q = q.view(b, s, H, D)
    k = k.view(b, s, Hkv, D)
    v = v.view(b, s, Hkv, D)
    q = F.rms_norm(q, (D,), P[f"{pre}.q_norm.weight"], 1e-6)
    k = F.rms_norm(k, (D,), P[f"{pre}.k_norm.weight"], 1e-6)
    q = ops.lumina_apply_rope(q, freqs_cis)
    k = ops.lumina_apply_rope(k, freqs_cis)
    rep = H // Hkv
    k = k.unsqueeze(3).repeat(1, 1, 1, rep, 1).flatten(2, 3)
    v = v.unsqueeze(3).repeat(1, 1, 1, rep, 1).flatten(2, 3)
    m4 = None
    if mask is not None:
        m4 = mask.bool().view(b, 1, 1, s).expand(-1, H, s, -1)
    out = sdpa(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), mask=m4,
               scale=math.sqrt(1 / D))
    return ops.linear(P, f"{pre}.out", out.permute(0, 2, 1, 3).flatten(-2))


def _feed_forward(P, pre, x):
    """FeedForward.forward, denoiser.py:212-219"""
    gate = ops.linear(P, f"{pre}.w3", x)
    h = ops.linear(P, f"{pre}.w1", x)
    return ops.linear(P, f"{pre}.w2", F.silu(h) * gate)


def _block(P, pre, x, freqs_cis, mask, cfg: LuminaCfg, adaln=None):
    """TransformerBlock.forward_with_adaln (:283-313) / forward_without_adaln (:315-338)."""
    eps = cfg.norm_eps
    if adaln is not None:
        mod = ops.linear(P, f"{pre}.adaLN_modulation.1", F.silu(adaln))
        sc_a, g_a, sc_m, g_m = mod.chunk(4, dim=1)
        a = _rms(P, f"{pre}.attention_norm1", x, eps) * (1 + sc_a.unsqueeze(1))
        a = _rms(P, f"{pre}.attention_norm2", _self_attention(P, f"{pre}.attention", a, freqs_cis, mask, cfg), eps)
        x = x + g_a.unsqueeze(1).tanh() * a
        m = _rms(P, f"{pre}.ffn_norm1", x, eps) * (1 + sc_m.unsqueeze(1))
        m = _rms(P, f"{pre}.ffn_norm2", _feed_forward(P, f"{pre}.feed_forward", m), eps)
        return x + g_m.unsqueeze(1).tanh() * m
    a = _rms(P, f"{pre}.attention_norm1", x, eps)
    a = _rms(P, f"{pre}.attention_norm2", _self_attention(P, f"{pre}.attention", a, freqs_cis, mask, cfg), eps)
    x = x + a
    m = _rms(P, f"{pre}.ffn_norm1", x, eps)
    m = _rms(P, f"{pre}.ffn_norm2", _feed_forward(P, f"{pre}.feed_forward", m), eps)
    return x + m


def patchify(image: torch.Tensor, p: int) -> torch.Tensor:
    """NextDiT.patchify, denoiser.py:609-631: (c,H,W) -> (H/p*W/p, p*p*c) ordered (ph, pw, c)."""
    c, h, w = image.shape
    x = image.view(c, h // p, p, w // p, p).permute(1, 3, 2, 4, 0)
    return x.flatten(2).flatten(0, 1)


def unpatchify(patches: torch.Tensor, h: int, w: int, p: int, c: int) -> torch.Tensor:
    """NextDiT.unpatchify, denoiser.py:633-653."""
    x = patches.reshape(h // p, w // p, p, p, c)
    return x.permute(4, 0, 2, 1, 3).flatten(3, 4).flatten(1, 2)


def position_ids(caption_length: int, ph: int, pw: int) -> torch.Tensor:
    """NextDiT.get_position_ids, denoiser.py:673-721 — (cap+ph*pw, 3) int32."""
    n = ph * pw
    ids = torch.zeros(caption_length + n, 3, dtype=torch.int32)
    ids[:caption_length, 0] = torch.arange(caption_length, dtype=torch.int32)
    ids[caption_length:, 0] = caption_length
    ids[caption_length:, 1] = torch.arange(ph, dtype=torch.int32).view(-1, 1).repeat(1, pw).flatten()
    ids[caption_length:, 2] = torch.arange(pw, dtype=torch.int32).view(1, -1).repeat(ph, 1).flatten()
    return ids


def pad_stack(items: list[torch.Tensor]) -> tuple[torch.Tensor, torch.Tensor]:
    """NextDiT._pad, denoiser.py:809-846: zero-pad along dim 0, stack, prefix bool mask."""
    mx = max(t.size(0) for t in items)
    mask = torch.zeros(len(items), mx, dtype=torch.bool)
    out = []
    for i, t in enumerate(items):
        mask[i, : t.size(0)] = True
        if t.size(0) < mx:
            pad = [0, 0] * (t.dim() - 1) + [0, mx - t.size(0)]
            t = F.pad(t, pad, value=0.0) if t.dtype != torch.bool else F.pad(t, pad, value=False)
        out.append(t)
    return torch.stack(out, 0), mask


def nextdit_forward(P, cfg: LuminaCfg, latents, caption_features, timestep, caption_mask):
    """NextDiT.forward, denoiser.py:949-1082 (no cached caption features).

    latents: list of (C,H,W) (or a (B,C,H,W) tensor); caption_features (B,L,Cdim); caption_mask
    (B,L) bool prefix mask.  Returns (list of (C,H,W) velocity predictions, caption_mask,
    refined caption features).
    """
    p = cfg.patch_size
    b = len(latents)
    caption_mask = caption_mask.bool()
    cap_lens = [int(m.sum().item()) for m in caption_mask]
    # 1. timestep embedding (TimestepEmbedder.forward :45-62, MLP input forced to fp32)
    t_emb = ops.timestep_embedding(timestep, cfg.timestep_embed_dim, flip_sin_to_cos=True, downscale_freq_shift=0)
    t = ops.linear(P, "t_embedder.mlp.2", F.silu(ops.linear(P, "t_embedder.mlp.0", t_emb.float())))
    # 2. patchify + position ids
    patches, sizes, ids = [], [], []
    for cl, img in zip(cap_lens, latents):
        h, w = img.shape[-2:]
        patches.append(patchify(img, p))
        sizes.append((h, w))
        ids.append(position_ids(cl, h // p, w // p))
    pos, overall_mask = pad_stack(ids)
    table = ops.lumina_freqs_table(cfg.axes_dims, cfg.axes_lens, float(cfg.theta))
    freqs = ops.lumina_freqs_cis(pos, table)
    img_pos_mask = overall_mask.clone()
    for i, cl in enumerate(cap_lens):
        img_pos_mask[i, :cl] = False
    # 4. caption refiner
    cap, _ = pad_stack([f[m] for f, m in zip(caption_features, caption_mask)])
    cap_freqs, cap_mask = pad_stack([freqs[i, : cap_lens[i]] for i in range(b)])
    cap = ops.linear(P, "cap_embedder.1", _rms(P, "cap_embedder.0", cap, cfg.norm_eps))
    for i in range(cfg.refiner_depth):
        cap = _block(P, f"context_refiner.{i}", cap, cap_freqs, cap_mask, cfg)
    # 5. noise refiner
    img, img_mask = pad_stack(patches)
    img_freqs, _ = pad_stack([freqs[i, m] for i, m in enumerate(img_pos_mask)])
    x = ops.linear(P, "x_embedder", img)
    for i in range(cfg.refiner_depth):
        x = _block(P, f"noise_refiner.{i}", x, img_freqs, img_mask, cfg, adaln=t)
    # 5.5 concat (concat_context :905-932)
    img_lens = [pt.size(0) for pt in patches]
    mx = max(c + n for c, n in zip(cap_lens, img_lens))
    ctx = torch.zeros(b, mx, cap.shape[-1], dtype=cap.dtype)
    for i in range(b):
        ctx[i, : cap_lens[i]] = cap[i, : cap_lens[i]]
        ctx[i, cap_lens[i] : cap_lens[i] + img_lens[i]] = x[i, : img_lens[i]]
    # 6. main layers
    for i in range(cfg.depth):
        ctx = _block(P, f"layers.{i}", ctx, freqs, overall_mask, cfg, adaln=t)
    # 7. final layer (FinalLayer.forward :411-426)
    scale = ops.linear(P, "final_layer.adaLN_modulation.1", F.silu(t))
    y = ops.fp32_layer_norm(ctx, cfg.hidden_dim, eps=1e-6) * (1 + scale.unsqueeze(1))
    y = ops.linear(P, "final_layer.linear", y)
    outs = []
    for i in range(b):
        h, w = sizes[i]
        outs.append(unpatchify(y[i][img_pos_mask[i]], h, w, p, cfg.out_channels))
    return outs, cap_mask, cap
