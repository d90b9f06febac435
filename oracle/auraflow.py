"""AuraFlow MMDiT oracle (CPU, functional).  TEST INFRASTRUCTURE ONLY.

Restates src/models/auraflow/denoiser.py as a pure function of a flat parameter dict keyed like
the reference's state dict.  Run under ``torch.autocast("cpu", dtype=torch.bfloat16)``.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

from . import ops
from .attention import sdpa_ref as sdpa


@dataclass
class AuraCfg:
    """Mirror of DenoiserConfig, src/models/auraflow/config.py:8-30."""

    in_channels: int = 4
    out_channels: int = 4
    patch_size: int = 2
    caption_projection_dim: int = 3072
    num_double_layers: int = 4
    num_single_layers: int = 32
    num_attention_heads: int = 12
    attention_head_dim: int = 256
    joint_attention_dim: int = 2048
    pos_embed_max_size: int = 96 * 96
    num_register_tokens: int = 8
    use_rope: bool = False
    rope_theta: int = 10000
    rope_dim_sizes: list = field(default_factory=lambda: [32, 112, 112])

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def find_multiple(n: int, k: int) -> int:
    """denoiser.py:79-92"""
    return n if n % k == 0 else n + k - (n % k)


def mlp_hidden(dim: int) -> int:
    """AuraMLP hidden width, denoiser.py:101-105 (hidden_dim = 4*dim passed by the blocks)."""
    return find_multiple(int(2 * (4 * dim) / 3), 256)


def _ln(x, dim):  # Fp32LayerNorm(elementwise_affine=False), denoiser.py:59-72
    return ops.fp32_layer_norm(x, dim, eps=1e-5)


def modulate(x, shift, scale):  # denoiser.py:75-76
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def _mlp(P, pre, x):  # AuraMLP.forward, denoiser.py:113-119
    h1 = F.silu(ops.linear(P, f"{pre}.c_fc1", x))
    h2 = ops.linear(P, f"{pre}.c_fc2", x)
    return ops.linear(P, f"{pre}.c_proj", h1 * h2)


def _heads(x, n_heads, head_dim):
    b, s, _ = x.shape
    return x.view(b, s, n_heads, head_dim)


def _attend(q, k, v, rope, scale=None):
    """q,k,v (B,S,H,D): optional RoPE (denoiser.py:195-199) then sdpa_in_BNLD (:35-55)."""
    q, k, v = q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)
    if rope is not None:
        q, k = ops.apply_rope(q, rope), ops.apply_rope(k, rope)
    out = sdpa(q, k, v, scale=scale)
    return out.transpose(1, 2).flatten(-2)


def _single_attention(P, pre, x, cfg: AuraCfg, rope):
    """SingleAttention.forward, denoiser.py:179-214"""
    H, D = cfg.num_attention_heads, cfg.attention_head_dim
    q = _ln(_heads(ops.linear(P, f"{pre}.w1q", x), H, D), D)
    k = _ln(_heads(ops.linear(P, f"{pre}.w1k", x), H, D), D)
    v = _heads(ops.linear(P, f"{pre}.w1v", x), H, D)
    return ops.linear(P, f"{pre}.w1o", _attend(q, k, v, rope, scale=1 / D**0.5))


def _double_attention(P, pre, c, x, cfg: AuraCfg, rope):
    """DoubleAttention.forward, denoiser.py:269-333"""
    H, D = cfg.num_attention_heads, cfg.attention_head_dim
    cq = _ln(_heads(ops.linear(P, f"{pre}.w1q", c), H, D), D)
    ck = _ln(_heads(ops.linear(P, f"{pre}.w1k", c), H, D), D)
    cv = _heads(ops.linear(P, f"{pre}.w1v", c), H, D)
    xq = _ln(_heads(ops.linear(P, f"{pre}.w2q", x), H, D), D)
    xk = _ln(_heads(ops.linear(P, f"{pre}.w2k", x), H, D), D)
    xv = _heads(ops.linear(P, f"{pre}.w2v", x), H, D)
    q, k, v = torch.cat([cq, xq], 1), torch.cat([ck, xk], 1), torch.cat([cv, xv], 1)
    out = _attend(q, k, v, rope)
    co, xo = out.split([c.shape[1], x.shape[1]], dim=1)
    return ops.linear(P, f"{pre}.w1o", co), ops.linear(P, f"{pre}.w2o", xo)


def _mod6(P, name, g):
    return ops.linear(P, name, F.silu(g)).chunk(6, dim=1)


def _mmdit_block(P, pre, c, x, g, cfg, rope):
    """MMDiTBlock.forward, denoiser.py:372-424"""
    dim = cfg.inner_dim
    c_res, x_res = c, x
    c_sh1, c_sc1, c_g1, c_sh2, c_sc2, c_g2 = _mod6(P, f"{pre}.modC.1", g)
    x_sh1, x_sc1, x_g1, x_sh2, x_sc2, x_g2 = _mod6(P, f"{pre}.modX.1", g)
    c = modulate(_ln(c, dim), c_sh1, c_sc1)
    x = modulate(_ln(x, dim), x_sh1, x_sc1)
    c, x = _double_attention(P, f"{pre}.attn", c, x, cfg, rope)
    c = _ln(c_res + c_g1.unsqueeze(1) * c, dim)
    c = c_res + c_g2.unsqueeze(1) * _mlp(P, f"{pre}.mlpC", modulate(c, c_sh2, c_sc2))
    x = _ln(x_res + x_g1.unsqueeze(1) * x, dim)
    x = x_res + x_g2.unsqueeze(1) * _mlp(P, f"{pre}.mlpX", modulate(x, x_sh2, x_sc2))
    return c, x


def _dit_block(P, pre, x, g, cfg, rope):
    """DiTBlock.forward, denoiser.py:456-482"""
    dim = cfg.inner_dim
    res = x
    sh1, sc1, g1, sh2, sc2, g2 = _mod6(P, f"{pre}.modCX.1", g)
    x = modulate(_ln(x, dim), sh1, sc1)
    x = _single_attention(P, f"{pre}.attn", x, cfg, rope)
    x = _ln(res + g1.unsqueeze(1) * x, dim)
    return res + g2.unsqueeze(1) * _mlp(P, f"{pre}.mlp", modulate(x, sh2, sc2))


def pe_index(cfg: AuraCfg, h: int, w: int) -> torch.Tensor:
    """MMDiT.pe_selection_index_based_on_dim, denoiser.py:678-695 — centre crop of the PE grid."""
    hp, wp = h // cfg.patch_size, w // cfg.patch_size
    side = int(cfg.pos_embed_max_size**0.5)
    grid = torch.arange(cfg.pos_embed_max_size).view(side, side)
    sh = side // 2 - hp // 2
    sw = side // 2 - wp // 2
    return grid[sh : sh + hp, sw : sw + wp].flatten()


def patchify(x: torch.Tensor, p: int) -> torch.Tensor:
    """denoiser.py:734-761: (B,C,H,W) -> (B, H/p*W/p, C*p*p)."""
    b, c, h, w = x.shape
    x = x.view(b, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.flatten(-3).flatten(1, 2)


def unpatchify(x: torch.Tensor, hp: int, wp: int, p: int, c: int) -> torch.Tensor:
    """denoiser.py:702-732: (B, hp*wp, p*p*c) -> (B,c,hp*p,wp*p) via 'nhwpqc->nchpwq'."""
    b = x.shape[0]
    x = x.reshape(b, hp, wp, p, p, c)
    x = torch.einsum("nhwpqc->nchpwq", x)
    return x.reshape(b, c, hp * p, wp * p)


def mmdit_forward(P, cfg: AuraCfg, latent, encoder_hidden_states, timestep):
    """MMDiT.forward, denoiser.py:763-870 (no shortcut / guidance embedders)."""
    b, _c, h, w = latent.shape
    cond = ops.linear(P, "cond_seq_linear", encoder_hidden_states[:b])
    cond = torch.cat([P["register_tokens"].repeat(b, 1, 1), cond], dim=1)
    t_freq = ops.aura_timestep_embedding(timestep, 256)
    g = ops.linear(P, "t_embedder.mlp.2", F.silu(ops.linear(P, "t_embedder.mlp.0", t_freq)))
    x = ops.linear(P, "init_x_linear", patchify(latent, cfg.patch_size))
    if cfg.use_rope:
        n_axes = len(cfg.rope_dim_sizes)
        pos = torch.cat([torch.zeros(cond.size(1), n_axes), ops.image_position_indices(h, w, n_axes)], dim=0)
        rope = ops.rope_frequencies(pos, cfg.rope_dim_sizes, cfg.rope_theta)
    else:
        x = x + P["positional_encoding"][:, pe_index(cfg, h, w)]
        rope = None
    for i in range(cfg.num_double_layers):
        cond, x = _mmdit_block(P, f"double_layers.{i}", cond, x, g, cfg, rope)
    if cfg.num_single_layers > 0:
        n_cond = cond.size(1)
        ctx = torch.cat([cond, x], dim=1)
        for i in range(cfg.num_single_layers):
            ctx = _dit_block(P, f"single_layers.{i}", ctx, g, cfg, rope)
        x = ctx[:, n_cond:]
    f_shift, f_scale = ops.linear(P, "modF.1", F.silu(g)).chunk(2, dim=1)
    x = ops.linear(P, "final_linear", modulate(x, f_shift, f_scale))
    return unpatchify(x, h // cfg.patch_size, w // cfg.patch_size, cfg.patch_size, cfg.out_channels)
