"""L0 operator oracles (CPU): timestep embedding, RoPE tables/apply, norms, losses, samplers, LoRA.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference lines it
restates.  All of them are written with plain torch ops so that, run under
``torch.autocast("cpu", dtype=torch.bfloat16)`` as the reference trainer does
(src/trainer/common.py:301), they reproduce the reference's cast points.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------
# src/modules/timestep/embedding.py:10-62
# ------------------------------------------------------------------------------------------
def timestep_embedding(
    timesteps: torch.Tensor,
    dim: int,
    flip_sin_to_cos: bool = False,
    downscale_freq_shift: float = 1.0,
    scale: float = 1.0,
    max_period: int = 10000,
) -> torch.Tensor:
    assert timesteps.dim() == 1
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    ang = scale * (timesteps[:, None].float() * torch.exp(exponent)[None, :])
    sin, cos = torch.sin(ang), torch.cos(ang)
    emb = torch.cat([cos, sin], dim=-1) if flip_sin_to_cos else torch.cat([sin, cos], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


# src/models/auraflow/denoiser.py:500-514 (t*1000, cos||sin order)
def aura_timestep_embedding(timestep: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
    half = dim // 2
    freqs = 1000 * torch.exp(-math.log(max_period) * torch.arange(start=0, end=half) / half).to(timestep.device)
    args = timestep[:, None] * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


# ------------------------------------------------------------------------------------------
# src/modules/positional_encoding/rope.py
# ------------------------------------------------------------------------------------------
def image_position_indices(height: int, width: int, rope_dim: int = 3, y_index: int = 1, x_index: int = 2):
    """rope.py:5-31 — (height//2 * width//2, rope_dim) float positions, (0, y, x) per patch."""
    hp, wp = height // 2, width // 2
    pos = torch.zeros(hp, wp, rope_dim)
    pos[..., y_index] += torch.arange(hp)[:, None]
    pos[..., x_index] += torch.arange(wp)[None, :]
    return pos.reshape(-1, rope_dim)


def rope_frequencies(position_indices: torch.Tensor, dim_sizes: list[int], theta: float) -> torch.Tensor:
    """rope.py:35-79 — fp64 angles -> fp32 (S, sum(dim)/2, 2) [cos, sin]."""
    assert len(dim_sizes) == position_indices.shape[-1]
    parts = []
    for i, dim in enumerate(dim_sizes):
        assert dim % 2 == 0
        scale = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        omega = 1.0 / (theta**scale)
        ang = torch.outer(position_indices[..., i], omega)
        parts.append(torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1).float())
    return torch.cat(parts, dim=-2)


def apply_rope(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """rope.py:83-111 — interleaved-pair rotation in fp32, cast back. x: (B,H,S,D); freqs (S,D/2,2)."""
    dt = x.dtype
    x = x.float()
    cos, sin = freqs[..., 0], freqs[..., 1]
    xe, xo = x[..., 0::2], x[..., 1::2]
    out = torch.stack([xe * cos - xo * sin, xe * sin + xo * cos], dim=-1).reshape(*x.shape)
    return out.to(dt)


# Lumina2 complex RoPE, src/models/lumina2/denoiser.py:429-482 (table) and :103-112 (apply)
def lumina_freqs_table(axes_dims: list[int], axes_lens: list[int], theta: float = 10000.0) -> list[torch.Tensor]:
    out = []
    for d, e in zip(axes_dims, axes_lens):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = torch.outer(torch.arange(e, dtype=torch.float64), freqs).float()
        out.append(torch.polar(torch.ones_like(ang), ang).to(torch.complex64))
    return out


def lumina_freqs_cis(ids: torch.Tensor, table: list[torch.Tensor]) -> torch.Tensor:
    """ids (B,S,3) int -> complex64 (B,S,sum(d)/2)."""
    parts = [table[i][ids[:, :, i].to(torch.int64)] for i in range(len(table))]
    return torch.cat(parts, dim=-1)


def lumina_apply_rope(x_in: torch.Tensor, freqs_cis: torch.Tensor) -> torch.Tensor:
    """x_in (B,S,H,D); freqs_cis (B,S,D/2) complex. fp32 complex multiply, cast back."""
    x = torch.view_as_complex(x_in.float().reshape(*x_in.shape[:-1], -1, 2))
    out = torch.view_as_real(x * freqs_cis.unsqueeze(2)).flatten(3)
    return out.type_as(x_in)


# SDXL 2-D RoPE adapter, src/models/sdxl/adapter/rope.py
def sdxl_image_position_ids(height: int, width: int, origin_position: str = "top_left") -> torch.Tensor:
    """rope.py:78-115 — (H*W, 2) int64 (y, x); "center" subtracts ceil(h // 2) / ceil(w // 2) (sic)."""
    import math
    y = torch.arange(height, dtype=torch.int64).view(-1, 1).repeat(1, width)
    x = torch.arange(width, dtype=torch.int64).view(1, -1).repeat(height, 1)
    if origin_position == "center":
        y = y - math.ceil(height // 2)
        x = x - math.ceil(width // 2)
    return torch.stack([y, x], dim=-1).view(height * width, 2)


def sdxl_context_position_ids(length: int) -> torch.Tensor:
    """rope.py:117-126 — (i, i)"""
    return torch.arange(length, dtype=torch.int64).view(-1, 1).repeat(1, 2)


def sdxl_rope_freqs(position_ids: torch.Tensor, rope_dims: list[int], theta: float = 10000.0) -> torch.Tensor:
    """rope.py:21-38 and :128-139 — per axis: float64 theta, fp32 radians, unit-modulus complex64; axes concatenated."""
    parts = []
    for ids, dim in zip(position_ids.chunk(2, dim=-1), rope_dims, strict=True):
        assert dim % 2 == 0
        exponent = torch.arange(0, dim, 2, dtype=torch.float64) / dim
        radians = (ids * (1.0 / (theta**exponent))).float()
        parts.append(torch.polar(torch.ones_like(radians), radians).to(torch.complex64))
    return torch.cat(parts, dim=-1)


def sdxl_apply_rope(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """rope.py:41-54 — x (B,H,S,D), freqs complex (S,D/2): complex multiply of interleaved pairs in fp32, cast back."""
    xc = torch.view_as_complex(x.float().reshape(*x.shape[:-1], -1, 2))
    return torch.view_as_real(xc * freqs.view(1, 1, *freqs.shape)).flatten(3).to(x.dtype)


# ------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------
def fp32_layer_norm(x: torch.Tensor, dim: int, weight=None, bias=None, eps: float = 1e-5) -> torch.Tensor:
    """auraflow/denoiser.py:59-72 and modules/norm.py:9-17: LN in fp32, cast back."""
    w = None if weight is None else weight.float()
    b = None if bias is None else bias.float()
    return F.layer_norm(x.float(), (dim,), w, b, eps).to(x.dtype)


# ------------------------------------------------------------------------------------------
# losses / noising
# ------------------------------------------------------------------------------------------
def ddpm_alphas_cumprod(beta_start=0.00085, beta_end=0.012, n=1000) -> torch.Tensor:
    """modules/loss/diffusion.py:23-34 scaled-linear betas -> cumprod(1-beta), fp32."""
    betas = torch.linspace(beta_start**0.5, beta_end**0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


def ddpm_noise(latents: torch.Tensor, timestep: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """modules/loss/diffusion.py:36-53 with the random noise passed in (fixtures store it)."""
    ac = ddpm_alphas_cumprod().to(latents.device)
    a = (ac[timestep] ** 0.5).flatten()
    s = ((1 - ac[timestep]) ** 0.5).flatten()
    while a.dim() < latents.dim():
        a, s = a.unsqueeze(-1), s.unsqueeze(-1)
    return a * latents + s * noise


def ddpm_loss(noise: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    """modules/loss/diffusion.py:57-68"""
    return F.mse_loss(pred, noise, reduction="mean")


def flow_noise(latents: torch.Tensor, timestep: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """modules/loss/flow_match.py:43-57: (1-t) x + t eps."""
    t = timestep.view([latents.size(0), *([1] * (latents.dim() - 1))])
    return (1 - t) * latents + t * noise


def flow_loss(latents: torch.Tensor, noise: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    """modules/loss/flow_match.py:85-96"""
    return F.mse_loss(pred, noise - latents, reduction="mean")


def flow_scaled_noise(latents: torch.Tensor, timestep: torch.Tensor, noise: torch.Tensor, clean_at_zero: bool = False) -> torch.Tensor:
    """modules/loss/flow_match.py:60-74 with the (already scaled) noise passed in: t x + (1-t) eps, or the
    clean-at-zero orientation (1-t) x + t eps."""
    t = timestep.view([latents.size(0), *([1] * (latents.dim() - 1))])
    return (1 - t) * latents + t * noise if clean_at_zero else t * latents + (1 - t) * noise


def x0_to_velocity(x0: torch.Tensor, noisy: torch.Tensor, timestep: torch.Tensor, eps: float = 1e-5, clean_at_zero: bool = False):
    """modules/loss/flow_match.py:102-116"""
    t = timestep.view([x0.size(0), *([1] * (x0.dim() - 1))])
    return (noisy - x0) / t.clamp_min(eps) if clean_at_zero else (x0 - noisy) / (1 - t).clamp_min(eps)


# ------------------------------------------------------------------------------------------
# linear with optional LoRA — src/modules/peft/lora.py:92-104
# parameters live in a flat dict; a LoRA-wrapped Linear `name` stores its base under
# `name.linear.{weight,bias}` and the adapter under `name.lora_down.weight`, `name.lora_up.weight`,
# `name.alpha` (the reference's state-dict layout).
# ------------------------------------------------------------------------------------------
def linear(P: dict, name: str, x: torch.Tensor) -> torch.Tensor:
    if f"{name}.lora_down.weight" in P:
        base = F.linear(x, P[f"{name}.linear.weight"], P.get(f"{name}.linear.bias"))
        down = P[f"{name}.lora_down.weight"]
        up = P[f"{name}.lora_up.weight"]
        rank = down.shape[0]
        lo = F.linear(F.linear(x, down), up, P.get(f"{name}.lora_up.bias"))
        return base + lo * (P[f"{name}.alpha"] / rank)
    return F.linear(x, P[f"{name}.weight"], P.get(f"{name}.bias"))


def conv2d(P: dict, name: str, x: torch.Tensor, stride: int = 1, padding: int = 0) -> torch.Tensor:
    if f"{name}.lora_down.weight" in P:
        # LoRAConv2d, src/modules/peft/lora.py:167-309 (forward :245-258): down = the base convolution's kernel / stride /
        # padding onto `rank` channels, up = 1x1, branch scaled by alpha / rank; base keys move under `<name>.conv2d.`
        base = F.conv2d(x, P[f"{name}.conv2d.weight"], P.get(f"{name}.conv2d.bias"), stride=stride, padding=padding)
        down = P[f"{name}.lora_down.weight"]
        lo = F.conv2d(F.conv2d(x, down, None, stride=stride, padding=padding), P[f"{name}.lora_up.weight"],
                      P.get(f"{name}.lora_up.bias"))
        return base + lo * (P[f"{name}.alpha"] / down.shape[0])
    return F.conv2d(x, P[f"{name}.weight"], P.get(f"{name}.bias"), stride=stride, padding=padding)


# ------------------------------------------------------------------------------------------
# train-time timestep samplers as VALUE TRANSFORMS of a stored raw draw
# (src/modules/timestep/sampling.py:10-92,148-162,191-207; src/models/lumina2/scheduler.py:48-72).
# The reference draws with torch.randn / torch.rand / torch.randint / np.random.choice and then applies these.
# ------------------------------------------------------------------------------------------
def _lin(x, x1=256.0, y1=0.5, x2=4096.0, y2=1.15):
    slope = (y2 - y1) / (x2 - x1)
    return slope * x + (y1 - slope * x1)


def sampler_transform(name: str, draw: torch.Tensor, latents_shape=None, **kw) -> torch.Tensor:
    if name == "sigmoid_randn":  # :70-81
        return (draw * kw.get("sigmoid_scale", 1.0)).sigmoid()
    if name == "shift_sigmoid_randn":  # :50-67
        s = kw.get("discrete_flow_shift", 3.1825)
        t = (draw * kw.get("sigmoid_scale", 1.0)).sigmoid()
        return (t * s) / (1 + (s - 1) * t)
    if name == "flux_shift_randn":  # :34-47
        _b, _c, h, w = latents_shape
        t = (draw * kw.get("sigmoid_scale", 1.0)).sigmoid()
        mu = _lin((h // 2) * (w // 2))
        return math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** 1.0)
    if name == "scale_shift_sigmoid_randn":  # :259-272
        return (draw * kw.get("std", 0.8) + kw.get("mean", -0.8)).sigmoid()
    if name == "uniform_rand":  # :84-92
        return draw
    if name in ("shift_uniform_rand", "shift_fraction_uniform_rand"):  # :95-106, :148-162
        s = kw.get("shift", 6.0)
        return (draw * s) / (1 + (s - 1) * draw)
    if name == "uniform_randint":  # :191-207
        return draw.int()
    if name == "lumina_lognorm":  # lumina2/scheduler.py:48-72
        _b, _c, h, w = latents_shape
        ps = kw.get("patch_size", 2)
        mu = _lin((h // ps) * (w // ps))
        t = 1 - draw.sigmoid()
        t = math.exp(mu) / (math.exp(mu) + (1 / t - 1) ** kw.get("sigma", 1.0))
        return 1 - t
    raise ValueError(name)


def create_fraction(denominators):  # sampling.py:110-125 (set iteration order = insertion history of the floats)
    import numpy as np

    uniq = set()
    for d in denominators:
        for i in range(0, d + 1):
            uniq.add(i / d)
    return np.array(list(uniq), dtype=np.float32)
