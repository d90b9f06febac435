"""AuraFlow v0.3 MMDiT denoiser for MI355X.

Operator surface of the reference's src/models/auraflow/denoiser.py (class names, constructor
arguments, parameter names/shapes, ``forward(latent, encoder_hidden_states, timestep, ...)``
:763-771).  Execution differences:

* q/k projections go through one fused HIP kernel doing the per-head fp32 LayerNorm (no affine,
  head_dim 256, reference :165-174) and — with ``use_rope`` — the interleaved-pair rotation
  (modules/positional_encoding/rope.py:83-111) in the same pass;
* attention is the HIP flash kernel on (B,S,H,D)-strided views (no BNLD<->BLND transposes, :35-55);
* AdaLN modulate (:75-76) is fused with the fp32 LayerNorm; gate*branch + residual is one kernel;
* SwiGLU (:113-119) is a fused kernel between the two library GEMMs.
"""
from __future__ import annotations

import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from ... import kernels
from ...modules.attention import scaled_dot_product_attention
from ...modules.positional_encoding.rope import RoPEFrequency
from ...utils.tensor import module_dtype
from .config import DenoiserConfig

DENOISER_TENSOR_PREFIX = "model."
_FREQ_CACHE: dict = {}

_ACT = {"silu": nn.SiLU, "gelu": nn.GELU, "relu": nn.ReLU}


def _act(name: str) -> nn.Module:
    try:
        return _ACT[name]()
    except KeyError:
        raise ValueError(f"unsupported hidden_act {name!r}") from None


class Fp32LayerNorm(nn.LayerNorm):
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return kernels.layer_norm(input, self.weight, self.bias, self.eps)


def modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def find_multiple(n: int, k: int) -> int:
    return n if n % k == 0 else n + k - (n % k)


class AuraMLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int | None = None, hidden_act: str = "silu") -> None:
        super().__init__()
        if hidden_dim is None:
            hidden_dim = 4 * input_dim
        n_hidden = find_multiple(int(2 * hidden_dim / 3), 256)
        self.c_fc1 = nn.Linear(input_dim, n_hidden, bias=False)
        self.c_fc2 = nn.Linear(input_dim, n_hidden, bias=False)
        self.c_proj = nn.Linear(n_hidden, input_dim, bias=False)
        self.hidden_act = hidden_act
        self.act = _act(hidden_act)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        a, b = _project_all(hidden_states, (self.c_fc1, self.c_fc2))
        gated = kernels.swiglu(a, b) if self.hidden_act == "silu" else self.act(a) * b
        return _proj(self.c_proj, gated)


class MultiHeadLayerNorm(nn.Module):
    """Per-head LayerNorm with a (heads, head_dim) weight (reference :123-141); unused by the shipped
    configs (mh_qknorm=False) — kept for interface parity."""

    def __init__(self, hidden_size: tuple[int, ...], eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        h = kernels.layer_norm(hidden_states, None, None, self.variance_epsilon)
        return (self.weight.float() * h.float()).to(hidden_states.dtype)


def _proj(layer: nn.Module, h: torch.Tensor) -> torch.Tensor:
    """layer(h); a plain frozen bf16 nn.Linear (modulation / embedder projections a LoRA run does not adapt) runs on the own
    projection GEMM (kernels.linear) instead of the library's: no stream-K / split-K launches with atomics left in the step,
    and the same kernels eagerly and under graph capture.  Adapter / quantised wrappers and trainable layers keep their forward."""
    if type(layer) is nn.Linear and h.is_cuda and kernels.linear_eligible(h, layer.weight):
        return kernels.linear(h, layer.weight, layer.bias)
    return layer(h)


def _seq(seq: nn.Sequential, h: torch.Tensor) -> torch.Tensor:
    """nn.Sequential.forward with its Linear members through `_proj` (activation -> Linear modulations, the timestep MLP)."""
    for m in seq:
        h = _proj(m, h)
    return h


def _project_all(x: torch.Tensor, layers) -> list[torch.Tensor]:
    """[layer(x) for layer in layers]; LoRA-wrapped siblings run as one autograd node whose data gradients are chained
    through the GEMM epilogue (kernels.lora_siblings) instead of being summed by separate elementwise passes."""
    outs = kernels.lora_siblings(x, layers) if x.is_cuda else None
    return outs if outs is not None else [_proj(layer, x) for layer in layers]


def _qk_norm(norm: nn.Module, x: torch.Tensor, rope: tuple[torch.Tensor, torch.Tensor] | None) -> torch.Tensor:
    """x (B,S,H,D) -> per-head norm (+ RoPE) fused."""
    if isinstance(norm, MultiHeadLayerNorm):
        h = norm(x)
        return h if rope is None else kernels.qk_norm_rope(h, "none", None, 0.0, rope[0], rope[1])
    cos, sin = rope if rope is not None else (None, None)
    return kernels.qk_norm_rope(x, "layer", None, norm.eps, cos, sin)


def _attend(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float | None) -> torch.Tensor:
    # (B,S,H,D) tensors; the kernel takes (B,H,S,D) *views* and writes (B,S,H,D) memory
    out = scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), scale=scale)
    return out.transpose(1, 2).flatten(-2)


def _rope_tables(rope_freqs: torch.Tensor | None, device) -> tuple[torch.Tensor, torch.Tensor] | None:
    if not isinstance(rope_freqs, torch.Tensor):
        return None
    f = rope_freqs.to(device)
    return f[..., 0].contiguous(), f[..., 1].contiguous()


class SingleAttention(nn.Module):
    def __init__(self, dim: int, n_heads: int, mh_qknorm: bool = False, use_flash_attn: bool = False, use_rope: bool = False):
        super().__init__()
        self.n_heads = n_heads
        self.head_dim = dim // n_heads
        self.use_flash_attn = use_flash_attn
        self.w1q = nn.Linear(dim, dim, bias=False)
        self.w1k = nn.Linear(dim, dim, bias=False)
        self.w1v = nn.Linear(dim, dim, bias=False)
        self.w1o = nn.Linear(dim, dim, bias=False)
        mk = (lambda: MultiHeadLayerNorm((n_heads, self.head_dim))) if mh_qknorm else (
            lambda: Fp32LayerNorm(self.head_dim, bias=False, elementwise_affine=False))
        self.q_norm1 = mk()
        self.k_norm1 = mk()
        self.use_rope = use_rope

    def forward(self, condition: torch.Tensor, rope_freqs: torch.Tensor | None = None) -> torch.Tensor:
        b, s, _ = condition.shape
        shape = (b, s, self.n_heads, self.head_dim)
        rope = _rope_tables(rope_freqs, condition.device)
        pq, pk, pv = _project_all(condition, (self.w1q, self.w1k, self.w1v))
        q = _qk_norm(self.q_norm1, pq.view(shape), rope)
        k = _qk_norm(self.k_norm1, pk.view(shape), rope)
        v = pv.view(shape)
        return _proj(self.w1o, _attend(q, k, v, 1 / self.head_dim**0.5))


class DoubleAttention(nn.Module):
    def __init__(self, dim: int, n_heads: int, mh_qknorm: bool = False, use_flash_attn: bool = False, use_rope: bool = False):
        super().__init__()
        self.n_heads = n_heads
        self.head_dim = dim // n_heads
        self.use_flash_attn = use_flash_attn
        for stream in ("1", "2"):  # 1 = condition tokens, 2 = image tokens
            for proj in "qkvo":
                setattr(self, f"w{stream}{proj}", nn.Linear(dim, dim, bias=False))
        mk = (lambda: MultiHeadLayerNorm((n_heads, self.head_dim))) if mh_qknorm else (
            lambda: Fp32LayerNorm(self.head_dim, bias=False, elementwise_affine=False))
        self.q_norm1 = mk()
        self.k_norm1 = mk()
        self.q_norm2 = mk()
        self.k_norm2 = mk()
        self.use_rope = use_rope

    def forward(self, condition: torch.Tensor, latent: torch.Tensor, rope_freqs: torch.Tensor | None = None):
        assert condition.shape[0] == latent.shape[0], "batch size must be the same"
        b, sc, _ = condition.shape
        sl = latent.shape[1]
        H, D = self.n_heads, self.head_dim
        rope = _rope_tables(rope_freqs, latent.device)
        rope_c = rope_l = None
        if rope is not None:
            rope_c = (rope[0][:sc], rope[1][:sc])
            rope_l = (rope[0][sc:], rope[1][sc:])
        q1, k1, v1 = _project_all(condition, (self.w1q, self.w1k, self.w1v))
        q2, k2, v2 = _project_all(latent, (self.w2q, self.w2k, self.w2v))
        q = torch.cat([_qk_norm(self.q_norm1, q1.view(b, sc, H, D), rope_c),
                       _qk_norm(self.q_norm2, q2.view(b, sl, H, D), rope_l)], dim=1)
        k = torch.cat([_qk_norm(self.k_norm1, k1.view(b, sc, H, D), rope_c),
                       _qk_norm(self.k_norm2, k2.view(b, sl, H, D), rope_l)], dim=1)
        v = torch.cat([v1.view(b, sc, H, D), v2.view(b, sl, H, D)], dim=1)
        out = _attend(q, k, v, None)
        c_out, l_out = out.split([sc, sl], dim=1)
        return _proj(self.w1o, c_out), _proj(self.w2o, l_out)


BLOCK_SKIPS = os.environ.get("VFT_BLOCK_SKIPS", "1") != "0"  # A/B switch


def _norm_with_skips(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, eps: float):
    """(AdaLN(x), x, x): a block's input feeds its first norm and both of its gated adds (`x + g1*att`, `x + g2*mlp`,
    src/models/auraflow/denoiser.py:392-422 of the reference); taking the two residual inputs from the norm's extra outputs
    lets the three gradients of x meet inside the norm-backward kernel instead of in two accumulation passes."""
    if BLOCK_SKIPS:
        return kernels.layer_norm_modulate(x, shift, scale, eps, skips=2)
    return kernels.layer_norm_modulate(x, shift, scale, eps), x, x


class MMDiTBlock(nn.Module):
    def __init__(self, dim: int, heads: int = 8, hidden_act: str = "silu", use_flash_attn: bool = False, use_rope: bool = False):
        super().__init__()
        ln = lambda: Fp32LayerNorm(dim, elementwise_affine=False, bias=False)  # noqa: E731
        self.normC1, self.normC2 = ln(), ln()
        self.mlpC = AuraMLP(dim, hidden_dim=dim * 4)
        self.modC = nn.Sequential(_act(hidden_act), nn.Linear(dim, 6 * dim, bias=False))
        self.normX1, self.normX2 = ln(), ln()
        self.mlpX = AuraMLP(dim, hidden_dim=dim * 4)
        self.modX = nn.Sequential(_act(hidden_act), nn.Linear(dim, 6 * dim, bias=False))
        self.attn = DoubleAttention(dim, heads, use_flash_attn=use_flash_attn, use_rope=use_rope)

    def forward(self, condition, patches, global_cond, rope_freqs=None, **kwargs):
        c_sh1, c_sc1, c_g1, c_sh2, c_sc2, c_g2 = _seq(self.modC, global_cond).chunk(6, dim=1)
        x_sh1, x_sc1, x_g1, x_sh2, x_sc2, x_g2 = _seq(self.modX, global_cond).chunk(6, dim=1)
        eps = self.normC1.eps
        c_n1, c_skip1, c_skip2 = _norm_with_skips(condition, c_sh1, c_sc1, eps)
        x_n1, x_skip1, x_skip2 = _norm_with_skips(patches, x_sh1, x_sc1, eps)
        c_att, x_att = self.attn(c_n1, x_n1, rope_freqs)
        # stream C
        c_mid = kernels.gate_residual(c_skip1, c_g1, c_att)
        c_mlp = self.mlpC(kernels.layer_norm_modulate(c_mid, c_sh2, c_sc2, eps))
        condition = kernels.gate_residual(c_skip2, c_g2, c_mlp)
        # stream X
        x_mid = kernels.gate_residual(x_skip1, x_g1, x_att)
        x_mlp = self.mlpX(kernels.layer_norm_modulate(x_mid, x_sh2, x_sc2, eps))
        patches = kernels.gate_residual(x_skip2, x_g2, x_mlp)
        return condition, patches


class DiTBlock(nn.Module):
    def __init__(self, dim: int, heads: int = 8, hidden_act: str = "silu", use_flash_attn: bool = False, use_rope: bool = False):
        super().__init__()
        self.norm1 = Fp32LayerNorm(dim, elementwise_affine=False, bias=False)
        self.norm2 = Fp32LayerNorm(dim, elementwise_affine=False, bias=False)
        self.modCX = nn.Sequential(_act(hidden_act), nn.Linear(dim, 6 * dim, bias=False))
        self.attn = SingleAttention(dim, heads, use_flash_attn=use_flash_attn, use_rope=use_rope)
        self.mlp = AuraMLP(dim, hidden_dim=dim * 4)

    def forward(self, context, global_cond, rope_freqs=None, **kwargs):
        sh1, sc1, g1, sh2, sc2, g2 = _seq(self.modCX, global_cond).chunk(6, dim=1)
        eps = self.norm1.eps
        # the block input feeds norm1 and BOTH gated adds: its three gradients meet inside norm1's backward kernel
        n1, skip1, skip2 = _norm_with_skips(context, sh1, sc1, eps)
        att = self.attn(n1, rope_freqs)
        mid = kernels.gate_residual(skip1, g1, att)
        mlp = self.mlp(kernels.layer_norm_modulate(mid, sh2, sc2, eps))
        return kernels.gate_residual(skip2, g2, mlp)


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_size: int, frequency_embedding_size: int = 256, hidden_act: str = "silu"):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), _act(hidden_act),
                                 nn.Linear(hidden_size, hidden_size))
        self.frequency_embedding_size = frequency_embedding_size

    @staticmethod
    def timestep_embedding(timestep: torch.Tensor, dim: int, max_period: int = 10000) -> torch.Tensor:
        half = dim // 2
        key = (dim, max_period, str(timestep.device))
        freqs = _FREQ_CACHE.get(key)
        if freqs is None:  # computed on the host exactly like the reference, then cached per device (graph-safe)
            freqs = (1000 * torch.exp(-math.log(max_period) * torch.arange(start=0, end=half) / half)).to(timestep.device)
            _FREQ_CACHE[key] = freqs
        args = timestep[:, None] * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, timestep: torch.Tensor) -> torch.Tensor:
        freq = self.timestep_embedding(timestep.float(), self.frequency_embedding_size)
        return _seq(self.mlp, freq.to(module_dtype(self.mlp[0])))


class MMDiT(nn.Module):
    def __init__(self, in_channels: int = 4, out_channels: int = 4, patch_size: int = 2, num_double_layers: int = 4,
                 num_single_layers: int = 8, num_heads: int = 4, attention_head_dim: int = 256,
                 caption_projection_dim: int = 3072, joint_attention_dim: int = 2048, max_pos_embed_size: int = 96 * 96,
                 n_register_tokens: int = 8, hidden_act: str = "silu", use_flash_attn: bool = False,
                 use_rope: bool = False, rope_dim_sizes: list[int] = [32, 112, 112], rope_theta: int = 10000,
                 use_shortcut: bool = False, use_guidance: bool = False):
        super().__init__()
        self.attention_head_dim = attention_head_dim
        self.num_attention_heads = num_heads
        self.inner_dim = attention_head_dim * num_heads

        self.t_embedder = TimestepEmbedder(self.inner_dim)
        self.cond_seq_linear = nn.Linear(joint_attention_dim, caption_projection_dim, bias=False)
        self.init_x_linear = nn.Linear(patch_size * patch_size * in_channels, self.inner_dim)
        self.rope_frequency = RoPEFrequency(dim_sizes=rope_dim_sizes, theta=rope_theta) if use_rope else None
        self.shortcut_embedder = TimestepEmbedder(self.inner_dim) if use_shortcut else None
        self.guidance_embedder = TimestepEmbedder(self.inner_dim) if use_guidance else None
        self.positional_encoding = nn.Parameter(torch.randn(1, max_pos_embed_size, self.inner_dim) * 0.1)
        self.register_tokens = nn.Parameter(torch.randn(1, n_register_tokens, self.inner_dim) * 0.02)

        blk = dict(hidden_act=hidden_act, use_flash_attn=use_flash_attn, use_rope=use_rope)
        self.double_layers = nn.ModuleList([MMDiTBlock(self.inner_dim, num_heads, **blk) for _ in range(num_double_layers)])
        self.single_layers = nn.ModuleList([DiTBlock(self.inner_dim, num_heads, **blk) for _ in range(num_single_layers)])
        self.final_linear = nn.Linear(self.inner_dim, patch_size * patch_size * out_channels, bias=False)
        self.modF = nn.Sequential(_act(hidden_act), nn.Linear(self.inner_dim, 2 * self.inner_dim, bias=False))

        self.out_channels = out_channels
        self.patch_size = patch_size
        self.num_double_layers = num_double_layers
        self.num_single_layers = num_single_layers
        self.num_layers = num_double_layers + num_single_layers
        self.max_pos_embed_size = max_pos_embed_size
        self.use_flash_attn = use_flash_attn
        self.use_rope = use_rope
        self.use_shortcut = use_shortcut
        self.use_guidance = use_guidance
        self.gradient_checkpointing = False
        self.h_max = self.w_max = int(max_pos_embed_size**0.5)
        self._pe_cache: dict = {}
        self._rope_cache: dict = {}
        self.init_weights()

    def init_weights(self):
        if self.final_linear.weight.is_meta:
            return
        nn.init.constant_(self.final_linear.weight, 0)
        for name, p in self.named_parameters():
            if ".mod" in name:
                nn.init.constant_(p, 0)
        nn.init.constant_(self.cond_seq_linear.weight, 0)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def _set_gradient_checkpointing(self, value: bool):
        self.gradient_checkpointing = value

    def pe_selection_index_based_on_dim(self, h: int, w: int) -> torch.Tensor:
        """Indices of the centred (h/p, w/p) window of the square positional-embedding grid."""
        hp, wp = h // self.patch_size, w // self.patch_size
        top = self.h_max // 2 - hp // 2
        left = self.w_max // 2 - wp // 2
        rows = torch.arange(top, top + hp).view(hp, 1) * self.w_max
        cols = torch.arange(left, left + wp).view(1, wp)
        return (rows + cols).flatten()

    def get_pos_encoding(self, h: int, w: int) -> torch.Tensor:
        key = (h, w, str(self.positional_encoding.device))
        idx = self._pe_cache.get(key)
        if idx is None:
            idx = self.pe_selection_index_based_on_dim(h, w).to(self.positional_encoding.device)
            self._pe_cache[key] = idx
        return self.positional_encoding[:, idx]

    def unpatchify(self, patches: torch.Tensor, height: int, width: int) -> torch.Tensor:
        b, p, c = patches.shape[0], self.patch_size, self.out_channels
        grid = patches.reshape(b, height, width, p, p, c).permute(0, 5, 1, 3, 2, 4)  # n c h p w q
        return grid.reshape(b, c, height * p, width * p)

    def patchify(self, image: torch.Tensor) -> torch.Tensor:
        b, c, h, w = image.shape
        p = self.patch_size
        tiles = image.reshape(b, c, h // p, p, w // p, p).permute(0, 2, 4, 1, 3, 5)  # n h w c p q
        return tiles.reshape(b, (h // p) * (w // p), c * p * p)

    def _rope_for(self, n_cond: int, height: int, width: int, device) -> torch.Tensor:
        key = (n_cond, height, width, str(device))
        tab = self._rope_cache.get(key)
        if tab is None:
            rf = self.rope_frequency
            idx = torch.cat([rf.get_text_position_indices(n_cond), rf.get_image_position_indices(height, width)], dim=0)
            tab = rf(idx).to(device)
            self._rope_cache[key] = tab
        return tab

    def forward(self, latent: torch.Tensor, encoder_hidden_states: torch.Tensor, timestep: torch.Tensor,
                shortcut_duration: int | None = None, guidance_scale: float | None = None, **kwargs) -> torch.Tensor:
        b, _c, height, width = latent.shape
        dtype = module_dtype(self.init_x_linear)
        cond = _proj(self.cond_seq_linear, encoder_hidden_states[:b].to(dtype))
        cond = torch.cat([self.register_tokens.expand(b, -1, -1).to(dtype), cond], dim=1)

        g = self.t_embedder(timestep)
        if shortcut_duration is not None:
            assert self.shortcut_embedder is not None, "shortcut_duration is provided but shortcut_embedder is not set"
            g = g + self.shortcut_embedder(shortcut_duration)
        if guidance_scale is not None:
            assert self.guidance_embedder is not None, "guidance_scale is provided but guidance_embedder is not set"
            g = g + self.guidance_embedder(timestep)

        x = _proj(self.init_x_linear, self.patchify(latent.to(dtype)))
        if self.rope_frequency is not None:
            rope = self._rope_for(cond.size(1), height, width, latent.device)
        else:
            x = x + self.get_pos_encoding(height, width).to(dtype)
            rope = None

        ckpt = self.gradient_checkpointing and torch.is_grad_enabled()
        for layer in self.double_layers:
            if ckpt:
                cond, x = checkpoint.checkpoint(layer, cond, x, g, rope, use_reentrant=False)
            else:
                cond, x = layer(cond, x, g, rope, **kwargs)
        if len(self.single_layers) > 0:
            n_cond = cond.size(1)
            ctx = torch.cat([cond, x], dim=1)
            for layer in self.single_layers:
                if ckpt:
                    ctx = checkpoint.checkpoint(layer, ctx, g, rope, use_reentrant=False)
                else:
                    ctx = layer(ctx, g, rope, **kwargs)
            x = ctx[:, n_cond:]

        f_shift, f_scale = _seq(self.modF, g).chunk(2, dim=1)
        x = _proj(self.final_linear, modulate(x, f_shift, f_scale))
        return self.unpatchify(x, height // self.patch_size, width // self.patch_size)


class Denoiser(MMDiT):
    def __init__(self, config: DenoiserConfig) -> None:
        super().__init__(in_channels=config.in_channels, out_channels=config.out_channels, patch_size=config.patch_size,
                         num_double_layers=config.num_double_layers, num_single_layers=config.num_single_layers,
                         num_heads=config.num_attention_heads, attention_head_dim=config.attention_head_dim,
                         caption_projection_dim=config.caption_projection_dim,
                         joint_attention_dim=config.joint_attention_dim, max_pos_embed_size=config.pos_embed_max_size,
                         n_register_tokens=config.num_register_tokens, hidden_act=config.hidden_act,
                         use_flash_attn=config.use_flash_attn, use_rope=config.use_rope,
                         rope_dim_sizes=config.rope_dim_sizes, rope_theta=config.rope_theta,
                         use_shortcut=config.use_shortcut, use_guidance=config.use_guidance)
        self.config = config

    @classmethod
    def from_config(cls, config: DenoiserConfig) -> "Denoiser":
        return cls(config)
