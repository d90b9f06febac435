"""Lumina-Image-2.0 NextDiT denoiser for MI355X.

Operator surface of the reference's src/models/lumina2/denoiser.py: class names, constructor
arguments, parameter names/shapes and the ragged call convention
``forward(latents, caption_features, timestep, caption_mask, cached_caption_features=None)
-> (nested latents, caption_mask, caption_features)`` (:949-957, :1082).

Execution differences (same results):
* caption lengths are read from the mask with ONE host transfer per step (the reference does a
  ``.item()`` per sample in four places: :735, :800, :924-925, :1039); all ragged bookkeeping after
  that is host integer work, and the prefix masks travel to the attention kernel as per-batch key
  lengths (``kv_len``) instead of an expanded (B,24,S,S) bool tensor (:150-155);
* GQA is native in the attention kernel (no k/v ``repeat``, :146-147);
* per-head RMSNorm on q/k plus the complex RoPE rotation (:96-112, :137-143) are one fused kernel
  (a complex multiply by e^{iθ} is the interleaved-pair rotation);
* sandwich RMSNorm + (1+scale) modulation and tanh-gated residuals are fused kernels.
"""
from __future__ import annotations

import math
import os
from typing import NamedTuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ... import kernels
from ...modules.attention import scaled_dot_product_attention
from ...modules.norm import FP32LayerNorm
from ...modules.timestep.embedding import get_timestep_embedding
from ...utils.tensor import module_dtype
from .config import DenoiserConfig

# A/B switch.  1 (default): the q | k | v views of the fused projection are one autograd node (their gradients return as one
# concatenated tensor) and a block's input reaches its gated add through the pre-norm's second output (the two gradients of the
# input meet inside the norm-backward kernel).  0: plain slices and plain fan-out, summed by autograd's elementwise passes.
GRAD_FUSION = os.environ.get("VFT_LUMINA_GRAD_FUSION", "1") != "0"

DENOISER_TENSOR_PREFIX = "model.diffusion_model."


class TimestepEmbedder(nn.Module):
    def __init__(self, hidden_dim: int, time_embed_dim: int, bias: bool = True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.time_embed_dim = time_embed_dim
        self.mlp = nn.Sequential(nn.Linear(time_embed_dim, hidden_dim, bias=bias), nn.SiLU(),
                                 nn.Linear(hidden_dim, hidden_dim, bias=bias))

    def init_weights(self):
        for m in self.mlp.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, timesteps: torch.Tensor) -> torch.Tensor:
        emb = get_timestep_embedding(timesteps, self.time_embed_dim, flip_sin_to_cos=True, downscale_freq_shift=0)
        return _seq(self.mlp, emb.to(module_dtype(self.mlp[0])))


class Rope(NamedTuple):
    """cos/sin tables (B, S, D/2) fp32 — the real and imaginary parts of the reference's freqs_cis."""

    cos: torch.Tensor
    sin: torch.Tensor

    def rows(self, sel) -> "Rope":
        return Rope(self.cos[sel], self.sin[sel])


def _proj(layer: nn.Module, h: torch.Tensor) -> torch.Tensor:
    """layer(h).  A plain frozen bf16 nn.Linear (the refiner blocks of a LoRA run: not adapted, not quantised) runs on the
    projection GEMM of the adapted layers (kernels.linear) instead of the library's: at these sizes hipBLASLt picks stream-K
    solutions whose partial sums meet through atomics — the only launches of a Lumina2 step that were not run-to-run
    reproducible.  LoRA / NF4 wrappers and trainable layers keep their own forward."""
    if type(layer) is nn.Linear and h.is_cuda and kernels.linear_eligible(h, layer.weight):
        return kernels.linear(h, layer.weight, layer.bias)
    return layer(h)


def _seq(seq: nn.Sequential, h: torch.Tensor) -> torch.Tensor:
    """nn.Sequential.forward with its Linear members through `_proj` (the conditioning MLPs: SiLU -> Linear, Linear -> SiLU -> Linear)."""
    for m in seq:
        h = _proj(m, h)
    return h


class SelfAttention(nn.Module):
    def __init__(self, hidden_dim: int, num_heads: int, num_kv_heads: int):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = hidden_dim // num_heads
        self.num_repeats = num_heads // num_kv_heads
        self.q_dim = num_heads * self.head_dim
        self.k_dim = num_kv_heads * self.head_dim
        self.v_dim = num_kv_heads * self.head_dim
        self.qkv = nn.Linear(hidden_dim, self.q_dim + self.k_dim + self.v_dim, bias=False)
        self.out = nn.Linear(self.q_dim, hidden_dim, bias=False)
        self.q_norm = nn.RMSNorm(self.head_dim, eps=1e-6)
        self.k_norm = nn.RMSNorm(self.head_dim, eps=1e-6)

    def init_weights(self):
        nn.init.xavier_uniform_(self.qkv.weight)
        nn.init.xavier_uniform_(self.out.weight)

    def forward(self, hidden_states: torch.Tensor, freqs_cis: Rope, mask: torch.Tensor | None = None,
                kv_len: torch.Tensor | None = None) -> torch.Tensor:
        b, s, _ = hidden_states.shape
        qkv = _proj(self.qkv, hidden_states)
        D = self.head_dim
        # strided views into the fused projection output: no split copies
        # (one autograd node for the three views: their gradients come back side by side in one pass)
        if GRAD_FUSION:
            q, k, v = kernels.split_last(qkv, (self.q_dim, self.k_dim, self.v_dim))
        else:
            q, k, v = qkv.split([self.q_dim, self.k_dim, self.v_dim], dim=-1)
        q = q.unflatten(-1, (self.num_heads, D))
        k = k.unflatten(-1, (self.num_kv_heads, D))
        v = v.unflatten(-1, (self.num_kv_heads, D))
        q = kernels.qk_norm_rope(q, "rms", self.q_norm.weight, 1e-6, freqs_cis.cos, freqs_cis.sin)
        k = kernels.qk_norm_rope(k, "rms", self.k_norm.weight, 1e-6, freqs_cis.cos, freqs_cis.sin)
        attn = scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask=mask,
                                            kv_len=kv_len, scale=math.sqrt(1 / D))
        return _proj(self.out, attn.transpose(1, 2).flatten(-2))


class FeedForward(nn.Module):
    def __init__(self, hidden_dim: int, intermediate_dim: int, multiple_of: int = 256):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.intermediate_dim = multiple_of * ((intermediate_dim + multiple_of - 1) // multiple_of)
        self.w1 = nn.Linear(hidden_dim, self.intermediate_dim, bias=False)
        self.w2 = nn.Linear(self.intermediate_dim, hidden_dim, bias=False)
        self.w3 = nn.Linear(hidden_dim, self.intermediate_dim, bias=False)

    def init_weights(self):
        for lin in (self.w1, self.w2, self.w3):
            nn.init.xavier_uniform_(lin.weight)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        pair = kernels.lora_siblings(hidden_states, (self.w1, self.w3)) if hidden_states.is_cuda else None  # chained data gradients
        a, b = pair if pair is not None else (_proj(self.w1, hidden_states), _proj(self.w3, hidden_states))
        return _proj(self.w2, kernels.swiglu(a, b))


class TransformerBlock(nn.Module):
    def __init__(self, hidden_dim: int, num_heads: int, num_kv_heads: int, multiple_of: int = 256,
                 norm_eps: float = 1e-5, use_adaln: bool = True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.use_adaln = use_adaln
        self.attention = SelfAttention(hidden_dim=hidden_dim, num_heads=num_heads, num_kv_heads=num_kv_heads)
        self.feed_forward = FeedForward(hidden_dim=hidden_dim, intermediate_dim=hidden_dim * 4, multiple_of=multiple_of)
        self.attention_norm1 = nn.RMSNorm(hidden_dim, eps=norm_eps)
        self.ffn_norm1 = nn.RMSNorm(hidden_dim, eps=norm_eps)
        self.attention_norm2 = nn.RMSNorm(hidden_dim, eps=norm_eps)
        self.ffn_norm2 = nn.RMSNorm(hidden_dim, eps=norm_eps)
        if use_adaln:
            self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(1024, 4 * hidden_dim, bias=True))

    def init_weights(self):
        self.attention.init_weights()
        self.feed_forward.init_weights()
        if self.use_adaln:
            nn.init.zeros_(self.adaLN_modulation[1].weight)
            nn.init.zeros_(self.adaLN_modulation[1].bias)

    def modulate(self, tensor: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        return tensor * (1 + scale.unsqueeze(1))

    def _norm(self, norm: nn.RMSNorm, x: torch.Tensor, scale: torch.Tensor | None = None) -> torch.Tensor:
        return kernels.rms_norm_scale(x, norm.weight, scale, norm.eps)

    def _norm_skip(self, norm: nn.RMSNorm, x: torch.Tensor, scale: torch.Tensor | None = None):
        """(norm(x), x) with the second output carrying the residual path (kernels.rms_norm_scale_skip)."""
        if GRAD_FUSION and x.is_cuda:
            return kernels.rms_norm_scale_skip(x, norm.weight, scale, norm.eps)
        return self._norm(norm, x, scale), x

    def forward(self, hidden_states: torch.Tensor, freqs_cis: Rope, adaln_input: torch.Tensor | None = None,
                mask: torch.Tensor | None = None, kv_len: torch.Tensor | None = None) -> torch.Tensor:
        x = hidden_states
        if self.use_adaln:
            assert adaln_input is not None, "adaln_input must be provided for AdaLN"
            sc_a, g_a, sc_m, g_m = _seq(self.adaLN_modulation, adaln_input).chunk(4, dim=1)
            # the block input feeds norm1 and the gated add: the two gradients meet inside norm1's backward kernel
            n, x = self._norm_skip(self.attention_norm1, x, sc_a)
            a = self.attention(n, freqs_cis, mask=mask, kv_len=kv_len)
            x = kernels.gate_residual(x, g_a, self._norm(self.attention_norm2, a), tanh=True)
            n, x = self._norm_skip(self.ffn_norm1, x, sc_m)
            m = self.feed_forward(n)
            return kernels.gate_residual(x, g_m, self._norm(self.ffn_norm2, m), tanh=True)
        a = self.attention(self._norm(self.attention_norm1, x), freqs_cis, mask=mask, kv_len=kv_len)
        x = x + self._norm(self.attention_norm2, a)
        m = self.feed_forward(self._norm(self.ffn_norm1, x))
        return x + self._norm(self.ffn_norm2, m)


class FinalLayer(nn.Module):
    def __init__(self, hidden_dim: int, patch_size: int = 2, out_channels: int = 16):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.patch_size = patch_size
        self.out_channels = out_channels
        self.norm_final = FP32LayerNorm(hidden_dim, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_dim, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(1024, hidden_dim, bias=True))

    def modulate(self, tensor: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
        return tensor * (1 + scale.unsqueeze(1))

    def forward(self, hidden_states: torch.Tensor, adaln_input: torch.Tensor) -> torch.Tensor:
        scale = _seq(self.adaLN_modulation, adaln_input)
        zero = torch.zeros_like(scale)
        return _proj(self.linear, kernels.layer_norm_modulate(hidden_states, zero, scale, self.norm_final.eps))


class RoPEEmbedder:
    """Per-axis rotation tables, angles in fp64 then fp32 cos/sin (the reference builds complex64
    e^{iθ} from the same fp32 angles, :443-463) and a gather by integer position ids (:465-482)."""

    def __init__(self, theta: float = 10000.0, axes_dims: list[int] = [32, 32, 32], axes_lens: list[int] = [300, 512, 512]):
        self.theta = theta
        self.axes_dims = axes_dims
        self.axes_lens = axes_lens
        self.tables = self.precompute(axes_dims, axes_lens, theta)

    @staticmethod
    def precompute(dims: list[int], ends: list[int], theta: float) -> list[tuple[torch.Tensor, torch.Tensor]]:
        out = []
        for d, e in zip(dims, ends):
            # (explicit device: the tables are plain tensors, not parameters — a `torch.device("meta")` construction
            # context, as used when a checkpoint is about to be assigned, must not turn them into meta tensors)
            inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device="cpu") / d))
            ang = torch.outer(torch.arange(e, dtype=torch.float64, device="cpu"), inv).float()
            out.append((torch.cos(ang), torch.sin(ang)))
        return out

    def __call__(self, ids: torch.Tensor) -> Rope:
        dev = ids.device
        self.tables = [(c.to(dev), s.to(dev)) for c, s in self.tables]
        idx = ids.long()
        cos = torch.cat([c[idx[..., i]] for i, (c, _) in enumerate(self.tables)], dim=-1)
        sin = torch.cat([s[idx[..., i]] for i, (_, s) in enumerate(self.tables)], dim=-1)
        return Rope(cos, sin)


class ImageSize(NamedTuple):
    height: int
    width: int

    @classmethod
    def from_tensor(cls, sizes: torch.Tensor) -> list["ImageSize"]:
        return [cls(height=int(h), width=int(w)) for h, w in sizes.tolist()]


def _pad_rows(items: list[torch.Tensor], device, want_mask: bool = True) -> tuple[torch.Tensor, torch.Tensor | None, list[int]]:
    """zero-pad along dim 0 and stack; returns (padded, prefix bool mask, lengths).  want_mask=False skips the mask (its
    construction copies the lengths to the device — not allowed while a HIP graph is being captured)."""
    lens = [int(t.size(0)) for t in items]
    mx = max(lens)
    if all(n == mx for n in lens):
        padded = torch.stack(items, dim=0)
    else:
        padded = torch.stack([F.pad(t, [0, 0] * (t.dim() - 1) + [0, mx - t.size(0)]) for t in items], dim=0)
    mask = (torch.arange(mx, device=device).unsqueeze(0) < torch.tensor(lens, device=device).unsqueeze(1)) if want_mask else None
    return padded, mask, lens


class _TakeRows(torch.autograd.Function):
    """out[j] = src[idx[j]] (idx[j] < 0: a zero row) for an index that hits every source row AT MOST ONCE — the row moves of
    the static-shape batch assembly below.  Backward is the inverse gather (no scatter-add, no atomics: deterministic), built
    from the index on the device with static shapes, so both directions replay from a HIP graph."""

    @staticmethod
    def forward(ctx, src, idx):
        valid = idx >= 0
        out = torch.where(valid.unsqueeze(-1), src.index_select(0, idx.clamp_min(0)), src.new_zeros(()))
        ctx.save_for_backward(idx, valid)
        ctx.rows = src.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        idx, valid = ctx.saved_tensors
        n = ctx.rows
        inv = torch.full((n + 1,), -1, dtype=torch.long, device=idx.device)
        # invalid entries all land in the spare slot n
        inv.scatter_(0, torch.where(valid, idx, torch.full_like(idx, n)), torch.arange(idx.numel(), device=idx.device))
        inv = inv[:n]
        gs = torch.where((inv >= 0).unsqueeze(-1), g.index_select(0, inv.clamp_min(0)), g.new_zeros(()))
        return gs, None


class NextDiT(nn.Module):
    def __init__(self, patch_size: int = 2, in_channels: int = 16, hidden_dim: int = 2304, timestep_embed_dim: int = 256,
                 num_layers: int = 24, num_refiner_layers: int = 2, num_heads: int = 24, num_kv_heads: int = 8,
                 multiple_of: int = 256, norm_eps: float = 1e-5, caption_dim: int = 5120,
                 axes_dims: list[int] = [16, 56, 56], axes_lens: list[int] = [1, 512, 512]):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.hidden_dim = hidden_dim
        self.axes_dims = axes_dims
        self.axes_lens = axes_lens
        blk = dict(hidden_dim=hidden_dim, num_heads=num_heads, num_kv_heads=num_kv_heads, multiple_of=multiple_of,
                   norm_eps=norm_eps)
        self.x_embedder = nn.Linear(patch_size * patch_size * in_channels, hidden_dim, bias=True)
        self.noise_refiner = nn.ModuleList([TransformerBlock(**blk) for _ in range(num_refiner_layers)])
        self.context_refiner = nn.ModuleList([TransformerBlock(**blk, use_adaln=False) for _ in range(num_refiner_layers)])
        self.t_embedder = TimestepEmbedder(hidden_dim=1024, time_embed_dim=timestep_embed_dim)
        self.cap_embedder = nn.Sequential(nn.RMSNorm(caption_dim, eps=norm_eps), nn.Linear(caption_dim, hidden_dim, bias=True))
        self.layers = nn.ModuleList([TransformerBlock(**blk) for _ in range(num_layers)])
        self.norm_final = nn.RMSNorm(hidden_dim, eps=norm_eps)  # present in checkpoints, unused in forward (as in the reference)
        self.final_layer = FinalLayer(hidden_dim=hidden_dim, patch_size=patch_size, out_channels=in_channels)
        self.rope_embedder = RoPEEmbedder(axes_dims=axes_dims, axes_lens=axes_lens)
        self.gradient_checkpointing = False

    # ---- layout helpers (integer/index work, bit-exact vs the reference) -------------------
    def patchify(self, image: torch.Tensor) -> torch.Tensor:
        """(C,H,W) -> (H/p * W/p, p*p*C), features ordered (ph, pw, c)."""
        c, h, w = image.shape
        p = self.patch_size
        return image.reshape(c, h // p, p, w // p, p).permute(1, 3, 2, 4, 0).reshape((h // p) * (w // p), p * p * c)

    def unpatchify(self, patches: torch.Tensor, latent_height: int, latent_width: int, patch_size: int = 2,
                   out_channels: int = 16) -> torch.Tensor:
        gh, gw = latent_height // patch_size, latent_width // patch_size
        grid = patches.reshape(gh, gw, patch_size, patch_size, out_channels).permute(4, 0, 2, 1, 3)
        return grid.reshape(out_channels, gh * patch_size, gw * patch_size)

    def nested_unpatchify(self, patches, image_sizes: list[ImageSize], padded: bool = False) -> torch.Tensor:
        images = [self.unpatchify(p, latent_height=s.height, latent_width=s.width, patch_size=self.patch_size,
                                  out_channels=self.out_channels) for p, s in zip(patches, image_sizes)]
        if padded:
            # what torch.nested.to_padded_tensor(nested, 0.0) returns, built directly: the nested tensor's backward copies its
            # size table from the host, which a HIP-graph capture does not allow
            mh, mw = max(s.height for s in image_sizes), max(s.width for s in image_sizes)
            if all(s.height == mh and s.width == mw for s in image_sizes):
                return torch.stack(images, dim=0)
            return torch.stack([F.pad(im, [0, mw - im.shape[-1], 0, mh - im.shape[-2]]) for im in images], dim=0)
        return torch.nested.as_nested_tensor(images)

    @torch.no_grad()
    def get_position_ids(self, caption_length: int, patches_height: int, patches_width: int,
                         device: torch.device = torch.device("cpu")) -> torch.Tensor:
        n = patches_height * patches_width
        ids = torch.zeros(caption_length + n, 3, dtype=torch.int32, device=device)
        ids[:caption_length, 0] = torch.arange(caption_length, dtype=torch.int32, device=device)
        ids[caption_length:, 0] = caption_length
        ys = torch.arange(patches_height, dtype=torch.int32, device=device)
        xs = torch.arange(patches_width, dtype=torch.int32, device=device)
        ids[caption_length:, 1] = ys.repeat_interleave(patches_width)
        ids[caption_length:, 2] = xs.repeat(patches_height)
        return ids

    @torch.no_grad()
    def get_caption_lens(self, captions, caption_masks=None) -> torch.Tensor:
        if caption_masks is None:
            lens = [int(c.size(0)) for c in captions]
        else:
            m = caption_masks if isinstance(caption_masks, torch.Tensor) else torch.stack(list(caption_masks))
            lens = m.sum(dim=-1).tolist()  # one host transfer
        return torch.tensor(lens, dtype=torch.int32, device=captions[0].device)

    @torch.no_grad()
    def get_image_lens(self, images, patch_size: int = 1) -> torch.Tensor:
        lens = [(im.size(-2) // patch_size) * (im.size(-1) // patch_size) for im in images]
        return torch.tensor(lens, dtype=torch.int32, device=images[0].device)

    # ---- forward ---------------------------------------------------------------------------
    def _run(self, layer: TransformerBlock, x, rope: Rope, t, mask, kv_len):
        if self.training and self.gradient_checkpointing:
            return checkpoint(layer, x, rope, t, mask, kv_len, use_reentrant=False)
        return layer(x, rope, adaln_input=t, mask=mask, kv_len=kv_len)

    def _layout(self, cap_lens: tuple[int, ...], sizes: tuple[ImageSize, ...], cap_rows: int, dev) -> dict:
        """Device-resident constants of one (caption lengths, image sizes) combination — position ids -> rotary tables of
        the [caption | image] sequences, their caption / image row subsets, the three length vectors and the caption prefix
        mask — built once and kept (16 most recent combinations): nothing of this depends on the step's data, and with it
        cached a forward pass copies nothing from the host, so it can be captured into a HIP graph."""
        key = (cap_lens, sizes, cap_rows, str(dev))
        cache = self.__dict__.setdefault("_layouts", {})
        hit = cache.get(key)
        if hit is not None:
            return hit
        p = self.patch_size
        b = len(cap_lens)
        img_lens = [(s.height // p) * (s.width // p) for s in sizes]
        tot_lens = [c + n for c, n in zip(cap_lens, img_lens)]
        max_len = max(tot_lens)
        with torch.no_grad():
            # position ids for [caption | image] per sample, padded to the longest sample
            ids = torch.zeros(b, max_len, 3, dtype=torch.int32, device=dev)
            for i, (cl, s) in enumerate(zip(cap_lens, sizes)):
                ids[i, : tot_lens[i]] = self.get_position_ids(cl, s.height // p, s.width // p, device=dev)
            rope = self.rope_embedder(ids)
            len_t = lambda L: torch.tensor(L, dtype=torch.int32, device=dev)  # noqa: E731
            mc, mi = max(cap_lens), max(img_lens)
            gather = torch.zeros(b, mi, dtype=torch.long, device=dev)
            for i in range(b):
                gather[i, : img_lens[i]] = torch.arange(cap_lens[i], cap_lens[i] + img_lens[i], device=dev)
            hit = {
                "rope": rope,
                "cap_rope": Rope(rope.cos[:, :mc].contiguous(), rope.sin[:, :mc].contiguous()),  # caption rows come first
                "img_rope": Rope(torch.gather(rope.cos, 1, gather.unsqueeze(-1).expand(-1, -1, rope.cos.shape[-1])),
                                 torch.gather(rope.sin, 1, gather.unsqueeze(-1).expand(-1, -1, rope.sin.shape[-1]))),
                "ck": len_t(list(cap_lens)), "ik": len_t(img_lens), "tk": len_t(tot_lens),
                "cap_mask": torch.arange(cap_rows, device=dev).unsqueeze(0) < len_t(list(cap_lens)).unsqueeze(1),
                "img_lens": img_lens, "tot_lens": tot_lens, "max_len": max_len,
            }
        if len(cache) >= 16:
            cache.pop(next(iter(cache)))
        cache[key] = hit
        return hit

    def _forward_static(self, latents: torch.Tensor, caption_features: torch.Tensor, timestep: torch.Tensor,
                        caption_mask: torch.Tensor):
        """The same forward pass with EVERY shape independent of the caption lengths (round 4): one same-size image batch
        (B,C,H,W), captions padded to L rows with a prefix mask.  The lengths stay on the device (`caption_mask.sum`): position
        ids -> rotary tables, the [caption | image] assembly and the final split are gathers by indices computed from them,
        the sequences are L + N rows long for every sample (rows behind a sample's end are dead: masked as keys through
        `kv_len`, ignored as queries), so ONE captured HIP graph serves every batch of a bucket — the reference's per-sample
        Python assembly (src/models/lumina2/denoiser.py:809-932) reads the lengths on the host four times per sample.
        Valid rows see exactly the operations of `forward` (same kernels, same key ranges)."""
        dev = timestep.device
        dtype = module_dtype(self.x_embedder)
        b, c, H, W = latents.shape
        p = self.patch_size
        gh, gw = H // p, W // p
        N, L = gh * gw, caption_features.shape[1]
        T = L + N
        with torch.no_grad():
            ck = caption_mask.bool().sum(dim=-1, dtype=torch.int32)  # (B,) valid caption rows — never leaves the device
            tk = ck + N
            ik = torch.full((b,), N, dtype=torch.int32, device=dev)
            lenb = ck.long().unsqueeze(1)
            s_ = torch.arange(T, device=dev).unsqueeze(0)
            is_cap = s_ < lenb
            j = (s_ - lenb).clamp(0, N - 1)
            in_img = (~is_cap) & (s_ < lenb + N)
            zero = torch.zeros((), dtype=torch.long, device=dev)
            ids = torch.stack([torch.where(is_cap, s_, torch.where(in_img, lenb, zero)),
                               torch.where(in_img, j // gw, zero), torch.where(in_img, j % gw, zero)], dim=-1)
            rope = self.rope_embedder(ids)
            cap_ids = torch.stack([torch.arange(L, device=dev), torch.zeros(L, dtype=torch.long, device=dev),
                                   torch.zeros(L, dtype=torch.long, device=dev)], dim=-1).unsqueeze(0).expand(b, -1, -1)
            cap_rope = self.rope_embedder(cap_ids)
            jn = torch.arange(N, device=dev)
            img_ids = torch.stack([lenb.expand(b, N), (jn // gw).unsqueeze(0).expand(b, N), (jn % gw).unsqueeze(0).expand(b, N)], dim=-1)
            img_rope = self.rope_embedder(img_ids)
            boff = (torch.arange(b, device=dev) * T).unsqueeze(1)  # source rows per sample: L caption rows, then N image rows
            gidx = (torch.where(is_cap, s_, torch.where(in_img, L + j, torch.full_like(s_, -1))) + torch.where(is_cap | in_img, boff, zero)).reshape(-1)
            oidx = (lenb + jn.unsqueeze(0) + boff).reshape(-1)

        t = self.t_embedder(timestep)
        cap = _proj(self.cap_embedder[1], kernels.rms_norm_scale(caption_features.to(dtype), self.cap_embedder[0].weight, None,
                                                          self.cap_embedder[0].eps))
        for layer in self.context_refiner:
            cap = self._run(layer, cap, cap_rope, None, None, ck)
        patches = latents.to(dtype).reshape(b, c, gh, p, gw, p).permute(0, 2, 4, 3, 5, 1).reshape(b, N, p * p * c)
        x = _proj(self.x_embedder, patches)
        for layer in self.noise_refiner:
            x = self._run(layer, x, img_rope, t, None, ik)
        ctx = _TakeRows.apply(torch.cat([cap, x], dim=1).reshape(b * T, -1), gidx).view(b, T, -1)
        for layer in self.layers:
            ctx = self._run(layer, ctx, rope, t, None, tk)
        out = self.final_layer(ctx, t)
        pieces = _TakeRows.apply(out.reshape(b * T, -1), oidx).view(b, gh, gw, p, p, self.out_channels)
        images = pieces.permute(0, 5, 1, 3, 2, 4).reshape(b, self.out_channels, gh * p, gw * p)
        return images, caption_mask.bool(), cap

    def forward(self, latents, caption_features, timestep: torch.Tensor, caption_mask, cached_caption_features=None,
                caption_lengths: list[int] | None = None, padded_output: bool = False, static_layout: bool = False):
        """`caption_lengths`: the number of valid caption tokens per sample as host integers, when the caller has them
        (the data loader knows them before the batch goes to the device); without them they are read back from
        `caption_mask` — the forward pass's only host synchronisation.  `padded_output`: return the images zero-padded to
        the largest one and stacked (= `torch.nested.to_padded_tensor(out, 0.0)` of the default nested output).
        `static_layout`: the shape-static form for one same-size batch (`_forward_static`): no host integers at all, one HIP
        graph for every caption-length combination; returns the stacked images."""
        if static_layout:
            if not (isinstance(latents, torch.Tensor) and not latents.is_nested and latents.dim() == 4 and isinstance(caption_features, torch.Tensor)
                    and caption_features.dim() == 3 and isinstance(caption_mask, torch.Tensor) and cached_caption_features is None):
                raise ValueError("static_layout needs one same-size image batch (B,C,H,W), dense caption features (B,L,C) and a (B,L) prefix mask")
            return self._forward_static(latents, caption_features, timestep, caption_mask)
        dev = timestep.device
        dtype = module_dtype(self.x_embedder)
        images = list(latents.unbind()) if isinstance(latents, torch.Tensor) else list(latents)
        b = len(images)
        p = self.patch_size

        if caption_lengths is not None:
            cap_lens = [int(n) for n in caption_lengths]
        else:
            cmask = caption_mask if isinstance(caption_mask, torch.Tensor) else _pad_rows([m.bool() for m in caption_mask], dev)[0]
            cap_lens = cmask.bool().sum(dim=-1).tolist()  # the step's single host sync
        sizes = [ImageSize(int(im.shape[-2]), int(im.shape[-1])) for im in images]
        cap_rows = cached_caption_features.shape[1] if cached_caption_features is not None else max(cap_lens)
        lay = self._layout(tuple(cap_lens), tuple(sizes), int(cap_rows), dev)
        img_lens, tot_lens, max_len, rope = lay["img_lens"], lay["tot_lens"], lay["max_len"], lay["rope"]

        t = self.t_embedder(timestep)

        # caption refiner (no AdaLN)
        cap_mask = lay["cap_mask"]
        if cached_caption_features is not None:
            cap = cached_caption_features
        else:
            # prefix masks: the valid caption rows of sample i are rows [0, cap_lens[i])
            cap_in, _, _ = _pad_rows([caption_features[i][: cap_lens[i]] for i in range(b)], dev, want_mask=False)
            cap = _proj(self.cap_embedder[1], kernels.rms_norm_scale(cap_in.to(dtype), self.cap_embedder[0].weight, None,
                                                              self.cap_embedder[0].eps))
            for layer in self.context_refiner:
                cap = self._run(layer, cap, lay["cap_rope"], None, None, lay["ck"])

        # noise refiner
        patches, _, _ = _pad_rows([self.patchify(im.to(dtype)) for im in images], dev, want_mask=False)
        x = _proj(self.x_embedder, patches)
        for layer in self.noise_refiner:
            x = self._run(layer, x, lay["img_rope"], t, None, lay["ik"])

        # [caption | image] per sample
        ctx = torch.zeros(b, max_len, cap.shape[-1], dtype=cap.dtype, device=dev)
        for i in range(b):
            ctx[i, : cap_lens[i]] = cap[i, : cap_lens[i]]
            ctx[i, cap_lens[i] : tot_lens[i]] = x[i, : img_lens[i]]

        for layer in self.layers:
            ctx = self._run(layer, ctx, rope, t, None, lay["tk"])

        out = self.final_layer(ctx, t)
        pieces = [out[i, cap_lens[i] : tot_lens[i]] for i in range(b)]
        return self.nested_unpatchify(pieces, sizes, padded=padded_output), cap_mask, cap


class Denoiser(NextDiT):
    def __init__(self, config: DenoiserConfig):
        super().__init__(patch_size=config.patch_size, in_channels=config.in_channels, hidden_dim=config.hidden_dim,
                         timestep_embed_dim=config.timestep_embed_dim, num_layers=config.depth,
                         num_refiner_layers=config.refiner_depth, num_heads=config.num_heads,
                         num_kv_heads=config.num_kv_heads, multiple_of=config.multiple_of, norm_eps=config.norm_eps,
                         caption_dim=config.caption_dim, axes_dims=config.axes_dims, axes_lens=config.axes_lens)
        self.config = config

    def set_gradient_checkpointing(self, value: bool):
        self.gradient_checkpointing = value
