"""Attention operator — same call surface as the reference's src/modules/attention.py:23-159, with
the MI355X kernel as the implementation.

``AttentionImplementation`` gains the value ``"hip"``; on this platform *every* accepted backend
value executes the hand-written flash-attention kernel (``libvft_hip.so: vft_attn_fwd/bwd``):
"eager"/"sdpa" name the same mathematics in the reference, and there is no second implementation
to dispatch to.  "flash_attention_2" / "xformers" raise, mirroring the reference's behaviour when
those wheels are absent (:133-134, :150).

Conventions kept from the reference: (B, H, S, D) in and out; fp32 inputs are cast to
``attention_dtype`` and the result stays in that dtype (:113-118); rank-4 assertion (:109-111);
``ValueError`` for an unknown backend (:159).  A mask must be a key-padding mask — one flag per (batch, key),
broadcast over heads and queries: the only kind the hot path produces (Lumina2's prefix mask,
lumina2/denoiser.py:150-155).  A prefix mask becomes per-batch key lengths; a key-padding mask WITH HOLES
(round 5) is served exactly by moving each batch element's valid keys to the front (softmax does not depend on
the order of the keys) and then masking by length.  Masks that differ per query or head (the reference hands
any mask to SDPA, :121-129) cannot be expressed by the kernel and raise ``ValueError``.
"""
from __future__ import annotations

import warnings
from typing import Literal

import torch

from .. import kernels

AttentionImplementation = Literal["eager", "flash_attention_2", "xformers", "sdpa", "hip"]

is_flash_attn_available = False
is_xformers_available = False


class _PermuteKeys(torch.autograd.Function):
    """x[b, h, order[b, j], :] -> position j (a per-batch permutation of the key axis); the backward applies the inverse
    permutation as another gather — no scatter-add, no atomics."""

    @staticmethod
    def forward(ctx, x, order):
        ctx.save_for_backward(order)
        idx = order[:, None, :, None].expand(x.shape[0], x.shape[1], order.shape[1], x.shape[3])
        return torch.take_along_dim(x, idx, dim=2)

    @staticmethod
    def backward(ctx, g):
        (order,) = ctx.saved_tensors
        inv = torch.empty_like(order)
        inv.scatter_(1, order, torch.arange(order.shape[1], device=order.device).expand_as(order))
        idx = inv[:, None, :, None].expand(g.shape[0], g.shape[1], inv.shape[1], g.shape[3])
        return torch.take_along_dim(g, idx, dim=2), None


def _key_mask_of(mask: torch.Tensor, batch: int, sk: int) -> torch.Tensor:
    """The (B, Sk) key flags of a (B,Sk) / (B,1,1,Sk) / (B,H,Sq,Sk)-expanded bool key-padding mask; ValueError otherwise."""
    if mask.dtype != torch.bool:
        raise ValueError("only boolean key-padding masks are supported by the HIP attention kernel")
    if mask.dim() == 2:
        if mask.shape != (batch, sk):
            raise ValueError(f"mask shape {tuple(mask.shape)} does not match (B={batch}, Sk={sk})")
        return mask
    if mask.dim() == 4:
        if mask.shape[0] != batch or mask.shape[-1] != sk:
            raise ValueError(f"mask shape {tuple(mask.shape)} does not match (B={batch}, Sk={sk})")
        # an expanded (B,1,1,Sk) mask has zero strides on the head / query axes
        if (mask.shape[1] != 1 and mask.stride(1) != 0) or (mask.shape[2] != 1 and mask.stride(2) != 0):
            raise ValueError("attention mask must be a key-padding mask broadcast over heads and queries")
        return mask[:, 0, 0, :]
    raise ValueError("attention mask must have rank 2 or 4")


def compact_keys(k: torch.Tensor, v: torch.Tensor, km: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Any (B, Sk) key mask -> (k', v', kv_len) with every batch element's valid keys moved to the front in their original
    order: attention over the first kv_len[b] keys of k' / v' equals masked attention over k / v.  Device-side only (a
    stable argsort and two gathers): usable while a HIP graph is being captured."""
    order = torch.argsort((~km).to(torch.uint8), dim=1, stable=True)
    return _PermuteKeys.apply(k, order), _PermuteKeys.apply(v, order), km.sum(dim=-1, dtype=torch.int32)


def mask_to_kv_len(mask: torch.Tensor, batch: int, sk: int) -> torch.Tensor:
    """(B,Sk) / (B,1,1,Sk) / (B,H,Sq,Sk)-expanded bool key-padding mask -> (B,) int32 lengths.

    The kernel masks by length, so the mask must be a per-batch *prefix* mask over keys, broadcast over heads and
    queries.  The reference (src/modules/attention.py:121-136) hands arbitrary masks to SDPA; anything this kernel
    cannot represent therefore raises ``ValueError`` instead of silently attending to the wrong keys.  The prefix
    test reads one flag back from the device, so this entry cannot be used while a HIP graph is being captured: hot
    paths that build their own lengths (Lumina2) pass ``kv_len=`` instead."""
    if mask.dtype != torch.bool:
        raise ValueError("only boolean key-padding masks are supported by the HIP attention kernel")
    if mask.dim() == 2:
        if mask.shape != (batch, sk):
            raise ValueError(f"mask shape {tuple(mask.shape)} does not match (B={batch}, Sk={sk})")
        km = mask
    elif mask.dim() == 4:
        if mask.shape[0] != batch or mask.shape[-1] != sk:
            raise ValueError(f"mask shape {tuple(mask.shape)} does not match (B={batch}, Sk={sk})")
        # an expanded (B,1,1,Sk) mask has zero strides on the head / query axes
        if (mask.shape[1] != 1 and mask.stride(1) != 0) or (mask.shape[2] != 1 and mask.stride(2) != 0):
            raise ValueError("attention mask must be a key-padding mask broadcast over heads and queries")
        km = mask[:, 0, 0, :]
    else:
        raise ValueError("attention mask must have rank 2 or 4")
    kv_len = km.sum(dim=-1, dtype=torch.int32)
    if mask.is_cuda and torch.cuda.is_current_stream_capturing():
        raise ValueError("a mask= cannot be validated during HIP-graph capture; pass kv_len= (per-batch key counts) instead")
    is_prefix = (km == (torch.arange(sk, device=km.device).unsqueeze(0) < kv_len.unsqueeze(1))).all()
    if not bool(is_prefix):
        raise ValueError("the HIP attention kernel supports key-padding masks whose valid keys form a prefix "
                         "(mask[b, :len_b] all True, the rest False); got a mask with holes")
    return kv_len


def scaled_dot_product_attention(
    q: torch.Tensor,  # (batch_size, num_heads, seq_len, head_dim)
    k: torch.Tensor,
    v: torch.Tensor,
    mask: torch.Tensor | None = None,
    scale: float | None = None,
    dropout: float = 0.0,
    backend: AttentionImplementation = "hip",
    attention_dtype: torch.dtype = torch.bfloat16,
    is_causal: bool = False,
    kv_len: torch.Tensor | None = None,
) -> torch.Tensor:
    assert q.dim() == k.dim() == v.dim() == 4
    if backend in ("flash_attention_2", "xformers"):
        raise AssertionError(f"{backend} is not available on this platform; use backend='hip'")
    if backend not in ("eager", "sdpa", "hip"):
        raise ValueError(f"Unknown backend: {backend}")
    if dropout != 0.0 or is_causal:
        raise ValueError("the MI355X attention kernel is non-causal with dropout 0 (all the hot path uses)")
    if attention_dtype != torch.bfloat16:
        raise ValueError("the MI355X attention kernel computes in bfloat16")
    if q.dtype != attention_dtype:
        q, k, v = q.to(attention_dtype), k.to(attention_dtype), v.to(attention_dtype)
    if mask is not None and kv_len is None:
        km = _key_mask_of(mask, q.shape[0], k.shape[2])
        capturing = mask.is_cuda and torch.cuda.is_current_stream_capturing()
        if not capturing:
            kv_len = km.sum(dim=-1, dtype=torch.int32)
            is_prefix = bool((km == (torch.arange(km.shape[1], device=km.device).unsqueeze(0) < kv_len.unsqueeze(1))).all())
        if capturing or not is_prefix:
            # holes in the key mask (or no way to look: the prefix test reads a flag back, which a capture forbids): valid
            # keys to the front, then mask by length — exact for any key-padding mask
            k, v, kv_len = compact_keys(k, v, km)
    return kernels.flash_attention(q, k, v, kv_len=kv_len, scale=scale)


def scaled_qkv_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    scale: float | None = None,
    use_flash: bool = False,
    attention_dtype: torch.dtype = torch.bfloat16,
) -> torch.Tensor:
    """Legacy AuraFlow entry (reference :31-95).  `use_flash` is accepted and ignored: both values
    run the HIP kernel.  The reference warns on every call; here once per process."""
    global _warned
    if not _warned:
        warnings.warn("scaled_qkv_attention is deprecated; use scaled_dot_product_attention")
        _warned = True
    return scaled_dot_product_attention(q, k, v, scale=scale, attention_dtype=attention_dtype)


_warned = False


def get_attn_implementation_label(use_flash_attention: bool) -> AttentionImplementation:
    return "hip"
