"""Attention oracle (CPU).

Follows src/modules/attention.py:98-159 (`scaled_dot_product_attention`): fp32 inputs are cast to
`attention_dtype` (bf16) first (:113-118), then softmax(q k^T * scale + mask) v, non-causal,
dropout 0; the result stays in the attention dtype.  `scaled_qkv_attention` (:31-95) is the same
computation without a mask.  The math below is done in fp32 on the bf16-rounded operands, which is
what the reference's CPU SDPA kernel accumulates in.
"""
from __future__ import annotations

import math

import torch


def sdpa(
    q: torch.Tensor,  # (B, Hq, Sq, D)
    k: torch.Tensor,  # (B, Hkv, Sk, D)
    v: torch.Tensor,
    mask: torch.Tensor | None = None,  # bool, broadcastable to (B, Hq, Sq, Sk); True = attend
    scale: float | None = None,
    attention_dtype: torch.dtype = torch.bfloat16,
    kv_len: torch.Tensor | None = None,  # (B,) alternative spelling of a key-padding mask
) -> torch.Tensor:
    assert q.dim() == k.dim() == v.dim() == 4
    out_dtype = attention_dtype if q.dtype == torch.float32 else q.dtype
    q32 = q.to(out_dtype).float()
    k32 = k.to(out_dtype).float()
    v32 = v.to(out_dtype).float()
    hq, hkv = q32.shape[1], k32.shape[1]
    if hq != hkv:  # GQA: the reference materialises the repeat (lumina2/denoiser.py:146-147)
        rep = hq // hkv
        k32 = k32.repeat_interleave(rep, dim=1)
        v32 = v32.repeat_interleave(rep, dim=1)
    if scale is None:
        scale = 1.0 / math.sqrt(q32.shape[-1])
    s = torch.matmul(q32, k32.transpose(-1, -2)) * scale
    if kv_len is not None:
        sk = k32.shape[2]
        km = torch.arange(sk)[None, :] < kv_len.to(torch.int64).cpu()[:, None]  # (B, Sk)
        mask = km[:, None, None, :] if mask is None else (mask & km[:, None, None, :])
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v32).to(out_dtype)


_TRUTH = False


class fp32_truth:
    """Context manager: make `sdpa_ref` keep everything in fp32 (no bf16 cast).  Used by tests to
    obtain a high-precision yardstick from the same restatement (run WITHOUT autocast)."""

    def __enter__(self):
        global _TRUTH
        self._old, _TRUTH = _TRUTH, True

    def __exit__(self, *exc):
        global _TRUTH
        _TRUTH = self._old


def sdpa_ref(q, k, v, mask=None, scale=None, attention_dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """The reference's own CPU execution of the op: cast fp32 -> attention dtype (:113-118) and call
    torch's SDPA in that dtype (:121-129).  Used inside the model oracles so that they track the
    reference bit-for-bit on CPU; `sdpa` above (fp32 math) is the higher-precision yardstick the GPU
    kernel is compared against."""
    import torch.nn.functional as F

    if _TRUTH:
        return F.scaled_dot_product_attention(q.float(), k.float(), v.float(), attn_mask=mask, scale=scale)
    if q.dtype == torch.float32:
        q, k, v = q.to(attention_dtype), k.to(attention_dtype), v.to(attention_dtype)
    return F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale)
