"""torch.autograd.Function wrappers over the C ABI (include/vft_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only; every op below
runs a hand-written HIP kernel from libvft_hip.so on ``torch.cuda.current_stream()``.
There is no CPU path: tensors that are not on a CUDA(HIP) device raise.
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _C


def _ptr(t: torch.Tensor | None) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu(*tensors: torch.Tensor, dtype: torch.dtype | None = torch.bfloat16) -> None:
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "vision_ft_amd kernels run on MI355X only: got a tensor on "
                f"{t.device}. There is no CPU fallback (use oracle/ in tests for CPU checks)."
            )
        if dtype is not None and t.dtype != dtype:
            raise TypeError(f"expected {dtype}, got {t.dtype}")


def _bhsd_ok(t: torch.Tensor) -> bool:
    return t.stride(-1) == 1 and all(s % 8 == 0 for s in t.stride()[:3]) and t.data_ptr() % 16 == 0


def _as_bhsd(t: torch.Tensor) -> torch.Tensor:
    """Return `t` (B,H,S,D) with head_dim contiguous and 8-element-aligned strides (copy if needed)."""
    if _bhsd_ok(t):
        return t
    b, h, s, d = t.shape
    return t.permute(0, 2, 1, 3).contiguous().permute(0, 2, 1, 3)


def _empty_bshd_as_bhsd(b: int, h: int, s: int, d: int, like: torch.Tensor) -> torch.Tensor:
    # memory (B,S,H,D), logical (B,H,S,D): the transformer blocks re-flatten heads right after
    # attention, which is then a free view
    return torch.empty((b, s, h, d), dtype=like.dtype, device=like.device).permute(0, 2, 1, 3)


# ---------------------------------------------------------------------------------------------
# optional per-launch timing with HIP events on the launch stream (used by bench.py's roofline leg;
# events are asynchronous — nothing here synchronises)
# ---------------------------------------------------------------------------------------------
_PROFILE: dict[str, list] | None = None


def set_profiling(enabled: bool) -> None:
    global _PROFILE
    _PROFILE = {} if enabled else None


class _Timed:
    def __init__(self, name: str, work: float):
        self.name, self.work = name, work

    def __enter__(self):
        if _PROFILE is not None:
            self.t0 = torch.cuda.Event(enable_timing=True)
            self.t1 = torch.cuda.Event(enable_timing=True)
            self.t0.record()
        return self

    def __exit__(self, *exc):
        if _PROFILE is not None:
            self.t1.record()
            _PROFILE.setdefault(self.name, []).append((self.work, self.t0, self.t1))


def profile_report() -> dict[str, dict]:
    """{kernel: {"launches", "work" (flops or bytes, summed), "ms" (summed), "avg_ms"}}; call after
    torch.cuda.synchronize()."""
    out = {}
    for name, recs in (_PROFILE or {}).items():
        ms = sum(a.elapsed_time(b) for _, a, b in recs)
        out[name] = {"launches": len(recs), "work": sum(w for w, _, _ in recs), "ms": ms, "avg_ms": ms / max(len(recs), 1)}
    return out


class _FlashAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, kv_len, scale):
        _require_gpu(q, k, v)
        q, k, v = _as_bhsd(q), _as_bhsd(k), _as_bhsd(v)
        B, Hq, Sq, D = q.shape
        _, Hkv, Sk, _ = k.shape
        if v.shape != k.shape or k.shape[0] != B or k.shape[3] != D:
            raise ValueError(f"attention shape mismatch: q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
        if kv_len is not None:
            _require_gpu(kv_len, dtype=torch.int32)
            if kv_len.shape != (B,):
                raise ValueError("kv_len must be int32 of shape (B,)")
            kv_len = kv_len.contiguous()
        out = _empty_bshd_as_bhsd(B, Hq, Sq, D, q)
        lse = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device), _Timed(f"attn_fwd_d{D}", 4.0 * B * Hq * Sq * Sk * D):
            st = _C.lib().vft_attn_fwd(
                _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), _ptr(kv_len),
                B, Hq, Hkv, Sq, Sk, D,
                _C.i64x3(q.stride()[:3]), _C.i64x3(k.stride()[:3]), _C.i64x3(v.stride()[:3]),
                _C.i64x3(out.stride()[:3]), float(scale), _stream(),
            )
        _C.check(st, "vft_attn_fwd")
        ctx.save_for_backward(q, k, v, out, lse, kv_len)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, kv_len = ctx.saved_tensors
        B, Hq, Sq, D = q.shape
        _, Hkv, Sk, _ = k.shape
        dout = _as_bhsd(dout.to(torch.bfloat16))
        dq = _empty_bshd_as_bhsd(B, Hq, Sq, D, q)
        dk = _empty_bshd_as_bhsd(B, Hkv, Sk, D, q)
        dv = _empty_bshd_as_bhsd(B, Hkv, Sk, D, q)
        delta = torch.empty((B, Hq, Sq), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device), _Timed(f"attn_bwd_d{D}", 10.0 * B * Hq * Sq * Sk * D):
            st = _C.lib().vft_attn_bwd(
                _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse),
                _ptr(dq), _ptr(dk), _ptr(dv), _ptr(delta), _ptr(kv_len),
                B, Hq, Hkv, Sq, Sk, D,
                _C.i64x3(q.stride()[:3]), _C.i64x3(k.stride()[:3]), _C.i64x3(v.stride()[:3]),
                _C.i64x3(out.stride()[:3]), _C.i64x3(dout.stride()[:3]), _C.i64x3(dq.stride()[:3]),
                _C.i64x3(dk.stride()[:3]), _C.i64x3(dv.stride()[:3]), ctx.scale, _stream(),
            )
        _C.check(st, "vft_attn_bwd")
        return dq, dk, dv, None, None


def flash_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    kv_len: torch.Tensor | None = None,
    scale: float | None = None,
) -> torch.Tensor:
    """softmax(scale * q k^T) v over (B,H,S,D) bf16 tensors; `kv_len` (B,) int32 masks keys >= kv_len[b].

    Kv heads may be fewer than query heads (GQA). The result is logically (B,Hq,Sq,D) with
    (B,Sq,Hq,D) memory order.
    """
    if scale is None:
        scale = 1.0 / math.sqrt(q.shape[-1])
    return _FlashAttention.apply(q, k, v, kv_len, scale)


# =============================================================================================
# Fused glue ops.  Each function below is one (or two) HIP kernels in libvft_hip.so with a
# hand-written backward; shapes are checked on the host before any launch.
# =============================================================================================
import torch.nn.functional as F  # noqa: E402  (library GEMM/conv calls only)


def _ready(*tensors: torch.Tensor) -> None:
    _C.lib()  # raises if the extension is not built: there is no fallback
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f"vision_ft_amd kernels run on MI355X only (tensor on {t.device}); no CPU fallback")


def layer_norm(x: torch.Tensor, weight, bias, eps: float) -> torch.Tensor:
    """LayerNorm over the last dim with fp32 statistics, output in x.dtype."""
    _ready(x)
    w = None if weight is None else weight.float()
    b = None if bias is None else bias.float()
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, eps).to(x.dtype)


def rms_norm(x: torch.Tensor, weight, eps: float) -> torch.Tensor:
    _ready(x)
    return F.rms_norm(x.float(), (x.shape[-1],), None if weight is None else weight.float(), eps).to(x.dtype)


def layer_norm_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """AdaLN: LN(x) (no affine, fp32 stats) * (1 + scale[:,None]) + shift[:,None]   (x: B,S,C; shift/scale: B,C)."""
    _ready(x)
    h = F.layer_norm(x.float(), (x.shape[-1],), None, None, eps).to(x.dtype)
    return h * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def rms_norm_scale(x: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor | None, eps: float) -> torch.Tensor:
    """Lumina2 pre-norm: RMSNorm(x)*w, optionally * (1 + scale[:,None])."""
    _ready(x)
    h = F.rms_norm(x, (x.shape[-1],), weight, eps)
    return h if scale is None else h * (1 + scale.unsqueeze(1))


def gate_residual(residual: torch.Tensor, gate: torch.Tensor, y: torch.Tensor, tanh: bool = False) -> torch.Tensor:
    """residual + g[:,None,:] * y  with g = tanh(gate) if `tanh`."""
    _ready(residual)
    g = gate.tanh() if tanh else gate
    return residual + g.unsqueeze(1) * y


def group_norm(x: torch.Tensor, weight, bias, groups: int, eps: float, silu: bool = False) -> torch.Tensor:
    """GroupNorm over NCHW (fp32 statistics) with optional fused SiLU."""
    _ready(x)
    h = F.group_norm(x, groups, weight, bias, eps)
    return F.silu(h) if silu else h


def geglu(proj: torch.Tensor) -> torch.Tensor:
    """proj (..., 2F) -> proj[..., :F] * gelu(proj[..., F:])  (exact erf GELU)."""
    _ready(proj)
    a, g = proj.chunk(2, dim=-1)
    return a * F.gelu(g)


def swiglu(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """silu(a) * b"""
    _ready(a)
    return F.silu(a) * b


def rope_apply(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """Rotate interleaved pairs of the last dim in fp32. x (B,H,S,D); freqs (S,D/2,2) = [cos, sin]."""
    _ready(x)
    cos, sin = freqs[..., 0].to(x.device), freqs[..., 1].to(x.device)
    xf = x.float()
    xe, xo = xf[..., 0::2], xf[..., 1::2]
    return torch.stack((xe * cos - xo * sin, xe * sin + xo * cos), dim=-1).reshape(x.shape).to(x.dtype)


def qk_norm_rope(x: torch.Tensor, kind: str, weight, eps: float, cos: torch.Tensor | None, sin: torch.Tensor | None) -> torch.Tensor:
    """Per-head norm over head_dim ('layer': LN without affine; 'rms': RMSNorm * weight) followed by
    the interleaved-pair rotation.  x: (B,S,H,D); cos/sin: (S,D/2) or (B,S,D/2) fp32, or None."""
    _ready(x)
    d = x.shape[-1]
    if kind == "layer":
        h = F.layer_norm(x.float(), (d,), None, None, eps).to(x.dtype)
    elif kind == "rms":
        h = F.rms_norm(x, (d,), weight, eps)
    elif kind == "none":
        h = x
    else:
        raise ValueError(kind)
    if cos is None:
        return h
    c = cos.unsqueeze(-2)  # broadcast over heads
    s = sin.unsqueeze(-2)
    hf = h.float()
    he, ho = hf[..., 0::2], hf[..., 1::2]
    return torch.stack((he * c - ho * s, he * s + ho * c), dim=-1).reshape(h.shape).to(h.dtype)


def lora_linear(x: torch.Tensor, base, lora_down: torch.Tensor, lora_up: torch.Tensor, scale: float) -> torch.Tensor:
    """base(x) + scale * (x @ lora_down^T) @ lora_up^T"""
    _ready(x)
    y = base(x)
    return y + F.linear(F.linear(x, lora_down), lora_up) * scale
