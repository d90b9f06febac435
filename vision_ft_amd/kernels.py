"""torch.autograd.Function wrappers over the C ABI (include/vft_hip.h).

PyTorch is used here for device memory, streams and autograd bookkeeping only; every op below
runs a hand-written HIP kernel from libvft_hip.so on ``torch.cuda.current_stream()``.
There is no CPU path: tensors that are not on a CUDA(HIP) device raise.
"""
from __future__ import annotations

import ctypes
import os
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
        if Hq == Hkv and Sq == Sk:
            # self-attention: one (B, S, 3, H, D) buffer, so that a fused q|k|v projection gets its output gradient as
            # a single (tokens, 3*H*D) matrix without a concatenation copy
            dqkv = torch.empty((B, Sq, 3, Hq, D), dtype=q.dtype, device=q.device)
            dq, dk, dv = (dqkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        else:
            dq = _empty_bshd_as_bhsd(B, Hq, Sq, D, q)
            dk = _empty_bshd_as_bhsd(B, Hkv, Sk, D, q)
            dv = _empty_bshd_as_bhsd(B, Hkv, Sk, D, q)
        ws_floats = int(_C.lib().vft_attn_bwd_workspace(B, Hq, Hkv, Sq, Sk, D))
        delta = torch.empty(ws_floats, dtype=torch.float32, device=q.device)  # delta (+ split-Q partials)
        with torch.cuda.device(q.device), _Timed(f"attn_bwd_d{D}", 10.0 * B * Hq * Sq * Sk * D):
            st = _C.lib().vft_attn_bwd(
                _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(dout), _ptr(lse),
                _ptr(dq), _ptr(dk), _ptr(dv), _ptr(delta), _ptr(kv_len),
                B, Hq, Hkv, Sq, Sk, D,
                _C.i64x3(q.stride()[:3]), _C.i64x3(k.stride()[:3]), _C.i64x3(v.stride()[:3]),
                _C.i64x3(out.stride()[:3]), _C.i64x3(dout.stride()[:3]), _C.i64x3(dq.stride()[:3]),
                _C.i64x3(dk.stride()[:3]), _C.i64x3(dv.stride()[:3]), ctx.scale, ws_floats, _stream(),
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


KIND_LAYER, KIND_RMS, KIND_NONE = 0, 1, 2

# LoRA layers over bf16 bases run on the fused projection GEMM (base + bias + adapter + skip in one launch per
# direction).  False restores the round-1 composition (library GEMM + in-place adapter passes) for A/B runs.
FUSED_LORA_GEMM = True


def _bf16_or_none(t):
    if t is None:
        return None
    return t if (t.dtype == torch.bfloat16 and t.is_contiguous()) else t.to(torch.bfloat16).contiguous()


class _RowNorm(torch.autograd.Function):
    """y = (norm(x) * w + b) * (1 + scale[b]) + shift[b] over the last dim (HIP: vft_norm_fwd / _bwd_dx).
    Weight / bias gradients (needed when the norm trains: the full fine-tune) come from vft_norm_bwd_params; the
    per-sample modulation gradients (no BASELINE config trains the modulation's producer) are reduced with torch ops
    from the saved statistics."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, shift, kind, eps, skip=False):
        C = x.shape[-1]
        x2 = _rows(x)
        rows = x2.shape[0]
        w, b_, sc, sh = (_bf16_or_none(t) for t in (weight, bias, scale, shift))
        rpb = rows // sc.shape[0] if sc is not None else (rows // sh.shape[0] if sh is not None else rows)
        y = torch.empty((rows, C), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if kind == KIND_LAYER else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _Timed("norm_fwd", 4.0 * rows * C):
            st = _C.lib().vft_norm_fwd(_ptr(x2), _ptr(y), _ptr(w), _ptr(b_), _ptr(sc), _ptr(sh), _ptr(mean), _ptr(rstd),
                                       rows, C, x2.stride(0), rpb, kind, float(eps), _stream())
        _C.check(st, "vft_norm_fwd")
        ctx.save_for_backward(x2, w, b_, sc, mean, rstd)
        ctx.meta = (kind, rpb, x.shape, weight, bias, scale, shift)
        ctx.skip = int(skip)
        if ctx.skip == 2:  # the block input feeds the norm AND two skip connections (AuraFlow's blocks)
            return y.view(x.shape), x, x.view_as(x)
        if skip:  # (normed, x): the second output carries the residual path, so that its gradient meets the
            return y.view(x.shape), x  # norm's own in ONE kernel instead of an autograd accumulation add
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dskip=None, dskip2=None):
        x2, w, b_, sc, mean, rstd = ctx.saved_tensors
        kind, rpb, xshape, weight, bias, scale, shift = ctx.meta
        rows, C = x2.shape

        def _res(d):
            if d is None:
                return None
            d = d.reshape(rows, C)
            if d.dtype != torch.bfloat16 or not d.is_contiguous():
                d = d.to(torch.bfloat16).contiguous()
            return d

        dres, dres2 = _res(dskip), _res(dskip2)
        if dres is None:
            dres, dres2 = dres2, None
        dy2 = dy.reshape(rows, C)
        if dy2.dtype != torch.bfloat16 or not dy2.is_contiguous():
            dy2 = dy2.to(torch.bfloat16).contiguous()
        dx = dw = db = dsc = dsh = None
        if ctx.needs_input_grad[0]:
            dx2 = torch.empty((rows, C), dtype=torch.bfloat16, device=dy.device)
            with torch.cuda.device(dy.device), _Timed("norm_bwd", 6.0 * rows * C):
                st = _C.lib().vft_norm_bwd_dx(_ptr(dy2), _ptr(x2), _ptr(w), _ptr(sc), _ptr(mean), _ptr(rstd), _ptr(dx2),
                                              _ptr(dres), _ptr(dres2), rows, C, x2.stride(0), rpb, kind, _stream())
            _C.check(st, "vft_norm_bwd_dx")
            dx = dx2.view(xshape)
        want_w = ctx.needs_input_grad[1] and weight is not None
        want_b = ctx.needs_input_grad[2] and bias is not None
        if want_w or want_b:
            # weight / bias gradients (the full fine-tune): one pass over dy and x, ordered partial sums
            pdt = (weight if want_w else bias).dtype
            f32 = pdt != torch.bfloat16
            odt = torch.float32 if f32 else torch.bfloat16
            # existing bf16 gradients (the trainer's flat buffer): accumulate in place, nothing left for autograd to add
            gw, gb = (_grad_target(weight) if want_w else None), (_grad_target(bias) if want_b else None)
            inplace = not f32 and (gw is not None or not want_w) and (gb is not None or not want_b)
            dwk = (gw if inplace else torch.empty(C, dtype=odt, device=dy.device)) if want_w else None
            dbk = (gb if inplace else torch.empty(C, dtype=odt, device=dy.device)) if want_b else None
            scratch = torch.empty(int(_C.lib().vft_norm_bwd_params_scratch_floats(rows, C)), dtype=torch.float32, device=dy.device)
            with torch.cuda.device(dy.device), _Timed("norm_bwd", 4.0 * rows * C):
                st = _C.lib().vft_norm_bwd_params(_ptr(dy2), _ptr(x2), _ptr(sc), _ptr(mean), _ptr(rstd), _ptr(dwk), _ptr(dbk),
                                                  _ptr(scratch), rows, C, x2.stride(0), rpb, kind, int(f32), int(inplace), _stream())
            _C.check(st, "vft_norm_bwd_params")
            if not inplace:
                dw = dwk.to(weight.dtype) if want_w else None
                db = dbk.to(bias.dtype) if want_b else None
        if any(ctx.needs_input_grad[3:5]):
            xf = x2.float()
            xh = (xf - mean[:, None]) * rstd[:, None] if kind == KIND_LAYER else xf * rstd[:, None]
            g = dy2.float()
            nb = rows // rpb
            if ctx.needs_input_grad[3] and scale is not None:
                base = xh if w is None else xh * w.float()
                if b_ is not None:
                    base = base + b_.float()
                dsc = (g * base).view(nb, rpb, C).sum(1).to(scale.dtype).view(scale.shape)
            if ctx.needs_input_grad[4] and shift is not None:
                dsh = g.view(nb, rpb, C).sum(1).to(shift.dtype).view(shift.shape)
        return dx, dw, db, dsc, dsh, None, None, None


def _norm(x, weight, bias, scale, shift, kind, eps, skip=False):
    _ready(x)
    _require_gpu(x)
    return _RowNorm.apply(x, weight, bias, scale, shift, kind, eps, skip)


def layer_norm_skip(x: torch.Tensor, weight, bias, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """(LayerNorm(x), x): use the second output as the residual input of the branch's last layer
    (`lora_linear(..., residual=x)` or `x + branch`); the backward then adds the residual gradient inside the
    norm-backward kernel."""
    return _norm(x, weight, bias, None, None, KIND_LAYER, eps, skip=True)


def layer_norm(x: torch.Tensor, weight, bias, eps: float) -> torch.Tensor:
    """LayerNorm over the last dim with fp32 statistics, output in x.dtype."""
    return _norm(x, weight, bias, None, None, KIND_LAYER, eps)


def rms_norm(x: torch.Tensor, weight, eps: float) -> torch.Tensor:
    return _norm(x, weight, None, None, None, KIND_RMS, eps)


def layer_norm_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, eps: float = 1e-5, skips: int = 0):
    """AdaLN: LN(x) (no affine, fp32 stats) * (1 + scale[:,None]) + shift[:,None]   (x: B,S,C; shift/scale: B,C).
    skips = 1 / 2: also returns x once / twice — use those outputs as the residual inputs of the block's gated adds, so
    that their gradients are added inside the norm-backward kernel instead of by autograd accumulation passes."""
    return _norm(x, None, None, scale, shift, KIND_LAYER, eps, skip=skips)


def rms_norm_scale(x: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor | None, eps: float) -> torch.Tensor:
    """Lumina2 pre-norm: RMSNorm(x)*w, optionally * (1 + scale[:,None])."""
    return _norm(x, weight, None, scale, None, KIND_RMS, eps)


def rms_norm_scale_skip(x: torch.Tensor, weight: torch.Tensor, scale: torch.Tensor | None, eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """(rms_norm_scale(x, ...), x): the second output carries the block's residual path (Lumina2: `x + tanh(g) * norm2(branch(norm1(x)))`,
    src/models/lumina2/denoiser.py:216-258), so that its gradient meets the norm's own inside the norm-backward kernel instead
    of in an autograd accumulation add (one elementwise pass over the hidden state per norm)."""
    return _norm(x, weight, None, scale, None, KIND_RMS, eps, skip=True)


class _SplitLast(torch.autograd.Function):
    """x[..., a:b] views of a fused projection output (Lumina2's q | k | v, src/models/lumina2/denoiser.py:96-112) as ONE autograd
    node: the backward writes the pieces' gradients side by side in one pass (a concatenation), where three slice nodes each
    embed their gradient in a zero-filled full-size tensor and autograd adds the three (3 fills + 3 copies + 2 adds of the
    (tokens, 3840) matrix per layer)."""

    @staticmethod
    def forward(ctx, x, *sizes):
        ctx.sizes = sizes
        ctx.shape, ctx.dtype = x.shape, x.dtype
        return tuple(x.split(list(sizes), dim=-1))

    @staticmethod
    def backward(ctx, *grads):
        parts = []
        for g, n in zip(grads, ctx.sizes):
            if g is None:
                g = torch.zeros((*ctx.shape[:-1], n), dtype=ctx.dtype, device=next(t for t in grads if t is not None).device)
            parts.append(g.to(ctx.dtype))
        return (torch.cat(parts, dim=-1), *([None] * len(ctx.sizes)))


def split_last(x: torch.Tensor, sizes) -> tuple[torch.Tensor, ...]:
    """Views x[..., :s0], x[..., s0:s0+s1], ... whose gradients come back as one concatenated tensor (see `_SplitLast`)."""
    return _SplitLast.apply(x, *[int(n) for n in sizes])


class _GateResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, residual, gate, y, use_tanh):
        B, S, C = y.shape
        res2, y2, g = residual.contiguous(), y.contiguous(), _bf16_or_none(gate)
        out = torch.empty_like(y2)
        with torch.cuda.device(y.device), _Timed("gate_residual", 6.0 * B * S * C):
            st = _C.lib().vft_gate_residual(_ptr(res2), _ptr(g), _ptr(y2), _ptr(out), B * S, C, S, int(use_tanh), _stream())
        _C.check(st, "vft_gate_residual")
        ctx.save_for_backward(g, y2)
        ctx.use_tanh = use_tanh
        ctx.gate_ref = gate
        return out

    @staticmethod
    def backward(ctx, dout):
        g, y2 = ctx.saved_tensors
        B, S, C = y2.shape
        d2 = dout.contiguous()
        dres = dout if ctx.needs_input_grad[0] else None
        dy = dgate = None
        if ctx.needs_input_grad[2]:
            dy = torch.empty_like(y2)
            with torch.cuda.device(dout.device), _Timed("gate_residual_bwd", 4.0 * B * S * C):
                st = _C.lib().vft_gate_residual(None, _ptr(g), _ptr(d2), _ptr(dy), B * S, C, S, int(ctx.use_tanh), _stream())
            _C.check(st, "vft_gate_residual(bwd)")
        if ctx.needs_input_grad[1]:
            dg = (d2.float() * y2.float()).sum(1)
            if ctx.use_tanh:
                dg = dg * (1 - torch.tanh(g.float()) ** 2)
            dgate = dg.to(ctx.gate_ref.dtype)
        return dres, dgate, dy, None


def clip_flat_grad_norm_(buffers: list[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_(params, max_norm) (norm type 2) when every gradient is a view into a few flat bf16 buffers
    (trainer/dp.py): total = sqrt(sum of squares over the buffers) on vft_sum_sq_bf16 — one pass per buffer instead of a
    multi-tensor pass over every gradient view, fp32 throughout, no ATen reduction — then
    buffers *= min(1, max_norm / (total + 1e-6)).  Everything stays on the device (capturable).  Returns the total norm."""
    dev = buffers[0].device
    tot = torch.empty((), dtype=torch.float32, device=dev)
    scratch = torch.empty(int(_C.lib().vft_mean_sq_scratch_floats()), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for i, b in enumerate(buffers):
            _require_gpu(b)
            st = _C.lib().vft_sum_sq_bf16(_ptr(b), _ptr(tot), _ptr(scratch), b.numel(), int(i > 0), _stream())
            _C.check(st, "vft_sum_sq_bf16")
    norm = tot.sqrt()
    coef = (float(max_norm) / (norm + 1e-6)).clamp(max=1.0)
    for b in buffers:
        b.mul_(coef)  # a 0-dim fp32 factor: the product is formed in fp32 and rounded once, like torch's clip_grad_norm_
    return norm


class _MeanSq(torch.autograd.Function):
    """mean(d^2) of a dense fp32 tensor (vft_mean_sq); backward 2 d g / n."""

    @staticmethod
    def forward(ctx, d):
        out = torch.empty((), dtype=torch.float32, device=d.device)
        scratch = torch.empty(int(_C.lib().vft_mean_sq_scratch_floats()), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device), _Timed("mean_sq", 4.0 * d.numel()):
            st = _C.lib().vft_mean_sq(_ptr(d), _ptr(out), _ptr(scratch), d.numel(), _stream())
        _C.check(st, "vft_mean_sq")
        ctx.save_for_backward(d)
        return out

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * (g * (2.0 / d.numel()))


def mse_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """F.mse_loss(pred.float(), target.float(), reduction="mean") — the reference's loss reduction
    (src/modules/loss/diffusion.py:62-68, loss/flow_match.py:85-96) — with the reduction on the two-launch, atomics-free HIP
    kernel (see csrc/eltwise.hip: ATen's one-launch reduction misbehaved inside replayed HIP graphs).  The difference is an ATen
    elementwise op (any dense layout: a sum does not care about the order); gradients flow through it as usual."""
    _require_gpu(pred, target, dtype=None)
    d = pred.float() - target.float()
    if not (d.is_contiguous() or d.is_contiguous(memory_format=torch.channels_last) or d.is_contiguous(memory_format=torch.channels_last_3d)):
        d = d.contiguous()
    return _MeanSq.apply(d)


def gate_residual(residual: torch.Tensor, gate: torch.Tensor, y: torch.Tensor, tanh: bool = False) -> torch.Tensor:
    """residual + g[:,None,:] * y  with g = tanh(gate) if `tanh`."""
    _ready(residual)
    _require_gpu(residual, y)
    return _GateResidual.apply(residual, gate, y, tanh)


def _is_nhwc(x: torch.Tensor) -> bool:
    """True when a (N,C,H,W) tensor is stored channels-last (and that is not also the NCHW order)."""
    return x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


class _GroupNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, groups, eps, silu):
        N, C, H, W = x.shape
        nhwc = _is_nhwc(x) and C % 8 == 0 and C <= 2560
        if not nhwc:
            x = x.contiguous()
        w, b_ = _bf16_or_none(weight), _bf16_or_none(bias)
        y = torch.empty_like(x)  # preserves the memory format
        stats = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
        nscr = _C.lib().vft_groupnorm_scratch_floats(N, C, H * W, groups, int(nhwc), 0)
        if nscr < 0:
            _C.check(1, "vft_groupnorm_scratch_floats")
        scratch = torch.empty(nscr, dtype=torch.float32, device=x.device)  # partial sums, folded in a fixed order
        fn = _C.lib().vft_groupnorm_nhwc_fwd if nhwc else _C.lib().vft_groupnorm_fwd
        with torch.cuda.device(x.device), _Timed("groupnorm_fwd", 6.0 * x.numel()):
            st = fn(_ptr(x), _ptr(y), _ptr(w), _ptr(b_), _ptr(stats), _ptr(scratch), N, C, H * W, groups, float(eps), int(silu),
                    _stream())
        _C.check(st, "vft_groupnorm_fwd")
        ctx.save_for_backward(x, w, b_, stats)
        ctx.meta = (groups, eps, silu, weight, bias, nhwc)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b_, stats = ctx.saved_tensors
        groups, eps, silu, weight, bias, nhwc = ctx.meta
        N, C, H, W = x.shape
        dy = dy.contiguous(memory_format=torch.channels_last) if nhwc else dy.contiguous()
        dx = torch.empty_like(x)
        bstats = torch.empty(N * groups * 2, dtype=torch.float32, device=x.device)
        need_w = ctx.needs_input_grad[1] and weight is not None
        need_b = ctx.needs_input_grad[2] and bias is not None
        dwf = torch.empty(C, dtype=torch.float32, device=x.device) if need_w else None
        dbf = torch.empty(C, dtype=torch.float32, device=x.device) if need_b else None
        nscr = _C.lib().vft_groupnorm_scratch_floats(N, C, H * W, groups, int(nhwc), int(need_w or need_b))
        scratch = torch.empty(max(nscr, 1), dtype=torch.float32, device=x.device)
        fn = _C.lib().vft_groupnorm_nhwc_bwd if nhwc else _C.lib().vft_groupnorm_bwd
        with torch.cuda.device(x.device), _Timed("groupnorm_bwd", 10.0 * x.numel()):
            st = fn(_ptr(x), _ptr(dy), _ptr(stats), _ptr(w), _ptr(b_), _ptr(dx), _ptr(bstats), _ptr(scratch), _ptr(dwf),
                    _ptr(dbf), N, C, H * W, groups, float(eps), int(silu), _stream())
        _C.check(st, "vft_groupnorm_bwd")
        dw = dwf.to(weight.dtype) if need_w else None
        db = dbf.to(bias.dtype) if need_b else None
        return dx, dw, db, None, None, None


def group_norm(x: torch.Tensor, weight, bias, groups: int, eps: float, silu: bool = False) -> torch.Tensor:
    """GroupNorm over (N,C,H,W) in NCHW or channels-last storage (fp32 statistics), optional fused SiLU;
    the output keeps the input's memory format."""
    _ready(x)
    _require_gpu(x)
    return _GroupNorm.apply(x, weight, bias, groups, eps, silu)


class _GeGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, proj):
        F2 = proj.shape[-1]
        p2 = _rows(proj)
        if p2.stride(0) != F2:
            p2 = p2.contiguous()
        M, Fh = p2.shape[0], F2 // 2
        out = torch.empty((M, Fh), dtype=torch.bfloat16, device=proj.device)
        with torch.cuda.device(proj.device), _Timed("geglu_fwd", 6.0 * M * Fh):
            st = _C.lib().vft_geglu_fwd(_ptr(p2), _ptr(out), M, Fh, _stream())
        _C.check(st, "vft_geglu_fwd")
        ctx.save_for_backward(p2)
        ctx.pshape = proj.shape
        return out.view(*proj.shape[:-1], Fh)

    @staticmethod
    def backward(ctx, dy):
        (p2,) = ctx.saved_tensors
        M, F2 = p2.shape
        d2 = dy.reshape(M, F2 // 2).contiguous()
        dproj = torch.empty_like(p2)
        with torch.cuda.device(dy.device), _Timed("geglu_bwd", 10.0 * M * (F2 // 2)):
            st = _C.lib().vft_geglu_bwd(_ptr(p2), _ptr(d2), _ptr(dproj), M, F2 // 2, _stream())
        _C.check(st, "vft_geglu_bwd")
        return dproj.view(ctx.pshape)


def geglu(proj: torch.Tensor) -> torch.Tensor:
    """proj (..., 2F) -> proj[..., :F] * gelu(proj[..., F:])  (exact erf GELU)."""
    _ready(proj)
    _require_gpu(proj)
    return _GeGLU.apply(proj)


class _SwiGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = a.contiguous(), b.contiguous()
        out = torch.empty_like(a2)
        with torch.cuda.device(a.device), _Timed("swiglu_fwd", 6.0 * a2.numel()):
            st = _C.lib().vft_swiglu_fwd(_ptr(a2), _ptr(b2), _ptr(out), a2.numel(), _stream())
        _C.check(st, "vft_swiglu_fwd")
        ctx.save_for_backward(a2, b2)
        return out

    @staticmethod
    def backward(ctx, dy):
        a2, b2 = ctx.saved_tensors
        d2 = dy.contiguous()
        da, db = torch.empty_like(a2), torch.empty_like(b2)
        with torch.cuda.device(dy.device), _Timed("swiglu_bwd", 10.0 * a2.numel()):
            st = _C.lib().vft_swiglu_bwd(_ptr(a2), _ptr(b2), _ptr(d2), _ptr(da), _ptr(db), a2.numel(), _stream())
        _C.check(st, "vft_swiglu_bwd")
        return da, db


def swiglu(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """silu(a) * b"""
    _ready(a, b)
    _require_gpu(a, b)
    return _SwiGLU.apply(a, b)


class _QkNormRope(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, cos, sin, kind, eps):
        B, S, H, D = x.shape
        if x.stride(3) != 1 or any(st % 8 for st in x.stride()[:3]) or x.data_ptr() % 16:
            x = x.contiguous()
        w = _bf16_or_none(weight)
        cs_b = 0
        if cos is not None:
            cos, sin = cos.float().contiguous(), sin.float().contiguous()
            if cos.dim() == 3:
                cs_b = cos.stride(0)
        y = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty(B * S * H, dtype=torch.float32, device=x.device) if kind == KIND_LAYER else None
        rstd = torch.empty(B * S * H, dtype=torch.float32, device=x.device) if kind != KIND_NONE else None
        with torch.cuda.device(x.device), _Timed("qk_norm_rope_fwd", 4.0 * x.numel()):
            st = _C.lib().vft_qk_norm_rope_fwd(_ptr(x), _ptr(y), _ptr(w), _ptr(cos), _ptr(sin), _ptr(mean), _ptr(rstd),
                                               B, S, H, D, _C.i64x3(x.stride()[:3]), cs_b, kind, float(eps), _stream())
        _C.check(st, "vft_qk_norm_rope_fwd")
        ctx.save_for_backward(x, w, cos, sin, mean, rstd)
        ctx.meta = (kind, cs_b, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, cos, sin, mean, rstd = ctx.saved_tensors
        kind, cs_b, weight = ctx.meta
        B, S, H, D = x.shape
        d2 = dy  # strided views (dq / dk slices of the attention backward's fused dq|dk|dv buffer) are read in place
        if dy.stride(3) != 1 or any(st % 8 for st in dy.stride()[:3]) or dy.data_ptr() % 16 or dy.dtype != torch.bfloat16:
            d2 = dy.to(torch.bfloat16).contiguous()
        dx = torch.empty((B, S, H, D), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device), _Timed("qk_norm_rope_bwd", 6.0 * x.numel()):
            st = _C.lib().vft_qk_norm_rope_bwd(_ptr(d2), _ptr(x), _ptr(w), _ptr(cos), _ptr(sin), _ptr(mean), _ptr(rstd),
                                               _ptr(dx), B, S, H, D, _C.i64x3(x.stride()[:3]), _C.i64x3(d2.stride()[:3]),
                                               cs_b, kind, _stream())
        _C.check(st, "vft_qk_norm_rope_bwd")
        dw = None
        if ctx.needs_input_grad[1] and weight is not None:
            g = d2.float()
            if cos is not None:
                c = cos.unsqueeze(-2) if cos.dim() == 3 else cos.unsqueeze(0).unsqueeze(-2)
                s_ = sin.unsqueeze(-2) if sin.dim() == 3 else sin.unsqueeze(0).unsqueeze(-2)
                ge, go = g[..., 0::2], g[..., 1::2]
                g = torch.stack((ge * c + go * s_, -ge * s_ + go * c), dim=-1).reshape(g.shape)
            xh = x.float() * rstd.view(B, S, H, 1)
            dw = (g * xh).sum((0, 1, 2)).to(weight.dtype)
        return dx, dw, None, None, None, None


_QK_KINDS = {"layer": KIND_LAYER, "rms": KIND_RMS, "none": KIND_NONE}


def qk_norm_rope(x: torch.Tensor, kind: str, weight, eps: float, cos: torch.Tensor | None, sin: torch.Tensor | None) -> torch.Tensor:
    """Per-head norm over head_dim ('layer': LN without affine; 'rms': RMSNorm * weight; 'none') followed
    by the interleaved-pair rotation, one fused kernel.  x: (B,S,H,D) (any 8-aligned strides);
    cos/sin: (S,D/2) or (B,S,D/2) fp32, or None.  Returns contiguous (B,S,H,D)."""
    _ready(x)
    _require_gpu(x)
    if kind not in _QK_KINDS:
        raise ValueError(f"unknown qk norm kind {kind!r}")
    if x.shape[-1] not in (64, 96, 128, 256):
        raise ValueError(f"head_dim {x.shape[-1]} unsupported (64, 96, 128, 256)")
    if cos is not None and cos.shape[-1] * 2 != x.shape[-1]:
        raise ValueError("rope table must have head_dim/2 columns")
    return _QkNormRope.apply(x, weight, cos, sin, _QK_KINDS[kind], eps)


def rope_apply(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """Rotate interleaved pairs of the last dim in fp32. x (B,H,S,D); freqs (S,D/2,2) = [cos, sin]."""
    f = freqs.to(x.device)
    out = qk_norm_rope(x.transpose(1, 2), "none", None, 0.0, f[..., 0], f[..., 1])
    return out.transpose(1, 2)


def _rows(x: torch.Tensor) -> torch.Tensor:
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0 or x2.data_ptr() % 16 != 0:
        x2 = x2.contiguous()
    return x2


def _lora_fwd_(x2: torch.Tensor, y2: torch.Tensor, A: torch.Tensor, B: torch.Tensor, s: float, residual: torch.Tensor | None = None):
    """y2 += s (x2 Aᵀ) Bᵀ (+ residual) in place; returns (t = x2 Aᵀ, Bᵀ) for backward."""
    M, K = x2.shape
    N, r = B.shape
    t = torch.empty((M, r), dtype=torch.bfloat16, device=x2.device)
    Bt = torch.empty((r, N), dtype=torch.bfloat16, device=x2.device)
    with _Timed("lora_fwd", 2.0 * M * K + 4.0 * M * N):
        st = _C.lib().vft_lora_fwd(_ptr(x2), _ptr(y2), _ptr(A), _ptr(B), _ptr(t), _ptr(Bt), _ptr(residual), M, K, N, r,
                                   x2.stride(0), y2.stride(0), residual.stride(0) if residual is not None else 0,
                                   float(s), _stream())
    _C.check(st, "vft_lora_fwd")
    return t, Bt


def _lora_bwd_dx_(dy2: torch.Tensor, dx2: torch.Tensor, Bt: torch.Tensor, A: torch.Tensor, s: float) -> torch.Tensor:
    """dx2 += s (dy2 B) A in place; returns u = dy2 B."""
    M, N = dy2.shape
    r, K = A.shape
    u = torch.empty((M, r), dtype=torch.bfloat16, device=dy2.device)
    with _Timed("lora_bwd_dx", 2.0 * M * N + 4.0 * M * K):
        st = _C.lib().vft_lora_bwd_dx(_ptr(dy2), _ptr(dx2), _ptr(Bt), _ptr(A), _ptr(u), M, N, K, r,
                                      dy2.stride(0), dx2.stride(0), float(s), _stream())
    _C.check(st, "vft_lora_bwd_dx")
    return u


def _lora_wgrad2(x2, u, dy2, t, A, B, s: float):
    M, K = x2.shape
    N, r = B.shape
    dA, dB = torch.empty_like(A), torch.empty_like(B)
    nbytes = _C.lib().vft_lora_wgrad_workspace(M, K, N, r)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x2.device)
    with _Timed("lora_wgrad2", 2.0 * M * (K + N)):
        st = _C.lib().vft_lora_wgrad2(_ptr(x2), _ptr(u), _ptr(dy2), _ptr(t), _ptr(dA), _ptr(dB), _ptr(ws), nbytes,
                                      M, K, N, r, x2.stride(0), dy2.stride(0), float(s), _stream())
    _C.check(st, "vft_lora_wgrad2")
    return dA, dB


# How often the adapter layers found their operand forms in the model's AdapterPack / wrote weight gradients in place
# (host-side counters: tests and the bench check that the fast paths are the ones that run)
PACK_STATS = {"pack_hit": 0, "pack_miss": 0, "grad_in_place": 0, "grad_returned": 0}


def _grad_target(p: torch.Tensor | None):
    """The tensor a weight gradient may be accumulated into directly: the parameter's existing dense bf16 `.grad` (the
    trainer points it into the flat gradient buffer).  None -> the gradient has to be returned to autograd."""
    g = getattr(p, "grad", None) if (p is not None and p.is_leaf) else None  # (a non-leaf — a padded copy — has no .grad to add into)
    if g is None or g.dtype != torch.bfloat16 or not g.is_contiguous() or not g.is_cuda or g.shape != p.shape:
        return None
    return g


class _WgradQueue:
    """Deferred LoRA weight-gradient reductions of one backward pass (per device).  Each layer's partial kernel writes its
    two reductions into the next slots of a persistent device table and its partial sums into a slice of a persistent
    arena; ONE `vft_lora_wgrad_reduce_many` launch — queued as an autograd engine callback, i.e. run when the backward pass
    that produced them ends — folds them all into the gradient tensors.  Everything is launch-only (no host-to-device
    copy), so the whole mechanism is captured into the trainer's HIP graphs like any other kernel."""

    CAP = 8192  # table entries (two per adapted layer call)
    queues: dict = {}

    def __init__(self, device):
        self.device = device
        self.desc_bytes = int(_C.lib().vft_lora_wgrad_desc_bytes())
        self.table = torch.zeros(self.CAP * self.desc_bytes, dtype=torch.uint8, device=device)
        self.arena = torch.empty(1 << 24, dtype=torch.float32, device=device)  # 64 MB to start with; grows outside capture
        self.retired: list = []   # outgrown arenas: kept for good — reductions queued earlier in this pass, and any HIP graph
                                  # captured before the growth, still address them (sizes double, so at most as much again)
        self.off = 0
        self.count = 0
        self.max_ne = 0
        self.callback_queued = False
        self.pending: set[int] = set()  # data_ptr of every gradient tensor with a reduction queued in this pass

    @classmethod
    def get(cls, device) -> "_WgradQueue":
        q = cls.queues.get(device)
        if q is None:
            q = cls.queues[device] = cls(device)
        return q

    def take(self, nfloats: int):
        """A slice of the arena for one layer's partial sums, or None when it cannot be had (table full, or the arena
        would have to grow while a HIP graph is being captured): the caller then reduces immediately."""
        nfloats = (nfloats + 63) // 64 * 64
        if self.count + 2 > self.CAP:
            return None
        if self.off + nfloats > self.arena.numel():
            if torch.cuda.is_current_stream_capturing():
                return None
            self.retired.append(self.arena)
            self.arena = torch.empty(max(2 * self.arena.numel(), 2 * nfloats), dtype=torch.float32, device=self.device)
            self.off = 0
        ws = self.arena[self.off: self.off + nfloats]
        self.off += nfloats
        return ws

    def flush(self) -> None:
        self.callback_queued = False
        if self.count:
            with torch.cuda.device(self.device):
                st = _C.lib().vft_lora_wgrad_reduce_many(_ptr(self.table), self.count, self.max_ne, _stream())
            self.count, self.off, self.max_ne = 0, 0, 0
            self.pending.clear()
            _C.check(st, "vft_lora_wgrad_reduce_many")

    def claim(self, targets) -> None:
        """Register the gradient tensors of one layer call.  The single reduce launch gives every table entry its own
        workgroups and each does `grad += sum`, non-atomically: two entries with the SAME target (one adapter applied twice
        in a forward pass, or two forward passes summed into one loss) would race and lose a contribution.  A repeated
        target therefore flushes what is queued first — launches on one stream are ordered, so the sums add up."""
        ptrs = {g.data_ptr() for pair in targets for g in pair}
        if self.pending & ptrs:
            self.flush()
        self.pending |= ptrs


DEFER_WGRAD_REDUCE = os.environ.get("VFT_DEFER_WGRAD", "1") != "0"  # A/B switch


def flush_deferred_wgrads() -> None:
    """Run the queued weight-gradient reductions now (normally the autograd engine does, at the end of backward)."""
    for q in _WgradQueue.queues.values():
        q.flush()


def _lora_wgrad_into(x2, u, dy2, t, params, r: int, rp: int, s: float):
    """Weight gradients of `len(params)` adapters that share x2 — params = [(lora_down, lora_up)], 1 or 3 (fused q|k|v);
    u / t (M, rp) carry their rank-r columns side by side, dy2 their output columns side by side.

    When every parameter already has a gradient tensor the kernel ADDS into it (what AccumulateGrad would do with a
    returned tensor — minus one elementwise launch per parameter) and the function returns None for each; inside a
    backward pass the reduction itself is deferred to ONE launch at the end of the pass (`_WgradQueue`).  Otherwise the
    gradients are returned: [dA_0, dB_0, dA_1, ...].  (With gradient tensors in place `torch.autograd.grad` on these
    parameters would therefore see None: the trainers use `backward()`.)"""
    G = len(params)
    M, K = x2.shape
    N = dy2.shape[1]
    targets = [(_grad_target(a), _grad_target(b)) for a, b in params]
    inplace = all(ga is not None and gb is not None for ga, gb in targets)
    PACK_STATS["grad_in_place" if inplace else "grad_returned"] += G
    if not inplace:
        targets = [(torch.empty((r, K), dtype=torch.bfloat16, device=x2.device),
                    torch.empty((N // G, r), dtype=torch.bfloat16, device=x2.device)) for _ in range(G)]
    nbytes = _C.lib().vft_lora_wgrad_workspace(M, K, N, rp)
    dd = (ctypes.c_void_p * 3)(*[t_[0].data_ptr() for t_ in targets])
    du = (ctypes.c_void_p * 3)(*[t_[1].data_ptr() for t_ in targets])
    if inplace and DEFER_WGRAD_REDUCE:
        q = _WgradQueue.get(x2.device)
        q.claim(targets)
        ws = q.take(max(nbytes // 4, 1))
        if ws is not None:
            with _Timed("lora_wgrad2", 2.0 * M * (K + N)):
                st = _C.lib().vft_lora_wgrad2g_defer(_ptr(x2), _ptr(u), _ptr(dy2), _ptr(t), dd, du, G, r, 1, _ptr(ws), nbytes,
                                                     M, K, N, rp, x2.stride(0), dy2.stride(0), float(s), _ptr(q.table), q.count,
                                                     _stream())
            _C.check(st, "vft_lora_wgrad2g_defer")
            q.count += 2
            q.max_ne = max(q.max_ne, K * rp, N * rp)
            if not q.callback_queued:
                try:
                    torch.autograd.Variable._execution_engine.queue_callback(q.flush)
                    q.callback_queued = True
                except RuntimeError:  # not inside a backward pass: nothing will call back
                    q.flush()
            return [None] * (2 * G)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=x2.device)
    with _Timed("lora_wgrad2", 2.0 * M * (K + N)):
        st = _C.lib().vft_lora_wgrad2g(_ptr(x2), _ptr(u), _ptr(dy2), _ptr(t), dd, du, G, r, 1 if inplace else 0, _ptr(ws), nbytes,
                                       M, K, N, rp, x2.stride(0), dy2.stride(0), float(s), _stream())
    _C.check(st, "vft_lora_wgrad2g")
    if inplace:
        return [None] * (2 * G)
    return [g for pair in targets for g in pair]


# Round 5: the backward's u = s dy B comes out of the weight-gradient pass over dy instead of its own pass (vft_lora_bwd_dy /
# vft_lora_bwd_x: dy and x are read once each for the rank-sized work).  A/B switch; also the path when adapters are frozen.
LORA_BWD_ONE_READ = os.environ.get("VFT_LORA_BWD_ONE_READ", "1") != "0"


def _lora_bwd_rank(x2, dy2, t, Bt, params, r: int, rp: int, s: float):
    """(u, grads) of one LoRA-adapted projection call (or a fused group of up to three that share x2): u (M, rp) = s dy B for
    the data gradient's rank extension, grads as `_lora_wgrad_into` returns them.  `Bt` = lora_up transposed, rank-padded
    (rp, N) (block-diagonal for a group).  One pass over dy gives u and d(lora_up), one pass over x gives d(lora_down)."""
    G = len(params)
    M, K = x2.shape
    N = dy2.shape[1]
    targets = [(_grad_target(a), _grad_target(b)) for a, b in params]
    inplace = all(ga is not None and gb is not None for ga, gb in targets)
    if not (LORA_BWD_ONE_READ and rp in (4, 8, 16) and Bt.is_contiguous() and Bt.dtype == torch.bfloat16 and dy2.stride(1) == 1):
        u = lora_down(dy2, Bt, s)
        return u, _lora_wgrad_into(x2, u, dy2, t, params, r, rp, 1.0)
    PACK_STATS["grad_in_place" if inplace else "grad_returned"] += G
    if not inplace:
        targets = [(torch.empty((r, K), dtype=torch.bfloat16, device=x2.device),
                    torch.empty((N // G, r), dtype=torch.bfloat16, device=x2.device)) for _ in range(G)]
    lib = _C.lib()
    dd = (ctypes.c_void_p * 3)(*[t_[0].data_ptr() for t_ in targets])
    du = (ctypes.c_void_p * 3)(*[t_[1].data_ptr() for t_ in targets])
    u = torch.empty((M, rp), dtype=torch.bfloat16, device=x2.device)
    u_part = torch.empty(int(lib.vft_lora_u_part_floats(M, N, rp)), dtype=torch.float32, device=x2.device)
    nb_dy, nb_x = int(lib.vft_lora_bwd_workspace(M, N, rp, 1)), int(lib.vft_lora_bwd_workspace(M, K, rp, 0))
    q = table = None
    ws_dy = ws_x = None
    if inplace and DEFER_WGRAD_REDUCE:
        q = _WgradQueue.get(x2.device)
        q.claim(targets)
        ws_dy = q.take(max(nb_dy // 4, 1))
        ws_x = q.take(max(nb_x // 4, 1)) if ws_dy is not None else None
        if ws_x is None:
            q = None
    if q is None:
        ws_dy = torch.empty(max(nb_dy // 4, 1), dtype=torch.float32, device=x2.device)
        ws_x = torch.empty(max(nb_x // 4, 1), dtype=torch.float32, device=x2.device)
    acc = 1 if inplace else 0
    with _Timed("lora_wgrad2", 2.0 * M * (K + N)):
        st = lib.vft_lora_bwd_dy(_ptr(dy2), _ptr(t), _ptr(Bt), _ptr(u), _ptr(u_part), du, G, r, acc, _ptr(ws_dy), nb_dy, M, N, rp,
                                 dy2.stride(0), float(s), 1.0, _ptr(q.table) if q is not None else None, q.count if q is not None else 0,
                                 _stream())
        _C.check(st, "vft_lora_bwd_dy")
        st = lib.vft_lora_bwd_x(_ptr(x2), _ptr(u), dd, G, r, acc, _ptr(ws_x), nb_x, M, K, rp, x2.stride(0), 1.0,
                                _ptr(q.table) if q is not None else None, (q.count + 1) if q is not None else 0, _stream())
        _C.check(st, "vft_lora_bwd_x")
    if q is not None:
        q.count += 2
        q.max_ne = max(q.max_ne, K * rp, N * rp)
        if not q.callback_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(q.flush)
                q.callback_queued = True
            except RuntimeError:  # not inside a backward pass: nothing will call back
                q.flush()
    if inplace:
        return u, [None] * (2 * G)
    return u, [g for pair in targets for g in pair]


class AdapterPack:
    """Operand forms of every LoRA adapter of a model that the fused projection GEMM consumes — lora_up transposed
    (for u = dy B), lora_down transposed (the data gradient's rank extension), the concatenated / block-diagonal /
    zero-padded forms of a fused q|k|v — kept in one flat buffer and rebuilt by ONE kernel launch (`vft_lora_pack`) at
    the start of each forward of the model, instead of ~4 tiny tensor ops per layer and step.

    Entries are created by `modules.peft.functional.build_adapter_pack`; a layer uses its entry only while the entry's
    snapshot of the parameters' version counters is current (i.e. the pack was refreshed after the last optimizer
    step), otherwise it falls back to building the forms itself."""

    def __init__(self, device: torch.device):
        self.device = device
        self.entries: list[dict] = []
        self._descs: list[tuple] = []  # (srcs, name, kind, groups, r, rp, L) per derived tensor
        self._table = None
        self._flat = None
        self._ptrs: tuple = ()

    def add(self, params: list[tuple[torch.Tensor, torch.Tensor]], rp: int) -> dict:
        """params = [(lora_down (r,K), lora_up (N,r))] * groups (all the same shape).  Returns the entry, whose tensors
        `A` (rp,K), `B` (groups*N, rp), `At` (K,rp), `Bt` (rp, groups*N) exist after `finalize()`."""
        (A0, B0), G = params[0], len(params)
        r, K = A0.shape
        N = B0.shape[0]
        e = {"params": params, "r": r, "rp": rp, "G": G, "K": K, "N": N, "ver": None,
             "alias": G == 1 and rp == r}  # then A / B are the parameters themselves
        self.entries.append(e)
        return e

    def finalize(self) -> None:
        total, plan = 0, []
        for e in self.entries:
            G, rp, K, N = e["G"], e["rp"], e["K"], e["N"]
            shapes = {"At": ((K, rp), 1), "Bt": ((rp, G * N), 3)}
            if not e["alias"]:
                shapes.update({"A": ((rp, K), 0), "B": ((G * N, rp), 2)})
            for name, (shape, kind) in shapes.items():
                n = shape[0] * shape[1]
                plan.append((e, name, shape, kind, total))
                total += (n + 7) // 8 * 8  # 16-byte aligned starts
        self._flat = torch.zeros(total, dtype=torch.bfloat16, device=self.device)
        self._plan = plan
        for e, name, shape, kind, off in plan:
            e[name] = self._flat[off: off + shape[0] * shape[1]].view(shape)
        self._build_table()

    def _src_ptrs(self) -> tuple:
        return tuple(p.data_ptr() for e in self.entries for pair in e["params"] for p in pair)

    def _build_table(self) -> None:
        rows = []
        for e, name, shape, kind, off in self._plan:
            which = 0 if kind <= 1 else 1  # lora_down for kinds 0/1, lora_up for kinds 2/3
            src = [pair[which].data_ptr() for pair in e["params"]] + [0, 0]
            L = e["K"] if kind <= 1 else e["N"]
            rows.append([src[0], src[1], src[2], e[name].data_ptr(), kind | (e["G"] << 32), e["r"] | (e["rp"] << 32), L])
        self._table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._ptrs = self._src_ptrs()

    def refresh(self) -> None:
        """Rebuild every derived tensor from the current parameter values (one launch; graph-capturable)."""
        if self._flat is None:
            return
        if self._src_ptrs() != self._ptrs:  # parameters were re-allocated (.to(), load_weights): new source addresses
            if torch.cuda.is_current_stream_capturing():
                for e in self.entries:
                    e["ver"] = None
                return
            self._build_table()
        with torch.cuda.device(self.device):
            st = _C.lib().vft_lora_pack(_ptr(self._table), self._table.shape[0], _stream())
        _C.check(st, "vft_lora_pack")
        for e in self.entries:
            e["ver"] = tuple(p._version for pair in e["params"] for p in pair)


def _pack_forms(entry: dict | None, params) -> tuple | None:
    """(A, B, At, Bt) of a pack entry if it is current for `params`, else None."""
    ok = (entry is not None and entry["ver"] is not None and len(params) == entry["G"]
          and all(a is ea and b is eb for (a, b), (ea, eb) in zip(params, entry["params"]))
          and entry["ver"] == tuple(p._version for pair in params for p in pair))
    PACK_STATS["pack_hit" if ok else "pack_miss"] += 1
    if not ok:
        return None
    if entry["alias"]:
        return params[0][0], params[0][1], entry["At"], entry["Bt"]
    return entry["A"], entry["B"], entry["At"], entry["Bt"]


class _LoRALinear(torch.autograd.Function):
    """y = x Wᵀ + b + s·(x Aᵀ) Bᵀ with a frozen base (W, b).  The base products are a GEMM (library for
    bf16 weights, the NF4-fused kernel for 4-bit weights); everything rank-sized is the HIP kernels
    above, applied in place on the GEMM outputs."""

    @staticmethod
    def forward(ctx, x, weight, bias, A, B, s, qstate, residual=None):
        x2 = _rows(x)
        A, B = A.contiguous(), B.contiguous()
        res2 = None
        if residual is not None:
            res2 = residual.reshape(-1, residual.shape[-1])
            if res2.stride(-1) != 1 or res2.stride(0) % 8 != 0 or res2.dtype != torch.bfloat16:
                res2 = res2.to(torch.bfloat16).contiguous()
        with torch.cuda.device(x.device):
            y2 = F.linear(x2, weight, bias) if qstate is None else _nf4_gemm(x2, qstate, bias, False)
            t, Bt = _lora_fwd_(x2, y2, A, B, s, res2)
        ctx.save_for_backward(x2, t, Bt, weight, A, B)
        ctx.res_shape = residual.shape if residual is not None else None
        ctx.s = s
        ctx.qstate = qstate
        ctx.xshape = x.shape
        return y2.view(*x.shape[:-1], B.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, t, Bt, weight, A, B = ctx.saved_tensors
        s = ctx.s
        dy2 = _rows(dy)
        dx = dA = dB = None
        with torch.cuda.device(dy.device):
            if ctx.needs_input_grad[0]:
                dx2 = dy2 @ weight if ctx.qstate is None else _nf4_gemm(dy2, ctx.qstate, None, True)
                u = _lora_bwd_dx_(dy2, dx2, Bt, A, s)
                dx = dx2.view(ctx.xshape)
            else:  # input of a frozen trunk's first adapted layer: dx is unused, u = dy·B still needed
                scratch = torch.empty((dy2.shape[0], A.shape[1]), dtype=dy.dtype, device=dy.device)
                u = _lora_bwd_dx_(dy2, scratch, Bt, A, s)
            if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
                dA, dB = _lora_wgrad2(x2, u, dy2, t, A, B, s)
        dres = dy.reshape(ctx.res_shape) if ctx.res_shape is not None and ctx.needs_input_grad[7] else None
        return dx, None, None, dA, dB, None, None, dres


def lora_down(x2: torch.Tensor, A: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """t = scale * (x2 @ A^T) (M, r): the adapter's low-rank activations (vft_lora_down); no autograd."""
    M, K = x2.shape
    r = A.shape[0]
    t = torch.empty((M, r), dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device), _Timed("lora_down", 2.0 * M * K):
        st = _C.lib().vft_lora_down(_ptr(x2), _ptr(A), _ptr(t), M, K, r, x2.stride(0), float(scale), _stream())
    _C.check(st, "vft_lora_down")
    return t


# ---------------------------------------------------------------------------------------------
# LoRA layers on the fused projection GEMM (vft_linear_fwd, csrc/linear8.hip): ONE GEMM launch per direction computes
# the frozen base product, the bias, the adapter update (as r extra K columns) and the block's skip connection, so the
# adapter no longer costs a read-modify-write pass over y / dx.  The rank-r activations come pre-scaled from
# vft_lora_down (t = s x A^T, u = s dy B), so lora_up / lora_down enter the GEMM as stored and the weight gradients
# need no further scaling: dA = u^T x, dB = dy^T t.
# ---------------------------------------------------------------------------------------------
def _ext_rank(r: int) -> int:
    """Rank the GEMM's extension runs at (4, 8 or 16): smaller / odd ranks are zero-padded."""
    return 4 if r <= 4 else 8 if r <= 8 else 16


def _pad_rank_rows(A: torch.Tensor, rp: int) -> torch.Tensor:  # (r, K) -> (rp, K)
    return A if A.shape[0] == rp else torch.cat([A, A.new_zeros(rp - A.shape[0], A.shape[1])], 0)


def _pad_rank_cols(B: torch.Tensor, rp: int) -> torch.Tensor:  # (N, r) -> (N, rp)
    return B if B.shape[1] == rp else torch.cat([B, B.new_zeros(B.shape[0], rp - B.shape[1])], 1)


def _transposed_weight(weight: torch.Tensor) -> torch.Tensor:
    """W^T (in, out) contiguous: the data gradient dy . W is the same NT GEMM on it.  Frozen weights keep the copy
    on the parameter object (288 GB of HBM: memory is not the constraint); trainable ones are transposed per call."""
    def make():
        w = weight.detach()
        R, C = w.shape
        if w.is_cuda and w.dtype == torch.bfloat16 and w.is_contiguous() and R % 8 == 0 and C % 8 == 0:
            out = torch.empty((C, R), dtype=torch.bfloat16, device=w.device)
            with torch.cuda.device(w.device):
                st = _C.lib().vft_transpose_bf16(_ptr(w), _ptr(out), R, C, C, R, _stream())
            _C.check(st, "vft_transpose_bf16")
            return out
        return w.t().contiguous()

    if weight.requires_grad:
        return make()
    hit = getattr(weight, "_vft_wt", None)
    if hit is None or hit[0] != weight._version or hit[1] != weight.data_ptr():
        hit = (weight._version, weight.data_ptr(), make())
        if not torch.cuda.is_current_stream_capturing():  # a captured tensor lives in the graph's private pool
            weight._vft_wt = hit
    return hit[2]


def fused_lora_eligible(weight: torch.Tensor, r: int) -> bool:
    """bf16 weight whose two dimensions are multiples of 64 (forward reduces over in, the data gradient over out),
    rank at most 16."""
    return (weight.dtype == torch.bfloat16 and weight.dim() == 2 and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0
            and r <= 16 and weight.is_cuda)


def _adapters_ok(weights) -> bool:
    """Every lora_down / lora_up handed to a grouped entry point is a CUDA bf16 contiguous tensor.  The kernels take raw
    pointers typed bf16: an fp32 adapter (schema-valid `peft.config.dtype: float32`) would be read as garbage, silently."""
    return all(w.is_cuda and w.dtype == torch.bfloat16 and w.is_contiguous() for w in weights)


def _bf16_rows(t: torch.Tensor | None) -> torch.Tensor | None:
    if t is None:
        return None
    t2 = t.reshape(-1, t.shape[-1])
    if t2.stride(-1) != 1 or t2.stride(0) % 8 != 0 or t2.dtype != torch.bfloat16 or t2.data_ptr() % 16 != 0:
        t2 = t2.to(torch.bfloat16).contiguous()
    return t2


class _LoRALinearFused(torch.autograd.Function):
    """residual + x W^T + b + s (x A^T) B^T, one GEMM launch per direction (reference: LoRALinear.forward,
    src/modules/peft/lora.py:92-104 = three GEMMs, a scale and an add; the residual is the transformer block's
    `x + layer(x)`, src/models/sdxl/denoiser.py:273-284).  `pack`: the layer's AdapterPack entry (operand forms built
    once per step for the whole model); without it they are built here."""

    @staticmethod
    def forward(ctx, x, weight, bias, A, B, s, residual=None, pack=None):
        x2 = _rows(x)
        r = A.shape[0]
        rp = _ext_rank(r)
        forms = _pack_forms(pack, [(A, B)])
        if forms is None:
            A_p, B_p = _pad_rank_rows(A.contiguous(), rp), _pad_rank_cols(B.contiguous(), rp)
            At = Bt = None
        else:
            A_p, B_p, At, Bt = forms
        res2 = _bf16_rows(residual)
        with torch.cuda.device(x.device):
            t = lora_down(x2, A_p, s)  # (M, rp), pre-scaled
            y2 = linear_fused(x2, weight, _bf16_or_none(bias), residual=res2, ext_x=t, ext_w=B_p)
        ctx.save_for_backward(x2, t, weight, A_p, B_p, At, Bt)
        ctx.params = (A, B)
        ctx.s, ctx.r, ctx.xshape = s, r, x.shape
        ctx.res_shape = residual.shape if residual is not None else None
        return y2.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, t, weight, A_p, B_p, At, Bt = ctx.saved_tensors
        s, r = ctx.s, ctx.r
        dy2 = _bf16_rows(dy)
        dx = dA = dB = None
        with torch.cuda.device(dy.device):
            Bt_ = Bt if Bt is not None else B_p.t().contiguous()
            if ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
                u, (dA, dB) = _lora_bwd_rank(x2, dy2, t, Bt_, [ctx.params], r, A_p.shape[0], s)  # u = s dy B rides in the pass over dy
            else:
                u = lora_down(dy2, Bt_, s)  # (M, rp) = s dy B
            if ctx.needs_input_grad[0]:
                dx2 = linear_fused(dy2, _transposed_weight(weight), None, ext_x=u, ext_w=At if At is not None else A_p.t().contiguous())
                dx = dx2.view(ctx.xshape)
        dres = dy.reshape(ctx.res_shape) if ctx.res_shape is not None and ctx.needs_input_grad[6] else None
        return dx, None, None, dA, dB, None, dres, None


class _LoRASiblings(torch.autograd.Function):
    """Several LoRA-adapted projections of the SAME input (AuraFlow's separate q / k / v Linears and the two up projections
    of its SwiGLU MLP, src/models/auraflow/denoiser.py:95-119,144-214): each is one fused GEMM launch forward; in the
    backward the data gradients are CHAINED through the GEMM's residual input — dx = dy_0 W_0 + ... — instead of three
    separate (tokens, C) tensors that autograd then adds with two more elementwise passes.
    Arguments after x: n, s, then per projection (weight, bias, A, B, pack)."""

    @staticmethod
    def forward(ctx, x, n, s, *args):
        x2 = _rows(x)
        outs, saved, metas = [], [x2], []
        with torch.cuda.device(x.device):
            for i in range(n):
                weight, bias, A, B, pack = args[5 * i: 5 * i + 5]
                r = A.shape[0]
                rp = _ext_rank(r)
                forms = _pack_forms(pack, [(A, B)])
                if forms is None:
                    A_p, B_p, At, Bt = _pad_rank_rows(A.contiguous(), rp), _pad_rank_cols(B.contiguous(), rp), None, None
                else:
                    A_p, B_p, At, Bt = forms
                t = lora_down(x2, A_p, s)
                y2 = linear_fused(x2, weight, _bf16_or_none(bias), ext_x=t, ext_w=B_p)
                outs.append(y2.view(*x.shape[:-1], weight.shape[0]))
                saved += [t, weight, A_p, B_p, At, Bt]
                metas.append((r, (A, B)))
        ctx.save_for_backward(*saved)
        ctx.n, ctx.s, ctx.metas, ctx.xshape = n, s, metas, x.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        saved = ctx.saved_tensors
        x2 = saved[0]
        n, s = ctx.n, ctx.s
        grads = [None] * (5 * n)
        dx2 = None
        with torch.cuda.device(x2.device):
            for i in range(n):
                t, weight, A_p, B_p, At, Bt = saved[1 + 6 * i: 7 + 6 * i]
                r, params = ctx.metas[i]
                if dys[i] is None:
                    continue
                dy2 = _bf16_rows(dys[i])
                Bt_ = Bt if Bt is not None else B_p.t().contiguous()
                if ctx.needs_input_grad[3 + 5 * i + 2] or ctx.needs_input_grad[3 + 5 * i + 3]:
                    u, (grads[5 * i + 2], grads[5 * i + 3]) = _lora_bwd_rank(x2, dy2, t, Bt_, [params], r, A_p.shape[0], s)
                else:
                    u = lora_down(dy2, Bt_, s)
                if ctx.needs_input_grad[0]:
                    dx2 = linear_fused(dy2, _transposed_weight(weight), None, residual=dx2, ext_x=u,
                                       ext_w=At if At is not None else A_p.t().contiguous())
        dx = dx2.view(ctx.xshape) if dx2 is not None else None
        return (dx, None, None, *grads)


def lora_siblings(x: torch.Tensor, layers) -> list[torch.Tensor] | None:
    """[layer(x) for layer in layers] for LoRALinear-like modules (`.linear`, `.lora_down`, `.lora_up`, `._scale()`) that share
    the input, with chained data gradients (see `_LoRASiblings`); None when the group is not eligible (the caller then
    applies the layers one by one)."""
    if not (FUSED_LORA_GEMM and x.is_cuda and len(layers) >= 2):
        return None
    s0 = None
    args = []
    for m in layers:
        base = getattr(m, "linear", None)
        if (base is None or not getattr(m, "enabled", False) or type(base) is not torch.nn.Linear or m.training_dropout_active()
                or base.weight.requires_grad or not fused_lora_eligible(base.weight, m.rank)):
            return None
        sc = m._scale()
        if s0 is None:
            s0 = sc
        elif sc != s0:
            return None
        if not _adapters_ok((m.lora_down.weight, m.lora_up.weight)):
            return None  # per-layer calls: kernels.lora_linear raises TypeError for what it cannot run
        args += [base.weight, base.bias, m.lora_down.weight, m.lora_up.weight, m.__dict__.get("_vft_pack")]
    _ready(x)
    _require_gpu(x)
    return list(_LoRASiblings.apply(x, len(layers), float(s0), *args))


class _LoRAGroupFused(torch.autograd.Function):
    """G = 2 or 3 LoRA-wrapped bias-free projections of ONE input with equal shapes (SDXL CrossAttention's to_k | to_v of
    the 77-token context, src/models/sdxl/denoiser.py:135-137) as one GEMM over the concatenated frozen weights with the
    adapters as one rank-G·r extension (the fused q|k|v scheme without the head views).  For the context projections this
    is about launch count and grid size: 1 848 rows fill 24-40 tiles of the 256 CUs per projection, and every adapter costs
    four more tiny launches per step.  Returns G (…, N) views of the (rows, G·N) output."""

    @staticmethod
    def forward(ctx, x, wcat, s, pack, G, *AB):
        x2 = _rows(x)
        params = [(AB[2 * g], AB[2 * g + 1]) for g in range(G)]
        r, N = params[0][0].shape[0], params[0][1].shape[0]
        rp = _ext_rank(G * r)
        forms = _pack_forms(pack, params)
        if forms is None:
            A_cat = _pad_rank_rows(torch.cat([a for a, _ in params], 0), rp)
            B_blk = _pad_rank_cols(torch.block_diag(*[b for _, b in params]), rp)
            At = Bt = None
        else:
            A_cat, B_blk, At, Bt = forms
        with torch.cuda.device(x.device):
            t = lora_down(x2, A_cat, s)
            y = linear_fused(x2, wcat, None, ext_x=t, ext_w=B_blk)  # (M, G N)
        ctx.save_for_backward(x2, wcat, t, A_cat, B_blk, At, Bt)
        ctx.params = params
        ctx.s, ctx.r, ctx.N, ctx.G, ctx.xshape = s, r, N, G, x.shape
        yv = y.view(*x.shape[:-1], G, N)
        return tuple(yv[..., g, :] for g in range(G))

    @staticmethod
    def backward(ctx, *dys):
        x2, wcat, t, A_cat, B_blk, At, Bt = ctx.saved_tensors
        s, r, N, G = ctx.s, ctx.r, ctx.N, ctx.G
        M = x2.shape[0]
        dy = torch.cat([(d if d is not None else x2.new_zeros(M, N)).reshape(M, N) for d in dys], dim=1).to(torch.bfloat16)
        grads = [None] * (2 * G)
        dx = None
        with torch.cuda.device(dy.device):
            Bt_ = Bt if Bt is not None else B_blk.t().contiguous()
            if any(ctx.needs_input_grad[5:]):
                u, grads = _lora_bwd_rank(x2, dy, t, Bt_, ctx.params, r, A_cat.shape[0], s)
            else:
                u = lora_down(dy, Bt_, s)
            if ctx.needs_input_grad[0]:
                dx = linear_fused(dy, _transposed_weight(wcat), None, ext_x=u,
                                  ext_w=At if At is not None else A_cat.t().contiguous()).view(ctx.xshape)
        return (dx, None, None, None, None, *grads)


def lora_group(x: torch.Tensor, wcat: torch.Tensor, adapters, scale: float, pack: dict | None = None):
    """[x W_g^T + s (x A_g^T) B_g^T for g] for G = 2 or 3 bias-free projections sharing x; wcat = cat of the frozen base
    weights (G N, C); adapters = [(lora_down, lora_up)] * G.  None when the fused GEMM does not take the shapes."""
    _ready(x, wcat)
    _require_gpu(x, wcat)
    G = len(adapters)
    r, N = adapters[0][0].shape[0], adapters[0][1].shape[0]
    flat = [t_ for pair in adapters for t_ in pair]
    if not (FUSED_LORA_GEMM and G in (2, 3) and fused_lora_eligible(wcat, G * r) and N % 64 == 0 and _adapters_ok(flat)):
        return None
    return _LoRAGroupFused.apply(x, wcat, float(scale), pack, G, *flat)


class _LoRAQKVFused(torch.autograd.Function):
    """q, k, v of SDXL's SelfAttention (three LoRA-wrapped bias-free projections of one input,
    src/models/sdxl/denoiser.py:32-94) as ONE GEMM per direction over the concatenated frozen weights, the three
    adapters as one rank-3r extension: A_cat = [Aq; Ak; Av] (3r, C), B_blk = blockdiag(Bq, Bk, Bv) (3N, 3r).
    Forward returns (B,H,S,D) views of the (tokens, 3N) output; backward takes the attention kernel's single
    dq|dk|dv buffer.  `pack`: the attention module's AdapterPack entry (A_cat, B_blk and their transposes)."""

    @staticmethod
    def forward(ctx, x, wqkv, Aq, Bq, Ak, Bk, Av, Bv, s, heads, pack=None):
        Bsz, S, C = x.shape
        x2 = _rows(x)
        r, N = Aq.shape[0], Bq.shape[0]
        rp = _ext_rank(3 * r)
        params = [(Aq, Bq), (Ak, Bk), (Av, Bv)]
        forms = _pack_forms(pack, params)
        if forms is None:
            A_cat = _pad_rank_rows(torch.cat([Aq, Ak, Av], 0), rp)
            B_blk = _pad_rank_cols(torch.block_diag(Bq, Bk, Bv), rp)
            At = Bt = None
        else:
            A_cat, B_blk, At, Bt = forms
        with torch.cuda.device(x.device):
            t = lora_down(x2, A_cat, s)
            y = linear_fused(x2, wqkv, None, ext_x=t, ext_w=B_blk)  # (M, 3N)
        ctx.save_for_backward(x2, wqkv, t, A_cat, B_blk, At, Bt)
        ctx.params = params
        ctx.s, ctx.r, ctx.N, ctx.xshape = s, r, N, x.shape
        y5 = y.view(Bsz, S, 3, heads, N // heads)
        return tuple(y5[:, :, i].permute(0, 2, 1, 3) for i in range(3))

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x2, wqkv, t, A_cat, B_blk, At, Bt = ctx.saved_tensors
        s, r, N = ctx.s, ctx.r, ctx.N
        M = x2.shape[0]
        esz = dq.element_size()
        fused = (dq.dtype == torch.bfloat16 and dq.stride() == dk.stride() == dv.stride() and dq.stride(3) == 1
                 and dq.stride(1) * dq.shape[1] == N and dq.stride(2) == 3 * N and dq.stride(0) == dq.shape[2] * 3 * N
                 and dk.data_ptr() == dq.data_ptr() + N * esz and dv.data_ptr() == dq.data_ptr() + 2 * N * esz)
        if fused:
            dy = torch.as_strided(dq, (M, 3 * N), (3 * N, 1))
        else:
            dy = torch.cat([g.permute(0, 2, 1, 3).reshape(M, N) for g in (dq, dk, dv)], dim=1).to(torch.bfloat16)
        grads = [None] * 6
        with torch.cuda.device(dy.device):
            Bt_ = Bt if Bt is not None else B_blk.t().contiguous()
            if any(ctx.needs_input_grad[2:8]):
                u, grads = _lora_bwd_rank(x2, dy, t, Bt_, ctx.params, r, A_cat.shape[0], s)  # u (M, rp) = s dy blockdiag(B)
            else:
                u = lora_down(dy, Bt_, s)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = linear_fused(dy, _transposed_weight(wqkv), None, ext_x=u,
                                  ext_w=At if At is not None else A_cat.t().contiguous()).view(ctx.xshape)
        return (dx, None, *grads, None, None, None)


class _LoRAQKV(torch.autograd.Function):
    """q, k, v = three LoRA-adapted bias-free projections of the same input (SDXL SelfAttention,
    src/models/sdxl/denoiser.py:32-94) with ONE base GEMM per direction: forward x·[Wq;Wk;Wv]ᵀ into a (tokens, 3C)
    buffer whose column slices are q | k | v (returned as (B,H,S,D) views: the attention kernel reads them in place),
    backward [dq|dk|dv]·[Wq;Wk;Wv] in one GEMM (the attention backward already wrote the three gradients into one
    buffer).  The rank-sized kernels run per projection on the column slices."""

    @staticmethod
    def forward(ctx, x, wqkv, Aq, Bq, Ak, Bk, Av, Bv, s, heads):
        Bsz, S, C = x.shape
        x2 = _rows(x)
        As = [t.contiguous() for t in (Aq, Ak, Av)]
        Bs = [t.contiguous() for t in (Bq, Bk, Bv)]
        N = Bs[0].shape[0]
        with torch.cuda.device(x.device):
            y = F.linear(x2, wqkv)  # (M, 3N)
            ts, Bts = [], []
            for i in range(3):
                t, Bt = _lora_fwd_(x2, y[:, i * N:(i + 1) * N], As[i], Bs[i], s)
                ts.append(t)
                Bts.append(Bt)
        ctx.save_for_backward(x2, wqkv, *ts, *Bts, *As, *Bs)
        ctx.s, ctx.xshape, ctx.N = s, x.shape, N
        y5 = y.view(Bsz, S, 3, heads, N // heads)
        return tuple(y5[:, :, i].permute(0, 2, 1, 3) for i in range(3))

    @staticmethod
    def backward(ctx, dq, dk, dv):
        x2, wqkv = ctx.saved_tensors[:2]
        ts, Bts, As, Bs = (ctx.saved_tensors[2 + 3 * i: 5 + 3 * i] for i in range(4))
        s, N = ctx.s, ctx.N
        M = x2.shape[0]
        esz = dq.element_size()
        fused = (dq.dtype == torch.bfloat16 and dq.stride() == dk.stride() == dv.stride() and dq.stride(3) == 1
                 and dq.stride(1) * dq.shape[1] == N and dq.stride(2) == 3 * N and dq.stride(0) == dq.shape[2] * 3 * N
                 and dk.data_ptr() == dq.data_ptr() + N * esz and dv.data_ptr() == dq.data_ptr() + 2 * N * esz)
        if fused:
            dy = torch.as_strided(dq, (M, 3 * N), (3 * N, 1))
        else:
            dy = torch.cat([g.permute(0, 2, 1, 3).reshape(M, N) for g in (dq, dk, dv)], dim=1).to(torch.bfloat16)
        grads = [None] * 6
        r = As[0].shape[0]
        with torch.cuda.device(dy.device):
            dx2 = dy @ wqkv  # (M, 3N) x (3N, C)
            if 3 * r <= 16:
                # the three adapters' data gradients in ONE pass over dy and dx: u = dy · blockdiag(Bq,Bk,Bv)
                # (M, 3r), dx += s · u · [Aq;Ak;Av] — the rank-3r instance of the same kernel
                u_all = _lora_bwd_dx_(dy, dx2, torch.block_diag(*Bts), torch.cat(list(As), 0), s)
                us = None
            else:
                us = [_lora_bwd_dx_(dy[:, i * N:(i + 1) * N], dx2, Bts[i], As[i], s) for i in range(3)]
            if 3 * r <= 16 and all(ctx.needs_input_grad[2:8]):
                # both weight gradients of the three adapters in one launch pair: dA_cat (3r, C) = u_allᵀ · x (x read
                # once instead of three times); dB_all (3N, 3r) = dyᵀ · t_all, of which the diagonal blocks are dB_i
                A_cat, B_all = torch.cat(list(As), 0), torch.block_diag(*Bs)
                dA_cat, dB_all = _lora_wgrad2(x2, u_all, dy, torch.cat(list(ts), 1), A_cat, B_all, s)
                for i in range(3):
                    grads[2 * i] = dA_cat[i * r:(i + 1) * r]
                    grads[2 * i + 1] = dB_all[i * N:(i + 1) * N, i * r:(i + 1) * r]
            else:
                if us is None:
                    us = [u_all[:, i * r:(i + 1) * r].contiguous() for i in range(3)]
                for i in range(3):
                    if ctx.needs_input_grad[2 + 2 * i] or ctx.needs_input_grad[3 + 2 * i]:
                        grads[2 * i], grads[2 * i + 1] = _lora_wgrad2(x2, us[i], dy[:, i * N:(i + 1) * N], ts[i], As[i], Bs[i], s)
        dx = dx2.view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return (dx, None, *grads, None, None)


def lora_qkv(x: torch.Tensor, wqkv: torch.Tensor, adapters, scale: float, heads: int, pack: dict | None = None):
    """(q, k, v) as (B,H,S,D) views for three LoRA-wrapped bias-free projections of `x` (B,S,C);
    wqkv = cat of the frozen base weights (3N, C); adapters = [(lora_down, lora_up)] * 3."""
    _ready(x, wqkv)
    _require_gpu(x, wqkv)
    (Aq, Bq), (Ak, Bk), (Av, Bv) = adapters
    _require_gpu(Aq, Bq, Ak, Bk, Av, Bv)  # fp32 adapters would be read as bf16 by the kernels: fail loudly
    if FUSED_LORA_GEMM and fused_lora_eligible(wqkv, 3 * Aq.shape[0]) and Bq.shape[0] % 64 == 0:
        return _LoRAQKVFused.apply(x, wqkv, Aq, Bq, Ak, Bk, Av, Bv, float(scale), int(heads), pack)
    return _LoRAQKV.apply(x, wqkv, Aq, Bq, Ak, Bk, Av, Bv, float(scale), int(heads))


class _QLoRALinear(torch.autograd.Function):
    """NF4 base + LoRA, each direction ONE GEMM launch: the rank-r activations (t = x Aᵀ forward, u = dy B backward)
    come from `vft_lora_down`, the update enters the NF4 GEMM's accumulators as r extra K columns (`vft_linear_nf4`
    with ext_x / ext_w), so neither the adapter nor the 4-bit base ever costs a separate pass over y / dx
    (reference: LoRALinear over BnbLinear4bit = bnb dequant kernel + GEMM + 3 adapter GEMMs + scale + add)."""

    @staticmethod
    def forward(ctx, x, bias, A, B, s, qstate, residual=None, pack=None):
        x2 = _rows(x)
        forms = _pack_forms(pack, [(A, B)])
        if forms is None:
            A_c, B_c, At, Bt = A.contiguous(), B.contiguous(), None, None
        else:
            A_c, B_c, At, Bt = forms
        res2 = _bf16_rows(residual)
        with torch.cuda.device(x.device):
            t = lora_down(x2, A_c, s)  # pre-scaled, so lora_up / lora_down enter the GEMMs as stored
            y2 = _nf4_gemm(x2, qstate, bias, False, ext_x=t, ext_w=B_c, residual=res2)
        ctx.save_for_backward(x2, t, A_c, B_c, At, Bt)
        ctx.params = (A, B)
        ctx.s, ctx.qstate, ctx.xshape = s, qstate, x.shape
        ctx.res_shape = residual.shape if residual is not None else None
        return y2.view(*x.shape[:-1], B.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, t, A, B, At, Bt = ctx.saved_tensors
        s = ctx.s
        dy2 = _bf16_rows(dy)
        dx = dA = dB = None
        with torch.cuda.device(dy.device):
            Bt_ = Bt if Bt is not None else B.t().contiguous()
            if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                u, (dA, dB) = _lora_bwd_rank(x2, dy2, t, Bt_, [ctx.params], A.shape[0], A.shape[0], s)
            else:
                u = lora_down(dy2, Bt_, s)  # (M, r) = s dy . B
            if ctx.needs_input_grad[0]:
                dx2 = _nf4_gemm(dy2, ctx.qstate, None, True, ext_x=u, ext_w=At if At is not None else A.t().contiguous())
                dx = dx2.view(ctx.xshape)
        dres = dy.reshape(ctx.res_shape) if ctx.res_shape is not None and ctx.needs_input_grad[6] else None
        return dx, None, dA, dB, None, None, dres, None


def lora_geglu_down(x: torch.Tensor, up_base, up_A, up_B, s1: float, down_base, down_A, down_B, s2: float,
                    residual: torch.Tensor | None = None, up_pack: dict | None = None, down_pack: dict | None = None) -> torch.Tensor:
    """residual + down(geglu(up(x))) for two LoRA-wrapped bf16 Linears (the SDXL FeedForward, src/models/sdxl/denoiser.py:175-207):
    both adapters ride in their projection's GEMM (forward and data gradient), a plain GEGLU in between.  Shapes the fused
    GEMM does not take (a dimension that is no multiple of 64) go through `lora_linear`'s general composition.  (Rounds 1-4
    also carried a third form that applied the adapters inside special GEGLU kernels; it lost to the fused GEMM in round 2
    and was removed in round 5 together with its kernels.)"""
    _ready(x, up_A, up_B, down_A, down_B)
    _require_gpu(x, up_A, up_B, down_A, down_B)
    proj = lora_linear(x, up_base, up_A, up_B, float(s1), None, up_pack)
    return lora_linear(geglu(proj), down_base, down_A, down_B, float(s2), residual, down_pack)


def lora_geglu_eligible(x: torch.Tensor, base, lora_down_w: torch.Tensor) -> bool:
    return (type(base) is torch.nn.Linear and base.weight.dtype == torch.bfloat16 and x.is_cuda
            and lora_down_w.shape[0] <= 8 and base.out_features % 16 == 0)


def lora_linear_geglu(x: torch.Tensor, base, lora_down_w: torch.Tensor, lora_up_w: torch.Tensor, scale: float) -> torch.Tensor:
    """geglu(base(x) + scale * (x @ lora_down^T) @ lora_up^T)."""
    return geglu(lora_linear(x, base, lora_down_w, lora_up_w, float(scale)))


class _LoRABranch(torch.autograd.Function):
    """y += s·(x Aᵀ) Bᵀ in place on the output of an arbitrary base layer (e.g. the NF4 linear)."""

    @staticmethod
    def forward(ctx, x, y, A, B, s):
        x2 = _rows(x)
        y2 = y.view(-1, y.shape[-1])
        A, B = A.contiguous(), B.contiguous()
        with torch.cuda.device(x.device):
            t, Bt = _lora_fwd_(x2, y2, A, B, s)
        ctx.mark_dirty(y)
        ctx.save_for_backward(x2, t, Bt, A, B)
        ctx.s = s
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, t, Bt, A, B = ctx.saved_tensors
        s = ctx.s
        dy2 = _rows(dy)
        dA = dB = None
        with torch.cuda.device(dy.device):
            dx2 = torch.zeros((x2.shape[0], x2.shape[1]), dtype=dy.dtype, device=dy.device)
            u = _lora_bwd_dx_(dy2, dx2, Bt, A, s)
            if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                dA, dB = _lora_wgrad2(x2, u, dy2, t, A, B, s)
        dx = dx2.view(ctx.xshape) if ctx.needs_input_grad[0] else None
        return dx, dy, dA, dB, None


def lora_linear(x: torch.Tensor, base, lora_down: torch.Tensor, lora_up: torch.Tensor, scale: float,
                residual: torch.Tensor | None = None, pack: dict | None = None) -> torch.Tensor:
    """base(x) + scale * (x @ lora_down^T) @ lora_up^T (+ residual) with the rank-sized work in fused HIP kernels.

    `base` is the wrapped layer: a plain ``nn.Linear`` takes the fully fused path (dx is produced in
    one buffer); anything else (NF4 linear, ...) gets the in-place low-rank branch on its output."""
    _ready(x, lora_down, lora_up)
    _require_gpu(x, lora_down, lora_up)
    if lora_down.shape[0] > 16:
        raise ValueError(f"LoRA rank {lora_down.shape[0]} > 16 is not supported by the MI355X kernels")
    if type(base) is torch.nn.Linear and base.weight.dtype == torch.bfloat16:
        if FUSED_LORA_GEMM and fused_lora_eligible(base.weight, lora_down.shape[0]):
            return _LoRALinearFused.apply(x, base.weight, base.bias, lora_down, lora_up, float(scale), residual, pack)
        return _LoRALinear.apply(x, base.weight, base.bias, lora_down, lora_up, float(scale), None, residual)
    if hasattr(base, "quant_state"):  # NF4Linear: dequant-fused GEMM for the base products
        dense = base.dense_weight() if hasattr(base, "dense_weight") else None
        if dense is not None and FUSED_LORA_GEMM and fused_lora_eligible(dense, lora_down.shape[0]):
            # compute cache on (modules/quant/nf4.py): the bf16 projection GEMM on the dequantised copy
            return _LoRALinearFused.apply(x, dense, base.bias, lora_down, lora_up, float(scale), residual, pack)
        q = base.quant_state()
        r = lora_down.shape[0]
        if r in (4, 8, 16) and q.in_features % 64 == 0 and q.out_features % 64 == 0:
            return _QLoRALinear.apply(x, base.bias, lora_down, lora_up, float(scale), q, residual, pack)
        return _LoRALinear.apply(x, None, base.bias, lora_down, lora_up, float(scale), q, residual)
    y = base(x)
    if not y.is_contiguous():
        y = y.contiguous()
    y = _LoRABranch.apply(x, y, lora_down, lora_up, float(scale))
    return y if residual is None else residual + y


# =============================================================================================
# GEMM on our MFMA kernel (bf16 weight) and NF4-fused linear
# =============================================================================================
# Problem sizes seen by the fused GEMM / implicit-GEMM convolution (tools/tune_lin8.py switches this on for one step of
# each workload to know what to time): None = off, else a set of ("lin", M, N, K) / ("conv", B, H, W, Cin, Cout).
SHAPE_LOG: set | None = None


def linear_fused(x2: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
                 residual: torch.Tensor | None = None, ext_x: torch.Tensor | None = None,
                 ext_w: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """y = x2 @ weight^T (+bias) (+ ext_x @ ext_w^T) (+residual) in one launch (vft_linear_fwd); no autograd."""
    _ready(x2, weight)
    _require_gpu(x2, weight, bias)
    M, K = x2.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x2.device)
    r = ext_x.shape[1] if ext_x is not None else 0
    if SHAPE_LOG is not None:
        SHAPE_LOG.add(("lin", M, N, K))
    with torch.cuda.device(x2.device), _Timed("linear_fwd", 2.0 * M * N * (K + r)):
        st = _C.lib().vft_linear_fwd(_ptr(x2), _ptr(weight), _ptr(out), _ptr(bias), _ptr(residual), _ptr(ext_x), _ptr(ext_w),
                                     M, N, K, r, x2.stride(0), weight.stride(0), out.stride(0),
                                     residual.stride(0) if residual is not None else 0, _stream())
    _C.check(st, "vft_linear_fwd")
    return out


# =============================================================================================
# plain (trainable or frozen) projection on the fused GEMM: full fine-tuning and the non-adapted Linears
# =============================================================================================
PLAIN_LINEAR_GEMM = os.environ.get("VFT_PLAIN_LINEAR", "1") != "0"  # A/B switch: 0 = plain nn.Linear stays on the library


# Rows below which a frozen projection stays on the library GEMM.  Round 3: 1 — the conditioning MLPs (a handful of rows:
# the batch) run on the own kernel too: slower by microseconds, but they were the last library GEMMs of a LoRA step, and the
# library picks other solutions (stream-K, atomics) under graph capture than eagerly: with them gone a captured step computes
# what the eager step computes.  VFT_LINEAR_MIN_ROWS=1024 restores the round-2 behaviour.
LINEAR_MIN_ROWS = int(os.environ.get("VFT_LINEAR_MIN_ROWS", "1"))


def linear_eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """nn.Linear calls the projection GEMM takes: FROZEN bf16 weights on the GPU (in_features a multiple of 64, out_features
    of 32: the data gradient reduces over out_features and vft_linear_fwd needs K % 32 == 0) and enough rows for the 128-row tiles to matter (tiny conditioning MLPs stay on the library).  Trainable weights
    (full fine-tune) stay on the tuned library GEMMs: measured A/B on one MI355X, SDXL full fine-tune B=16 runs 27.4
    images/s there and 26.1 here — the per-step transposes of 2.6 B weights for the data gradient and the table-tuned
    library solutions outweigh the fused bias / skip epilogue."""
    return (PLAIN_LINEAR_GEMM and not weight.requires_grad and x.is_cuda and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 2
            and weight.shape[1] % 64 == 0 and weight.shape[0] % 32 == 0 and x.numel() // x.shape[-1] >= LINEAR_MIN_ROWS)


class _Linear(torch.autograd.Function):
    """residual + x W^T + b on vft_linear_fwd (bias and the skip add in the epilogue) with the data gradient on the same
    kernel (nn.Linear of every projection the reference builds, e.g. src/models/sdxl/denoiser.py:62-64,183-204).  The
    weight gradient is a plain library GEMM that ACCUMULATES straight into the parameter's gradient tensor when there is
    one (the trainer's flat buffer): no separate accumulation pass per parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        x2 = _rows(x)
        res2 = _bf16_rows(residual)
        with torch.cuda.device(x.device):
            y2 = linear_fused(x2, weight, _bf16_or_none(bias), residual=res2)
        ctx.save_for_backward(x2, weight)
        ctx.params = (weight, bias)
        ctx.xshape = x.shape
        ctx.res_shape = residual.shape if residual is not None else None
        return y2.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        wparam, bparam = ctx.params
        dy2 = _bf16_rows(dy)
        dx = dw = db = None
        with torch.cuda.device(dy.device):
            if ctx.needs_input_grad[0]:
                dx = linear_fused(dy2, _transposed_weight(weight), None).view(ctx.xshape)
            if ctx.needs_input_grad[1]:
                g = _grad_target(wparam)
                if g is not None:
                    g.addmm_(dy2.t(), x2)  # beta = 1: accumulate in place
                else:
                    dw = dy2.t() @ x2
            if bparam is not None and ctx.needs_input_grad[2]:
                gb = _grad_target(bparam)
                if gb is not None:
                    gb.add_(dy2.sum(0, dtype=torch.float32))
                else:
                    db = dy2.sum(0, dtype=torch.float32).to(bparam.dtype)
        dres = dy.reshape(ctx.res_shape) if ctx.res_shape is not None and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, residual: torch.Tensor | None = None) -> torch.Tensor:
    """residual + F.linear(x, weight, bias) on the fused projection GEMM (see `linear_eligible`)."""
    _ready(x, weight)
    _require_gpu(x, weight, bias)
    return _Linear.apply(x, weight, bias, residual)


# Trainable projections (full fine-tune, BASELINE config 2).  Round 3: ALL THREE GEMMs of such a layer run on our kernels —
#   forward        y = x W^T + b (+ residual)   vft_linear_fwd (lin8), bias and skip add in the epilogue
#   data gradient  dx = dy W                    vft_linear_fwd on a transposed copy of W made in the backward pass (the weight
#                                               changes every step; 2.6 B parameters = ~3 ms of copies per step)
#   weight grad    dW = dy^T x                  vft_linear_wgrad (wg8), accumulated straight into the parameter's gradient
# Measured on one MI355X at the B = 24 shapes (tools/exp/trained_linear_bench.py; library = hipBLASLt with the committed
# solution table): forward 86.2 vs 82.4 ms per step (the K = 640 / 1280 FeedForward up-projections lose: 10-20 K-tiles per
# output tile), data gradient 80.3 vs 87.0, weight gradients 124.0 vs 168.3 with everything on wg8 (the library is 2.5x
# slower on the small attention matrices; wg8 is 3-15 % slower on the three FeedForward shapes).  VFT_TRAINED_LIBRARY=1
# restores the round-2 split (library forward / data gradient / large weight gradients) for A/B runs.
TRAINED_LINEAR_WGRAD = os.environ.get("VFT_TRAINED_WGRAD", "1") != "0"  # A/B switch
TRAINED_ON_LIBRARY = os.environ.get("VFT_TRAINED_LIBRARY", "0") != "0"   # A/B switch: round-2 behaviour


def trained_linear_eligible(x: torch.Tensor, weight: torch.Tensor) -> bool:
    if not (TRAINED_LINEAR_WGRAD and weight.requires_grad and x.is_cuda and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and weight.dim() == 2 and torch.is_grad_enabled()):
        return False
    R, Cc = weight.shape
    M = x.numel() // x.shape[-1]
    return M >= 4096 and R % 8 == 0 and Cc % 8 == 0


def _wgrad_on_kernel(M: int, R: int, Cc: int) -> bool:
    """Round 2: only small outputs (at most 30 tiles of 256 x 256: the attention projections) took the split-reduction
    kernel; now every shape its 32-bit offsets cover does, unless the library split is asked for."""
    tiles = ((R + 255) // 256) * ((Cc + 255) // 256)
    return (tiles <= 30 or not TRAINED_ON_LIBRARY) and (M + 130) * max(R, Cc) * 2 < 2**31


def _gemm_on_kernel(weight: torch.Tensor) -> bool:
    """Forward / data gradient of a trainable weight on vft_linear_fwd: both reduction lengths multiples of 64."""
    return not TRAINED_ON_LIBRARY and weight.shape[0] % 64 == 0 and weight.shape[1] % 64 == 0


class _LinearTrained(torch.autograd.Function):
    """F.linear of a trainable weight (src/models/sdxl/denoiser.py:62-64,135-137,183-204), optionally + residual."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual):
        ctx.save_for_backward(x, weight)
        ctx.params = (weight, bias)
        ctx.has_res = residual is not None
        x2 = _rows(x)
        if _gemm_on_kernel(weight) and x2.stride(1) == 1:
            with torch.cuda.device(x.device):
                y2 = linear_fused(x2, weight, _bf16_or_none(bias), residual=_bf16_rows(residual))
            return y2.view(*x.shape[:-1], weight.shape[0])
        with _Timed("lib_gemm_fwd", 2.0 * x2.shape[0] * weight.shape[0] * weight.shape[1]):
            y = F.linear(x, weight, bias)
        return y if residual is None else residual + y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        wparam, bparam = ctx.params
        dy2, x2 = _bf16_rows(dy), _bf16_rows(x)
        dx = dw = db = None
        flops = 2.0 * dy2.shape[0] * weight.shape[0] * weight.shape[1]
        if ctx.needs_input_grad[0]:
            if _gemm_on_kernel(weight):
                with torch.cuda.device(dy.device):
                    dx = linear_fused(dy2, _transposed_weight(weight), None).view(x.shape)
            else:
                with _Timed("lib_gemm_dgrad", flops):
                    dx = (dy2 @ weight).view(x.shape)
        if ctx.needs_input_grad[1]:
            g = _grad_target(wparam)
            if _wgrad_on_kernel(dy2.shape[0], weight.shape[0], weight.shape[1]) and linear_wgrad_eligible(dy2, x2):
                with torch.cuda.device(dy.device):
                    if g is not None:
                        linear_wgrad(dy2, x2, out=g, accumulate=True)
                    else:
                        dw = linear_wgrad(dy2, x2)
            else:
                with _Timed("lib_gemm_wgrad", flops):
                    dw = dy2.t() @ x2
        if bparam is not None and ctx.needs_input_grad[2]:
            db = colsum(dy2, bparam)
        dres = dy if ctx.has_res and ctx.needs_input_grad[3] else None
        return dx, dw, db, dres


def linear_trained(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None,
                   residual: torch.Tensor | None = None) -> torch.Tensor:
    _ready(x, weight)
    _require_gpu(x, weight, bias)
    return _LinearTrained.apply(x, weight, bias, residual)


# =============================================================================================
# 3x3 convolution (implicit GEMM on the fused-linear kernel), channels-last
# =============================================================================================
def conv3x3_eligible(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups) -> bool:
    """True when `kernels.conv3x3` can run this nn.Conv2d call (everything else stays on MIOpen)."""
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] % 32 == 0 and weight.shape[0] % 32 == 0
            and x.shape[0] * x.shape[2] * x.shape[3] < 2**31)


def _conv3x3_call(x_nhwc: torch.Tensor, w_ohwi: torch.Tensor, bias, B, H, W, Cin, Cout, tag: str) -> torch.Tensor:
    y = torch.empty((B, Cout, H, W), dtype=torch.bfloat16, device=x_nhwc.device, memory_format=torch.channels_last)
    if SHAPE_LOG is not None:
        SHAPE_LOG.add(("conv", B, H, W, Cin, Cout))
    with torch.cuda.device(x_nhwc.device), _Timed(tag, 18.0 * B * H * W * Cin * Cout):
        st = _C.lib().vft_conv3x3_fwd(_ptr(x_nhwc), _ptr(w_ohwi), _ptr(y), _ptr(bias), B, H, W, Cin, Cout, _stream())
    _C.check(st, "vft_conv3x3_fwd")
    return y


def _dgrad_weight(weight: torch.Tensor) -> torch.Tensor:
    """w'[Cin,3,3,Cout] (storage of a channels-last (Cin,Cout,3,3) tensor) = w[Cout,2-kh,2-kw,Cin]."""
    def make():
        return weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)

    if weight.requires_grad:  # full fine-tune: the weight changes every step
        return make()
    # cached on the parameter object itself (lives and dies with it): (version, data_ptr, flipped/transposed copy)
    hit = getattr(weight, "_vft_dgrad_weight", None)
    if hit is None or hit[0] != weight._version or hit[1] != weight.data_ptr():
        hit = (weight._version, weight.data_ptr(), make())
        if not torch.cuda.is_current_stream_capturing():  # a captured tensor lives in the graph's private pool
            weight._vft_dgrad_weight = hit
    return hit[2]


class _Conv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        x = x.contiguous(memory_format=torch.channels_last)
        w = weight.contiguous(memory_format=torch.channels_last)
        b_ = _bf16_or_none(bias)
        y = _conv3x3_call(x, w, b_, B, H, W, Cin, Cout, "conv3x3_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv3x3_call(dy, _dgrad_weight(weight), None, B, H, W, Cout, Cin, "conv3x3_dgrad")
        need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        if need_w and conv3x3_wgrad_eligible(B, H, W, Cin, Cout):
            # weight gradient (full fine-tune only) on vft_conv3x3_wgrad, accumulated straight into the parameter's
            # channels-last gradient (the trainer's flat buffer) when there is one
            g2 = _conv_grad_target(ctx.wparam)
            with torch.cuda.device(dy.device):
                if g2 is not None:
                    conv3x3_wgrad(dy, x, out=g2, accumulate=True)
                else:
                    dw = conv3x3_wgrad(dy, x).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).to(weight.dtype)
            if need_b:
                db = colsum(dy.permute(0, 2, 3, 1).reshape(-1, Cout), ctx.bparam)
        elif need_w or need_b:  # geometries the kernel does not take: MIOpen
            _, dw, db = torch.ops.aten.convolution_backward(
                dy, x, weight, [Cout] if ctx.has_bias else None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                [False, bool(need_w), bool(need_b)])
        return dx, dw, db


# ---- the convolutions that used to stay on MIOpen (round 5): narrow channels and stride 2 -----------------------------------
# MIOpen ran the 4-channel stem / head and the two stride-2 Downsample convolutions of the SDXL UNet with kernels that
# accumulate through atomics: the last launches of the SDXL step whose results changed from run to run.
NATIVE_SMALL_CONVS = os.environ.get("VFT_NATIVE_SMALL_CONVS", "1") != "0"  # A/B switch (0: MIOpen, as in rounds 1-4)


def conv3x3_narrow_eligible(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups) -> bool:
    """A 3x3 / stride 1 / padding 1 convolution whose channel counts are below the implicit-GEMM kernel's granule (the SDXL
    stem 4 -> 320 and head 320 -> 4, src/models/sdxl/denoiser.py:1114): runs the same kernel on zero-padded channels."""
    return (NATIVE_SMALL_CONVS and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and (weight.shape[1] % 32 != 0 or weight.shape[0] % 32 != 0)
            and x.shape[0] * x.shape[2] * x.shape[3] < 2**31)


def conv3x3_narrow(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """conv3x3 on channels padded with zeros to multiples of 64 (input) / 64 (output), the surplus output channels dropped.
    Pure composition: the pads and the slice are differentiable copies, so the data gradient, the weight gradient
    (vft_conv3x3_wgrad on the padded operands, then sliced) and the bias gradient (vft_colsum) come from `_Conv3x3`.
    Cost at the SDXL sizes: 16x the stem's / head's real flops = ~0.15 ms per launch, against 0.3-0.5 ms on MIOpen."""
    Cout, Cin = weight.shape[0], weight.shape[1]
    ci, co = (-Cin) % 64, (-Cout) % 64
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, ci)) if ci else x
    wp = torch.nn.functional.pad(weight, (0, 0, 0, 0, 0, ci, 0, co)) if (ci or co) else weight
    bp = torch.nn.functional.pad(bias, (0, co)) if (bias is not None and co) else bias
    y = conv3x3(xp.contiguous(memory_format=torch.channels_last), wp, bp)
    return y[:, :Cout] if co else y


def conv3x3_s2_eligible(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups) -> bool:
    """The SDXL Downsample: 3x3 / stride 2 / padding 1, even image, channels % 64 == 0 (src/models/sdxl/denoiser.py:371-408)."""
    return (NATIVE_SMALL_CONVS and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (2, 2) and tuple(padding) == (1, 1)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0 and x.shape[0] * x.shape[2] * x.shape[3] < 2**31)


class _Conv3x3S2(torch.autograd.Function):
    """Forward: vft_conv3x3_s2_fwd (the implicit GEMM gathering every second pixel).  Backward: the output gradient is spread
    onto the even pixels of a zero H x W image; on that image the data gradient is the stride-1 call on the flipped,
    transposed weight and the weight gradient the stride-1 vft_conv3x3_wgrad (the zeros contribute nothing) — 4x the minimal
    flops on two launches per step (+0.8 ms), no atomics anywhere."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        x = x.contiguous(memory_format=torch.channels_last)
        w = weight.contiguous(memory_format=torch.channels_last)
        y = torch.empty((B, Cout, H // 2, W // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        with torch.cuda.device(x.device), _Timed("conv3x3_fwd", 18.0 * B * (H // 2) * (W // 2) * Cin * Cout):
            st = _C.lib().vft_conv3x3_s2_fwd(_ptr(x), _ptr(w), _ptr(y), _ptr(_bf16_or_none(bias)), B, H, W, Cin, Cout, _stream())
        _C.check(st, "vft_conv3x3_s2_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.wparam, ctx.bparam = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        dx = dw = db = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = ctx.has_bias and ctx.needs_input_grad[2]
        if need_x or need_w:
            up = torch.empty((B, Cout, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last).zero_()
            up[:, :, ::2, ::2] = dy
            if need_x:
                dx = _conv3x3_call(up, _dgrad_weight(weight), None, B, H, W, Cout, Cin, "conv3x3_dgrad")
            if need_w:
                g2 = _conv_grad_target(ctx.wparam)
                with torch.cuda.device(dy.device):
                    if g2 is not None and conv3x3_wgrad_eligible(B, H, W, Cin, Cout):
                        conv3x3_wgrad(up, x, out=g2, accumulate=True)
                    elif conv3x3_wgrad_eligible(B, H, W, Cin, Cout):
                        dw = conv3x3_wgrad(up, x).view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).to(weight.dtype)
                    else:
                        dw = torch.ops.aten.convolution_backward(dy, x, weight, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1,
                                                                 [False, True, False])[1]
        if need_b:
            db = colsum(dy.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).reshape(-1, Cout), ctx.bparam)
        return dx, dw, db


def conv3x3_s2(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """3x3 / stride 2 / padding 1 convolution of a (B,Cin,H,W) tensor, channels-last in and out."""
    _ready(x, weight)
    _require_gpu(x, weight, bias)
    return _Conv3x3S2.apply(x, weight, bias)


CONV_WGRAD_KERNEL = os.environ.get("VFT_CONV_WGRAD", "1") != "0"  # A/B switch


def conv3x3_wgrad_eligible(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    return (CONV_WGRAD_KERNEL and W % 8 == 0 and Cin % 8 == 0 and Cout % 8 == 0 and B * H * W >= 4096
            and (B * H * W + 130) * max(Cin, Cout) * 2 < 2**31)


def _conv_grad_target(weight: torch.Tensor | None):
    """[Cout, 9*Cin] view ([tap][Cin] rows) of a convolution weight's existing channels-last bf16 gradient, or None."""
    g = getattr(weight, "grad", None) if (weight is not None and weight.is_leaf) else None
    if g is None or g.dtype != torch.bfloat16 or not g.is_cuda or g.dim() != 4 or g.shape != weight.shape:
        return None
    g2 = g.permute(0, 2, 3, 1)
    if not g2.is_contiguous() or g.data_ptr() % 16:
        return None
    return g2.reshape(g.shape[0], -1)


def conv3x3_wgrad(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    """dW[Cout, 9*Cin] (rows = [tap][Cin], the channels-last weight layout) of a 3x3 / stride 1 / padding 1 convolution from
    the channels-last output gradient dy (B,Cout,H,W) and input x (B,Cin,H,W): implicit im2col inside vft_conv3x3_wgrad."""
    _require_gpu(dy, x)
    B, Cin, H, W = x.shape
    Cout = dy.shape[1]
    dy = dy.contiguous(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
    if out is None:
        out = torch.empty((Cout, 9 * Cin), dtype=torch.bfloat16, device=dy.device)
        accumulate = False
    if out.shape != (Cout, 9 * Cin) or not out.is_contiguous() or out.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("conv3x3_wgrad: `out` must be a contiguous (Cout, 9*Cin) bf16 / fp32 matrix")
    lib = _C.lib()
    scratch = torch.empty(int(lib.vft_conv3x3_wgrad_scratch_floats(B * H * W, Cout, Cin)), dtype=torch.float32, device=dy.device)
    with _Timed("conv3x3_wgrad", 2.0 * B * H * W * Cout * 9 * Cin):
        st = lib.vft_conv3x3_wgrad(_ptr(dy), _ptr(x), _ptr(out), _ptr(scratch), B, H, W, Cin, Cout, int(bool(accumulate)),
                                   int(out.dtype == torch.float32), _stream())
    _C.check(st, "vft_conv3x3_wgrad")
    return out


def _nhwc_rows(x: torch.Tensor) -> torch.Tensor:
    """(B,C,H,W) channels-last -> its (B*H*W, C) token matrix (a view)."""
    x = x.contiguous(memory_format=torch.channels_last)
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1])


def conv1x1_eligible(x: torch.Tensor, weight: torch.Tensor, stride, padding, dilation, groups) -> bool:
    return (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and tuple(weight.shape[2:]) == (1, 1) and tuple(stride) == (1, 1) and tuple(padding) == (0, 0)
            and tuple(dilation) == (1, 1) and groups == 1 and weight.shape[1] % 64 == 0 and weight.shape[0] % 64 == 0)


class _Conv1x1(torch.autograd.Function):
    """add + conv1x1(x): on channels-last storage a 1x1 convolution is the projection GEMM over the pixel rows, so the
    ResidualBlock's skip convolution and its `h + skip(x)` add (src/models/sdxl/denoiser.py:553-586) are one
    vft_linear_fwd launch (bias and the add in the epilogue); the data gradient is the same call on the transposed weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, add):
        B, Cin, H, W = x.shape
        Cout = weight.shape[0]
        x2 = _nhwc_rows(x)
        w2 = weight.reshape(Cout, Cin)
        a2 = _nhwc_rows(add) if add is not None else None
        y = torch.empty((B, Cout, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
        linear_fused(x2, w2, _bf16_or_none(bias), residual=a2, out=y.permute(0, 2, 3, 1).reshape(-1, Cout))
        ctx.save_for_backward(x2, weight)
        ctx.shape, ctx.has_bias, ctx.has_add = (B, Cin, H, W), bias is not None, add is not None
        ctx.wparam, ctx.bparam = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        B, Cin, H, W = ctx.shape
        Cout = weight.shape[0]
        dy = dy.contiguous(memory_format=torch.channels_last)
        dy2 = dy.permute(0, 2, 3, 1).reshape(-1, Cout)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B, Cin, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
            wt = weight.detach().reshape(Cout, Cin).t().contiguous() if weight.requires_grad else _conv1x1_wt(weight)
            linear_fused(dy2, wt, None, out=dx.permute(0, 2, 3, 1).reshape(-1, Cin))
        if ctx.needs_input_grad[1]:  # full fine-tune: the projection weight-gradient kernel over the pixel rows
            if _wgrad_on_kernel(dy2.shape[0], Cout, Cin) and dy2.shape[0] >= 4096 and linear_wgrad_eligible(dy2, x2):
                g = getattr(ctx.wparam, "grad", None)
                g2 = g.reshape(Cout, Cin) if (g is not None and g.dtype == torch.bfloat16 and g.is_cuda and g.shape == weight.shape
                                              and g.reshape(Cout, Cin).data_ptr() == g.data_ptr()
                                              and g.reshape(Cout, Cin).is_contiguous() and g.data_ptr() % 16 == 0) else None
                with torch.cuda.device(dy.device):
                    if g2 is not None:
                        linear_wgrad(dy2, x2, out=g2, accumulate=True)
                    else:
                        dw = linear_wgrad(dy2, x2).reshape(weight.shape).to(weight.dtype)
            else:
                dw = (dy2.t() @ x2).reshape(weight.shape).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2, ctx.bparam)
        return dx, dw, db, (dy if ctx.has_add and ctx.needs_input_grad[3] else None)


class _AddChannelBias(torch.autograd.Function):
    """h (B,C,H,W) + v broadcast over the spatial positions — v: (C,) the bias of a convolution that stays on MIOpen, or (B,C)
    the ResidualBlock's time-embedding projection (src/models/sdxl/denoiser.py:524-551: `h + emb_out[..., None, None]`).
    The forward is ATen's broadcast add; the BACKWARD is why this exists (round 4): autograd would sum the gradient over
    (N,H,W) / (H,W) with ATen's long-reduction kernel — workgroups meeting through a semaphore buffer zeroed by a memset node
    — and in replayed HIP graphs those sums came back non-finite after the process had synchronised an idle stream (24-27
    parameters of the SDXL full fine-tune).  Here they are column sums of the channels-last gradient on vft_colsum
    (ordered partials, no atomics, no memset)."""

    @staticmethod
    def forward(ctx, h, v):
        ctx.per_batch = v.dim() == 2
        ctx.vdtype = v.dtype
        return h + (v.view(v.shape[0], v.shape[1], 1, 1) if ctx.per_batch else v.view(1, -1, 1, 1)).to(h.dtype)

    @staticmethod
    def backward(ctx, g):
        gv = None
        if ctx.needs_input_grad[1]:
            B, C, H, W = g.shape
            gc = g if g.is_contiguous(memory_format=torch.channels_last) else g.contiguous(memory_format=torch.channels_last)
            rows = gc.permute(0, 2, 3, 1).reshape(B, H * W, C)
            if gc.dtype != torch.bfloat16:
                rows = rows.to(torch.bfloat16)
            if ctx.per_batch and C % 8 == 0 and rows.is_contiguous() and rows.data_ptr() % 16 == 0:
                lib = _C.lib()
                gv = torch.empty((B, C), dtype=torch.bfloat16, device=g.device)
                scratch = torch.empty(B * int(lib.vft_norm_bwd_params_scratch_floats(H * W, C)), dtype=torch.float32, device=g.device)
                with torch.cuda.device(g.device), _Timed("colsum", 2.0 * B * H * W * C):
                    st = lib.vft_colsum_batched(_ptr(rows), _ptr(gv), _ptr(scratch), B, H * W, C, C, 0, _stream())
                _C.check(st, "vft_colsum_batched")
                gv = gv.to(ctx.vdtype)
            elif ctx.per_batch:
                gv = torch.stack([colsum(rows[b]) for b in range(B)]).to(ctx.vdtype)
            else:
                gv = colsum(rows.reshape(B * H * W, C)).to(ctx.vdtype)
        return (g if ctx.needs_input_grad[0] else None), gv


def add_channel_bias(h: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """h + v[..., None, None] with the gradient of v reduced by vft_colsum (see _AddChannelBias).  bf16 activations only (the
    trainers' dtype): any other dtype keeps ATen's broadcast add and type promotion, as the reference computes it."""
    if not (h.is_cuda and v.requires_grad and torch.is_grad_enabled() and h.dtype == torch.bfloat16):
        return h + (v.view(v.shape[0], v.shape[1], 1, 1) if v.dim() == 2 else v.view(1, -1, 1, 1))
    return _AddChannelBias.apply(h, v)


def colsum(dy2: torch.Tensor, param: torch.Tensor | None = None) -> torch.Tensor | None:
    """Column sums of a (rows, C) bf16 matrix on vft_colsum — the bias gradient of a trainable projection / convolution.
    With `param` (the bias parameter) whose `.grad` is an existing dense bf16 tensor the sums are ACCUMULATED into it and
    None is returned (nothing left for autograd to add); otherwise the sums are returned in the parameter's dtype."""
    rows, C = dy2.shape
    if dy2.dtype == torch.bfloat16 and dy2.is_contiguous() and C in (1, 2, 4) and rows % (8 // C) == 0 and rows >= 8 // C:
        # narrow tensors (the 4-channel head convolution): 8 / C consecutive rows are one 8-column row of the same memory;
        # fold the 8 sums afterwards (a (8 / C, C) add: no reduction kernel)
        k = 8 // C
        folded = colsum(dy2.view(rows // k, 8), None).view(k, C)
        out = folded[0]
        for i in range(1, k):
            out = out + folded[i]
        return out if param is None else out.to(param.dtype)
    # (round 4: no row-count threshold any more — below 1 024 rows this used to hand the sum to ATen, whose long reductions
    #  meet through a semaphore buffer + memset node that replayed HIP graphs cannot be trusted with)
    if (dy2.dtype != torch.bfloat16 or dy2.stride(1) != 1 or C % 8 or dy2.stride(0) % 8 or dy2.data_ptr() % 16 or rows < 1):
        out = dy2.sum(0)
        return out if param is None else out.to(param.dtype)
    g = _grad_target(param)
    odt = torch.float32 if (param is not None and param.dtype == torch.float32) else torch.bfloat16
    out = g if g is not None else torch.empty(C, dtype=odt, device=dy2.device)
    lib = _C.lib()
    scratch = torch.empty(int(lib.vft_norm_bwd_params_scratch_floats(rows, C)), dtype=torch.float32, device=dy2.device)
    with torch.cuda.device(dy2.device), _Timed("colsum", 2.0 * rows * C):
        st = lib.vft_colsum(_ptr(dy2), _ptr(out), _ptr(scratch), rows, C, dy2.stride(0), int(g is not None),
                            int(out.dtype == torch.float32), _stream())
    _C.check(st, "vft_colsum")
    if g is not None:
        return None
    return out if param is None else out.to(param.dtype)


def linear_wgrad_eligible(dy2: torch.Tensor, x2: torch.Tensor) -> bool:
    """2-D bf16 operands with unit column stride whose extents the kernel's 32-bit offsets cover."""
    if dy2.dim() != 2 or x2.dim() != 2 or dy2.dtype != torch.bfloat16 or x2.dtype != torch.bfloat16:
        return False
    M, R = dy2.shape
    Cc = x2.shape[1]
    if dy2.stride(1) != 1 or x2.stride(1) != 1 or R % 8 or Cc % 8 or dy2.stride(0) % 8 or x2.stride(0) % 8:
        return False
    if (dy2.data_ptr() | x2.data_ptr()) & 15:
        return False
    return (M + 130) * dy2.stride(0) * 2 < 2**31 and (M + 130) * x2.stride(0) * 2 < 2**31


def linear_wgrad(dy2: torch.Tensor, x2: torch.Tensor, out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    """dW[R, Cc] (+)= dy2[M, R]ᵀ · x2[M, Cc] on the hand-written split-reduction kernel (vft_linear_wgrad): the weight
    gradient of a trainable projection.  `out` (bf16 or fp32, row-major, may be a view into a flat gradient buffer)
    receives the result, added to its previous content when `accumulate`."""
    _require_gpu(dy2, x2)
    M, R = dy2.shape
    Cc = x2.shape[1]
    if out is None:
        out = torch.empty((R, Cc), dtype=torch.bfloat16, device=dy2.device)
        accumulate = False
    if out.dim() != 2 or out.shape != (R, Cc) or out.stride(1) != 1 or out.dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("linear_wgrad: `out` must be a row-major (R, Cc) bf16 / fp32 matrix")
    lib = _C.lib()
    scratch = torch.empty(int(lib.vft_linear_wgrad_scratch_floats(M, R, Cc)), dtype=torch.float32, device=dy2.device)
    with torch.cuda.device(dy2.device), _Timed("linear_wgrad", 2.0 * M * R * Cc):
        st = lib.vft_linear_wgrad(_ptr(dy2), _ptr(x2), _ptr(out), _ptr(scratch), M, R, Cc, dy2.stride(0), x2.stride(0),
                                  out.stride(0), int(bool(accumulate)), int(out.dtype == torch.float32), _stream())
    _C.check(st, "vft_linear_wgrad")
    return out


def _conv1x1_wt(weight: torch.Tensor) -> torch.Tensor:
    """(Cin, Cout) transpose of a frozen 1x1 convolution weight, cached on the parameter."""
    hit = getattr(weight, "_vft_wt", None)
    if hit is None or hit[0] != weight._version or hit[1] != weight.data_ptr():
        hit = (weight._version, weight.data_ptr(), weight.detach().reshape(weight.shape[0], weight.shape[1]).t().contiguous())
        if not torch.cuda.is_current_stream_capturing():
            weight._vft_wt = hit
    return hit[2]


def conv1x1(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None, add: torch.Tensor | None = None) -> torch.Tensor:
    """add + 1x1 / stride 1 convolution of a (B,Cin,H,W) tensor, channels-last in and out (Cin, Cout multiples of 64)."""
    _ready(x, weight)
    _require_gpu(x, weight, bias)
    return _Conv1x1.apply(x, weight, bias, add)


def conv3x3(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """3x3 / stride 1 / padding 1 convolution of a (B,Cin,H,W) tensor, channels-last in and out."""
    _ready(x, weight)
    _require_gpu(x, weight, bias)
    return _Conv3x3.apply(x, weight, bias)


class NF4State:
    """Device-side quantisation state of one NF4 weight (bitsandbytes layout, see modules/quant/nf4.py)."""

    __slots__ = ("packed", "absmax", "absmax_is_u8", "nested_absmax", "nested_code", "code", "offset", "out_features",
                 "in_features", "_absmax_f32")

    def __init__(self, packed, absmax, nested_absmax, nested_code, code, offset, out_features, in_features):
        self.packed, self.absmax = packed, absmax
        self.absmax_is_u8 = absmax.dtype == torch.uint8
        self.nested_absmax, self.nested_code, self.code = nested_absmax, nested_code, code
        self.offset, self.out_features, self.in_features = float(offset), int(out_features), int(in_features)
        self._absmax_f32 = None

    def absmax_f32(self) -> torch.Tensor:
        """One fp32 absmax per 64 weights; a double-quantised absmax (uint8 codes, nested fp32 absmax per 256
        blocks, offset) is expanded once: absmax = nested_code[q8] * nested_absmax[blk // 256] + offset."""
        if self._absmax_f32 is None:
            if self.absmax_is_u8:
                n = self.absmax.numel()
                nested = self.nested_absmax.float().repeat_interleave(256)[:n]
                self._absmax_f32 = (self.nested_code.float()[self.absmax.long()] * nested + self.offset).contiguous()
            else:
                self._absmax_f32 = self.absmax.float().contiguous()
        return self._absmax_f32


# NF4 GEMM form (modules/quant/nf4.py documents the switch): True = the launch dequantises the weight ONCE into a transient bf16
# workspace (all workgroups share the work, grid barrier, then the bf16 main loop of lin8); False = round 3's form, codes
# dequantised inside the K-loop by every row tile.  Either way the 4-bit codes are the only copy that persists in HBM.
NF4_DEQUANT_ONCE = os.environ.get("VFT_NF4_INLOOP", "0") == "0"
_NF4_WS: dict = {}  # device -> list of workspaces ever handed out (never freed: captured graphs keep their addresses)
_NF4_LAST_STREAM: dict = {}  # device -> the stream the scratch was last used on (eager launches and graph replays)


def _nf4_workspace(device: torch.device, nbytes: int) -> torch.Tensor:
    """One scratch per device, shared by every NF4 layer: it holds the transient bf16 image of ONE weight and, in its first
    16 KiB, the grid barrier's words (launch counter, arrival words, sticky error flag; zero at first use).  Launches on one
    stream are ordered and a launch is done with the scratch when it ends; a launch that arrives on ANOTHER stream is made
    to wait for everything the previous stream has queued (`_nf4_order_streams`: an event recorded there, waited for here),
    so two NF4 GEMMs of one device never overlap whatever streams their callers use (validation / sampling on a
    side stream, a user-attached encoder, a changed DP schedule).  Keyed by device and not by stream on purpose — the eager
    warm-up (side stream) creates and zeroes it, so the capture that follows finds it and records no 32 MB fill into the
    graph; inside a capture the graph's own edges order the launches (torch.cuda.graph() synchronises the device on entry),
    and a replay is announced by `nf4_note_replay`.  Grows by allocating a NEW buffer (the old one stays alive for graphs
    captured with it)."""
    pool = _NF4_WS.setdefault(device, [])
    if not pool or pool[-1].numel() < nbytes:
        pool.append(torch.zeros(max(nbytes, 32 << 20), dtype=torch.uint8, device=device))
    return pool[-1]


def _nf4_order_streams(device: torch.device) -> None:
    if torch.cuda.is_current_stream_capturing():
        return
    cur = torch.cuda.current_stream(device)
    last = _NF4_LAST_STREAM.get(device)
    if last is not None and last != cur:
        ev = torch.cuda.Event()
        ev.record(last)  # behind everything `last` has queued so far, its NF4 launches included
        cur.wait_event(ev)
    _NF4_LAST_STREAM[device] = cur


def nf4_note_replay(device: torch.device | None = None) -> None:
    """Call right before replaying a captured graph that may contain NF4 launches (GraphedTrainStep does): the replaying
    stream takes the scratch over from whatever stream used it last, with the same event dependency as an eager launch."""
    if _NF4_WS:
        for dev in ([device] if device is not None else list(_NF4_WS)):
            _nf4_order_streams(dev)


def nf4_barrier_timed_out() -> bool:
    """True if a dequantising launch ever gave up waiting at its grid barrier (the sticky error word of every scratch; a
    host read = a synchronisation: call it where the host reads the loss anyway).  Such a launch wrote NaN to its output
    instead of computing on the incomplete weight image."""
    return any(int(ws[:16].view(torch.int32)[2]) != 0 for pool in _NF4_WS.values() for ws in pool)


def nf4_check_barrier() -> None:
    """Raise if `nf4_barrier_timed_out()`: the trainers call this at their logging / checkpoint cadence."""
    if _NF4_WS and nf4_barrier_timed_out():
        raise RuntimeError(
            "vft_linear_nf4: a dequantising launch timed out at its grid barrier (part of its persistent grid was kept off "
            "the chip — a concurrent kernel on another stream or process, or a CU mask); its outputs were written as NaN. "
            "Set VFT_NF4_INLOOP=1 (in-loop dequantisation, no grid barrier) or remove the concurrent work.")


def nf4_reset_barrier() -> None:
    """Zero the barrier words of every scratch (tests that provoke a time-out: the arrival counts of that launch are
    incomplete for good).  Synchronises."""
    torch.cuda.synchronize()
    for pool in _NF4_WS.values():
        for ws in pool:
            ws[:16384].zero_()
    torch.cuda.synchronize()


def _nf4_gemm(a2: torch.Tensor, q: NF4State, bias, transpose_w: bool, ext_x=None, ext_w=None, residual=None) -> torch.Tensor:
    """NF4-weight GEMM (vft_linear_nf4), optional fused rank extension / residual."""
    M = a2.shape[0]
    N = q.in_features if transpose_w else q.out_features
    K = q.out_features if transpose_w else q.in_features
    if K % 64 != 0:
        # a reduction dimension the dequantising GEMM does not take (none in the three models): dense copy + library GEMM
        w = nf4_dequantize(q)
        out = a2 @ (w if transpose_w else w.t())
        if bias is not None:
            out = out + bias
        if ext_x is not None:
            out = out + ext_x @ ext_w.t()
        return out if residual is None else out + residual.reshape(out.shape)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=a2.device)
    r = ext_x.shape[1] if ext_x is not None else 0
    ws, ws_bytes = None, 0
    if NF4_DEQUANT_ONCE:
        ws_bytes = int(_C.lib().vft_linear_nf4_workspace(q.out_features, q.in_features))
        ws = _nf4_workspace(a2.device, ws_bytes)
        _nf4_order_streams(a2.device)
    with torch.cuda.device(a2.device), _Timed("gemm_nf4", 2.0 * M * N * (K + r)):
        st = _C.lib().vft_linear_nf4(_ptr(a2), _ptr(out), _ptr(bias), _ptr(residual), _ptr(ext_x), _ptr(ext_w), _ptr(q.packed),
                                     _ptr(q.absmax_f32()), _ptr(q.code), q.out_features, q.in_features, M, r, a2.stride(0),
                                     out.stride(0), residual.stride(0) if residual is not None else 0, int(transpose_w),
                                     _ptr(ws), ws_bytes, _stream())
    _C.check(st, "vft_linear_nf4")
    return out


def nf4_dequantize(q: NF4State) -> torch.Tensor:
    out = torch.empty((q.out_features, q.in_features), dtype=torch.bfloat16, device=q.packed.device)
    with torch.cuda.device(out.device):
        st = _C.lib().vft_nf4_dequant(_ptr(out), _ptr(q.packed), _ptr(q.absmax), int(q.absmax_is_u8), _ptr(q.nested_absmax),
                                      _ptr(q.nested_code), _ptr(q.code), q.offset, q.out_features, q.in_features, _stream())
    _C.check(st, "vft_nf4_dequant")
    return out


class _NF4Linear(torch.autograd.Function):
    """y = x · dequant(W)ᵀ + b with the dequantisation fused into the GEMM prologue; backward gives
    dx = dy · dequant(W) through the transposed-staging variant (the 4-bit weight is frozen)."""

    @staticmethod
    def forward(ctx, x, q: NF4State, bias):
        x2 = _rows(x)
        y2 = _nf4_gemm(x2, q, bias, False)
        ctx.q = q
        ctx.xshape = x.shape
        return y2.view(*x.shape[:-1], q.out_features)

    @staticmethod
    def backward(ctx, dy):
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _nf4_gemm(_rows(dy), ctx.q, None, True).view(ctx.xshape)
        return dx, None, None


def nf4_linear(x: torch.Tensor, q: NF4State, bias: torch.Tensor | None) -> torch.Tensor:
    _ready(x)
    _require_gpu(x, bias)
    if x.shape[-1] != q.in_features:
        raise ValueError(f"nf4_linear: input has {x.shape[-1]} features, weight expects {q.in_features}")
    return _NF4Linear.apply(x, q, bias)
