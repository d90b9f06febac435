"""ctypes binding of libvft_hip.so (the C ABI declared in include/vft_hip.h).

The library is built in-tree by ``vision_ft_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no fallback: if the shared object is missing, ``lib()`` raises, and every op in
``vision_ft_amd.kernels`` raises with it.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFT_LIB_PATH") or os.path.join(_HERE, "libvft_hip.so")  # (the override is for A/B builds)

_lib: ctypes.CDLL | None = None


class VftError(RuntimeError):
    """A C-ABI entry point returned non-zero."""


_I64P = POINTER(c_int64)
_I32P = POINTER(c_int32)

# name -> (restype, argtypes); kept in the order of include/vft_hip.h
SIGNATURES: dict[str, tuple] = {
    "vft_version": (c_int, []),
    "vft_last_error": (c_char_p, []),
    "vft_device_info": (c_int, [c_int, c_char_p, c_int]),
    "vft_attn_fwd": (
        c_int,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        + [c_int] * 6
        + [_I64P] * 4
        + [c_float, c_void_p],
    ),
    "vft_attn_bwd": (
        c_int,
        [c_void_p] * 11 + [c_int] * 6 + [_I64P] * 8 + [c_float, c_int64, c_void_p],
    ),
    "vft_attn_bwd_workspace": (c_int64, [c_int] * 6),
    "vft_lora_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_int64, c_int64, c_int64, c_float, c_void_p]),
    "vft_lora_down": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_int64, c_float, c_void_p]),
    "vft_lora_down_variant": (c_int, [c_int]),
    "vft_lora_wgrad_target": (c_int, [c_int]),
    "vft_lora_bwd_dx": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_int64, c_int64, c_float, c_void_p]),
    "vft_lora_wgrad_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vft_lora_wgrad2": (c_int, [c_void_p] * 7 + [c_size_t] + [c_int] * 4 + [c_int64, c_int64, c_float, c_void_p]),
    "vft_lora_wgrad2g": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p, c_size_t] + [c_int] * 4 + [c_int64, c_int64, c_float, c_void_p]),
    "vft_lora_wgrad_desc_bytes": (c_size_t, []),
    "vft_lora_wgrad2g_defer": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p, c_size_t] + [c_int] * 4 + [c_int64, c_int64, c_float, c_void_p, c_int, c_void_p]),
    "vft_lora_wgrad_reduce_many": (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    "vft_lora_u_part_floats": (c_int64, [c_int, c_int, c_int]),
    "vft_lora_bwd_workspace": (c_size_t, [c_int, c_int, c_int, c_int]),
    "vft_lora_bwd_dy": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int64, c_float, c_float, c_void_p, c_int, c_void_p]),
    "vft_lora_bwd_x": (c_int, [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int64, c_float, c_void_p, c_int, c_void_p]),
    "vft_lora_pack": (c_int, [c_void_p, c_int, c_void_p]),
    "vft_norm_fwd": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int64, c_int, c_int, c_float, c_void_p]),
    "vft_norm_bwd_dx": (c_int, [c_void_p] * 9 + [c_int64, c_int, c_int64, c_int, c_int, c_void_p]),
    "vft_linear_wgrad_scratch_floats": (c_int64, [c_int64, c_int, c_int]),
    "vft_linear_wgrad": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "vft_conv3x3_wgrad_scratch_floats": (c_int64, [c_int64, c_int, c_int]),
    "vft_conv3x3_wgrad": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "vft_norm_bwd_params_scratch_floats": (c_int64, [c_int64, c_int]),
    "vft_colsum": (c_int, [c_void_p] * 3 + [c_int64, c_int, c_int64, c_int, c_int, c_void_p]),
    "vft_colsum_batched": (c_int, [c_void_p] * 3 + [c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "vft_norm_bwd_params": (c_int, [c_void_p] * 8 + [c_int64, c_int, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "vft_qk_norm_rope_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [_I64P, c_int64, c_int, c_float, c_void_p]),
    "vft_qk_norm_rope_bwd": (c_int, [c_void_p] * 8 + [c_int] * 4 + [_I64P, _I64P, c_int64, c_int, c_void_p]),
    "vft_groupnorm_scratch_floats": (c_int64, [c_int] * 6),
    "vft_groupnorm_fwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "vft_groupnorm_bwd": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "vft_groupnorm_nhwc_fwd": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "vft_groupnorm_nhwc_bwd": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_float, c_int, c_void_p]),
    "vft_geglu_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vft_geglu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vft_swiglu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vft_swiglu_bwd": (c_int, [c_void_p] * 5 + [c_int64, c_void_p]),
    "vft_gate_residual": (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_void_p]),
    "vft_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p]),
    "vft_mean_sq_scratch_floats": (c_int64, []),
    "vft_mean_sq": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "vft_sum_sq_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "vft_linear_nf4": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_int64] * 3 + [c_int, c_void_p, c_int64, c_void_p]),
    "vft_linear_nf4_workspace": (c_int64, [c_int, c_int]),
    "vft_conv3x3_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "vft_conv3x3_s2_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "vft_linear_fwd": (c_int, [c_void_p] * 7 + [c_int] * 4 + [c_int64] * 4 + [c_void_p]),
    "vft_linear_variant": (c_int, [c_int]),
    "vft_linear_tile_hint": (c_int, [c_int] * 5),
    "vft_attn_variant": (c_int, [c_int]),
    "vft_nf4_dequant": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_float, c_int, c_int, c_void_p]),
}


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C vision_ft_amd/csrc`). "
                "vision_ft_amd has no CPU fallback."
            )
        cdll = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)
            fn.restype = res
            fn.argtypes = args
        _lib = cdll
        if os.environ.get("VFT_NO_TILE_TABLE", "") != "1":
            load_tile_table()
    return _lib


TILE_TABLE = os.path.join(_HERE, "tuning_tables", "gfx950_lin8_tiles.csv")


def load_tile_table(path: str | None = None) -> int:
    """Register the measured tile choice of every (M, N, K, conv) row of the committed table with the library
    (vft_linear_tile_hint; tools/tune_lin8.py writes the table).  Shapes that are not listed keep the analytic choice.
    Returns the number of rows registered."""
    path = path or TILE_TABLE
    if not os.path.exists(path):
        return 0
    n = 0
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#") or line.startswith("M,"):
                continue
            m, nn, k, conv, cfg = (int(v) for v in line.split(",")[:5])
            check(lib().vft_linear_tile_hint(m, nn, k, conv, cfg), "vft_linear_tile_hint")
            n += 1
    return n


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().vft_last_error().decode("utf-8", "replace")
        if status == 1:
            raise ValueError(f"{what}: {msg}")
        raise VftError(f"{what}: {msg}")


def i64x3(values) -> ctypes.Array:
    return (c_int64 * 3)(*[int(v) for v in values])
