"""NF4 (bitsandbytes 4-bit NormalFloat) oracle in numpy.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in the third-party wheel ``bitsandbytes==0.48.2``
(uv.lock:307-308 of the reference), which is absent from /root/reference and from this image, and
the reference's own tests (tests/test_modules_quant.py:153-193) assert only class types, key names,
uint8 storage and reload self-consistency.  What follows restates the *published* algorithm
(QLoRA, Dettmers et al. 2023, and bitsandbytes' functional.quantize_4bit / dequantize_4bit /
quantize_blockwise) and is anchored on the reference's call sites:
  src/modules/quant/bnb.py:37-129        BnbLinear4bit (compress_statistics=True, uint8 storage)
  src/modules/quant/functional.py:342-371 quantize_state_dict -> quantize_4bit(...).as_dict(packed=True)
  src/modules/quant/functional.py:277-300 checkpoint key sniffing ("quant_state.bitsandbytes__nf4")

Wire format of a quantised Linear weight `W` (out, in) stored under `<name>.weight`:
  <name>.weight                                   uint8 ((out*in+1)//2, 1); element 2i in the HIGH nibble
  <name>.weight.absmax                            fp32 (nblocks,)           [no double quant]
                                                  uint8 (nblocks,)          [double quant]
  <name>.weight.quant_map                         fp32 (16,) NF4 code book
  <name>.weight.nested_absmax                     fp32 (ceil(nblocks/256),) [double quant]
  <name>.weight.nested_quant_map                  fp32 (256,) dynamic 8-bit map [double quant]
  <name>.weight.quant_state.bitsandbytes__nf4     uint8 JSON: quant_type, blocksize, dtype, shape,
                                                  nested_blocksize, nested_dtype, nested_offset
"""
from __future__ import annotations

import json

import numpy as np

# The 16 NF4 code values (QLoRA appendix E; bitsandbytes functional.get_4bit_type("nf4")).
NF4_CODE = np.array(
    [
        -1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453,
        -0.28444138169288635, -0.18477343022823334, -0.09105003625154495, 0.0,
        0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
        0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0,
    ],
    dtype=np.float32,
)


def dynamic_map_8bit() -> np.ndarray:
    """bitsandbytes functional.create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8)."""
    data: list[float] = []
    non_sign_bits = 7
    max_exponent_bits = 7
    additional_items = 2 ** (non_sign_bits - max_exponent_bits) - 1
    for i in range(max_exponent_bits):
        fraction_items = int(2 ** (i + non_sign_bits - max_exponent_bits) + 1)
        boundaries = np.linspace(0.1, 1, fraction_items, dtype=np.float32)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
        data += (-(10 ** (-(max_exponent_bits - 1) + i)) * means).tolist()
    if additional_items > 0:
        boundaries = np.linspace(0.1, 1, additional_items + 1, dtype=np.float32)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += means.tolist()
        data += (-means).tolist()
    data.append(0.0)
    data.append(1.0)
    data += [0.0] * (256 - len(data))
    data.sort()
    return np.array(data, dtype=np.float32)


def _nearest_code(x: np.ndarray, code: np.ndarray) -> np.ndarray:
    """index of the nearest code value (code sorted ascending); ties go to the lower index like a
    `x > midpoint` decision tree."""
    mid = (code[:-1].astype(np.float64) + code[1:].astype(np.float64)) / 2.0
    return np.searchsorted(mid, x.astype(np.float64), side="left").astype(np.uint8)


def quantize_blockwise_8bit(a: np.ndarray, blocksize: int = 256):
    code = dynamic_map_8bit()
    flat = a.astype(np.float32).ravel()
    n = flat.size
    nb = (n + blocksize - 1) // blocksize
    pad = np.zeros(nb * blocksize, dtype=np.float32)
    pad[:n] = flat
    blocks = pad.reshape(nb, blocksize)
    absmax = np.abs(blocks).max(axis=1)
    safe = np.where(absmax == 0, 1.0, absmax).astype(np.float32)
    q = _nearest_code(blocks / safe[:, None], code).ravel()[:n]
    return q, absmax.astype(np.float32), code


def dequantize_blockwise_8bit(q: np.ndarray, absmax: np.ndarray, code: np.ndarray, blocksize: int = 256):
    vals = code[q.astype(np.int64)]
    idx = np.arange(q.size) // blocksize
    return (vals * absmax[idx]).astype(np.float32)


def quantize_nf4(w: np.ndarray, blocksize: int = 64, double_quant: bool = True, dtype_name: str = "bfloat16") -> dict:
    """Returns the dict of arrays described in the module docstring, keyed by the suffix after
    `<name>.weight` ('' for the packed data)."""
    shape = list(w.shape)
    flat = w.astype(np.float32).ravel()
    n = flat.size
    assert n % blocksize == 0, "bitsandbytes pads the last block; the hot path only meets multiples"
    blocks = flat.reshape(-1, blocksize)
    absmax = np.abs(blocks).max(axis=1).astype(np.float32)
    safe = np.where(absmax == 0, 1.0, absmax)
    q = _nearest_code(blocks / safe[:, None], NF4_CODE).ravel()
    packed = ((q[0::2] << 4) | q[1::2]).astype(np.uint8).reshape(-1, 1)
    out = {"": packed, ".quant_map": NF4_CODE.copy()}
    meta = {"quant_type": "nf4", "blocksize": blocksize, "dtype": dtype_name, "shape": shape}
    if double_quant:
        offset = np.float32(absmax.mean(dtype=np.float32))
        q8, absmax2, code2 = quantize_blockwise_8bit(absmax - offset, 256)
        out[".absmax"] = q8
        out[".nested_absmax"] = absmax2
        out[".nested_quant_map"] = code2
        meta.update({"nested_blocksize": 256, "nested_dtype": "float32", "nested_offset": float(offset)})
    else:
        out[".absmax"] = absmax
    out[".quant_state.bitsandbytes__nf4"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8).copy()
    return out


def absmax_of(q: dict) -> np.ndarray:
    """fp32 per-block absmax of a quantised weight dict (undoing the double quantisation)."""
    meta = json.loads(bytes(q[".quant_state.bitsandbytes__nf4"].tolist()).decode())
    if ".nested_absmax" in q:
        am = dequantize_blockwise_8bit(q[".absmax"], q[".nested_absmax"], q[".nested_quant_map"],
                                       meta.get("nested_blocksize", 256))
        return (am + np.float32(meta["nested_offset"])).astype(np.float32)
    return q[".absmax"].astype(np.float32)


def dequantize_nf4(q: dict) -> np.ndarray:
    """fp32 dequantised weight with the recorded shape (the caller casts to the compute dtype)."""
    meta = json.loads(bytes(q[".quant_state.bitsandbytes__nf4"].tolist()).decode())
    bs = meta["blocksize"]
    packed = q[""].ravel()
    codes = np.empty(packed.size * 2, dtype=np.int64)
    codes[0::2] = packed >> 4
    codes[1::2] = packed & 0xF
    am = absmax_of(q)
    vals = q[".quant_map"][codes] * am[np.arange(codes.size) // bs]
    return vals.astype(np.float32).reshape(meta["shape"])
