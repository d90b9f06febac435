"""Real-image front half — CPU restatement.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the resampling numerics: the reference resizes with torchvision 0.22 (`v2` transforms,
src/dataset/text_to_image.py:140-155, src/dataset/transform.py:35-58), which is absent from /root/reference and from this
image, and no reference test holds a resized image.  What is restated is torchvision's published tensor path:
`resize(..., BICUBIC, antialias=True)` on a float tensor = ATen `_upsample_bicubic2d_aa` (separable, Keys cubic
a = -0.5, support widened by the downscale factor, weights normalised per output pixel, half-pixel centres), computed
in fp32 and cast back to the input dtype.  The integer work around it (cover size, crop window, size conditions) is
exact by construction and follows the reference lines cited below."""
from __future__ import annotations

import math

import numpy as np


def cover_size(w: int, h: int, target_w: int, target_h: int) -> tuple[int, int]:
    """transform.py:43-49 — (scaled_h, scaled_w) = ceil(side * max(tw / w, th / h))."""
    f = max(target_w / w, target_h / h)
    return math.ceil(h * f), math.ceil(w * f)


def _cubic(x: np.ndarray, a: float = -0.5) -> np.ndarray:
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                    np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def _aa_weights(in_size: int, out_size: int) -> list[tuple[int, np.ndarray]]:
    scale = in_size / out_size
    support = 2.0 * scale if scale >= 1.0 else 2.0
    invscale = 1.0 / scale if scale >= 1.0 else 1.0
    rows = []
    for i in range(out_size):
        center = scale * (i + 0.5)
        xmin = max(0, int(center - support + 0.5))
        xsize = min(in_size, int(center + support + 0.5)) - xmin
        w = _cubic((np.arange(xsize) + xmin - center + 0.5) * invscale).astype(np.float32)
        rows.append((xmin, w / w.sum(dtype=np.float32)))
    return rows


def resize_bicubic_aa(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img (C,H,W) float -> (C,out_h,out_w) float32; horizontal pass then vertical pass, fp32 accumulation."""
    c, h, w = img.shape
    x = img.astype(np.float32)
    tmp = np.empty((c, h, out_w), np.float32)
    for j, (x0, wt) in enumerate(_aa_weights(w, out_w)):
        tmp[:, :, j] = (x[:, :, x0:x0 + len(wt)] * wt).sum(-1, dtype=np.float32)
    out = np.empty((c, out_h, out_w), np.float32)
    for i, (y0, wt) in enumerate(_aa_weights(h, out_h)):
        out[:, i, :] = (tmp[:, y0:y0 + len(wt), :] * wt[None, :, None]).sum(1, dtype=np.float32)
    return out


def normalise_half(u8: np.ndarray) -> np.ndarray:
    """text_to_image.py:142-148 — uint8 -> fp16 / 255 -> (x - 0.5) / 0.5, every step rounded to fp16."""
    x = (u8.astype(np.float16) / np.float16(255.0)).astype(np.float16)
    return ((x - np.float16(0.5)).astype(np.float16) / np.float16(0.5)).astype(np.float16)


def size_conditions(resized_hw: tuple[int, int], top: int, left: int, crop_hw: tuple[int, int]) -> dict:
    """text_to_image.py:207-217 — original_size is the RESIZED (h, w), target_size the crop (h, w), crop coords (top, left)."""
    return {"original_size": [resized_hw[0], resized_hw[1]], "target_size": [crop_hw[0], crop_hw[1]],
            "crop_coords_top_left": [top, left]}
