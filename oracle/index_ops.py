"""Integer / string work on the path (bit-exact): buckets, nearest-bucket lookup, bucket batches,
collate, LoRA target-key selection, adapter key extraction.  TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import math
import re

import numpy as np


# src/dataset/aspect_ratio_bucket.py:19-60
def generate_buckets(target_area: int = 1024 * 1024, start_size: int = 1024, step: int = 64,
                     min_size: int = 64) -> np.ndarray:
    rows = []
    w = start_size
    while w >= min_size:
        h_rounded = round((target_area / w) / step) * step
        if h_rounded < min_size:
            break
        # NB: the reference walks *every* h from h_rounded down to (exclusive) min_size and keeps
        # duplicates — do not "fix" this
        for h in range(h_rounded, min_size, -step):
            rows.append((w, h))
            if w != h_rounded:
                rows.append((h, w))
        w -= step
    return np.array(rows).reshape(-1, 2)


# src/dataset/aspect_ratio_bucket.py:124-152 (resolution-descending scan with the early break)
def find_nearest(buckets: np.ndarray, width: int, height: int) -> int:
    ars = buckets[:, 0] / buckets[:, 1]
    res = buckets[:, 0] * buckets[:, 1]
    provided = width / height
    min_diff = float("inf")
    best = None
    for idx in np.argsort(-res):
        bw, bh = buckets[idx]
        if bw > width or bh > height:
            continue
        diff = abs(provided - ars[idx])
        if diff > min_diff and best is not None:
            break
        min_diff = diff
        best = int(idx)
    assert best is not None
    return best


# src/dataset/bucket.py:78-85,112-121 — index lists of the batches a BucketDataset emits
def bucket_batch_indices(num_items: int, batch_size: int, num_repeats: int = 1) -> list[list[int]]:
    num_samples = math.ceil(num_items * num_repeats / batch_size)
    out = []
    for idx in range(num_samples):
        real = idx % num_items
        start = real * batch_size
        out.append([(start + j) % num_items for j in range(batch_size)])
    return out


# src/utils/state_dict.py:15-42
def get_target_keys(include, exclude, keys: list[str]) -> list[str]:
    """`include` / `exclude` entries are plain substrings (str) or {"regex": "..."} dicts
    (RegexMatch: re.match, anchored at the start).  Returns a *set-ordered* list like the reference;
    compare as sets."""
    matched: set[str] = set()
    for pat in include:
        if isinstance(pat, str):
            matched.update(k for k in keys if pat in k)
        else:
            rx = re.compile(pat["regex"])
            matched.update(k for k in keys if rx.match(k))
    for pat in exclude:
        if isinstance(pat, str):
            matched.difference_update([k for k in keys if pat in k])
        else:
            rx = re.compile(pat["regex"])
            matched.difference_update([k for k in keys if rx.match(k)])
    return list(matched)
