"""Aspect-ratio buckets (integer work; bit-exact with src/dataset/aspect_ratio_bucket.py:19-152).

Kept quirks, pinned by tests/golden/index.json:
* ``generate_buckets`` walks EVERY height from the rounded one down to (exclusive) ``min_size`` for
  each width and keeps duplicates (base 1024 / step 64 / min 384 -> 420 rows);
* ``find_nearest`` scans buckets by descending area and stops at the first bucket whose aspect
  difference grows (so 1920x1080 lands in 1280x832, not the closest-ratio bucket overall).
"""
from __future__ import annotations

from abc import ABC

import numpy as np
import torch.utils.data as data

from .bucket import Bucket
from .util import DatasetConfig


class AspectRatioBucket(Bucket, ABC):
    width: int
    height: int


def generate_buckets(target_area: int = 1024 * 1024, start_size: int = 1024, step: int = 64, min_size: int = 64) -> np.ndarray:
    rows: list[tuple[int, int]] = []
    for w in range(start_size, min_size - 1, -step):
        top = round(target_area / w / step) * step
        if top < min_size:
            break
        for h in range(top, min_size, -step):
            rows.append((w, h))
            if w != top:
                rows.append((h, w))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 2)


class AspectRatioBucketConfig(DatasetConfig):
    bucket_base_size: int = 1024
    step: int = 64
    min_size: int = 384

    @property
    def buckets(self) -> np.ndarray:
        return generate_buckets(target_area=self.bucket_base_size**2, start_size=self.bucket_base_size,
                                step=self.step, min_size=self.min_size)

    def generate_buckets(self) -> list[AspectRatioBucket]:
        raise NotImplementedError

    def get_dataset(self) -> data.Dataset:
        raise NotImplementedError


class AspectRatioBucketManager:
    def __init__(self, buckets: np.ndarray):
        self.buckets = buckets
        self.aspect_ratios = buckets[:, 0] / buckets[:, 1]
        self.resolutions = buckets[:, 0] * buckets[:, 1]
        self._by_area = np.argsort(-self.resolutions)

    def __len__(self) -> int:
        return self.buckets.shape[0]

    def __iter__(self):
        for w, h in self.buckets:
            yield w, h

    def aspect_ratio(self, width: int, height: int) -> float:
        return width / height

    def find_nearest(self, width: int, height: int) -> int:
        want = width / height
        best, best_diff = None, float("inf")
        for idx in self._by_area:
            bw, bh = self.buckets[idx]
            if bw > width or bh > height:  # a bucket never upsamples
                continue
            diff = abs(want - self.aspect_ratios[idx])
            if best is not None and diff > best_diff:
                break
            best, best_diff = int(idx), diff
        assert best is not None
        return best


def print_arb_info(bucket_ds, print_fn=print) -> None:
    print_fn(f"===== {len(bucket_ds)} buckets =====")
    for i, b in enumerate(bucket_ds):
        print_fn(f"Bucket {i:>3} | {b.width:>6,}x{b.height:<6,} | {b.num_items:>8,} images")
