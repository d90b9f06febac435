"""Bucket-as-batch datasets (reference: src/dataset/bucket.py:45-121): one Dataset item is one full
batch drawn from a single bucket; indices wrap around so every batch is full-size."""
from __future__ import annotations

import math
from abc import ABC
from collections.abc import Sequence

import numpy as np
import torch.utils.data as data


class Bucket(ABC):
    def __init__(self, items: Sequence, batch_size: int, num_repeats: int = 1):
        self.items = items
        self.num_items = len(items)
        self.batch_size = batch_size
        self.num_repeats = num_repeats

    def __len__(self) -> int:
        return self.num_items * self.num_repeats

    def to_local_idx(self, idx: int | slice) -> int | list:
        if isinstance(idx, slice):
            start, stop, step = idx.indices(10**10)
            return (np.arange(start, stop, step) % self.num_items).tolist()
        return idx % self.num_items

    def __getitem__(self, idx: int | slice):
        return self.items[self.to_local_idx(idx)]  # type: ignore[index]


class BucketDataset(data.Dataset):
    def __init__(self, bucket: Bucket):
        self.bucket = bucket
        self.num_samples = math.ceil(len(bucket) / bucket.batch_size)

    def __len__(self) -> int:
        return self.num_samples

    def __getitem__(self, idx: int):
        first = (idx % self.bucket.num_items) * self.bucket.batch_size
        return self.bucket[first : first + self.bucket.batch_size]
