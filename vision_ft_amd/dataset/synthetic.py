"""Synthetic bucketed latents: the fixed-shape stand-in for TextToImageBucket + VAE/text encoders
(src/dataset/text_to_image.py:182-230, src/models/*/pipeline.py encode_*), as BASELINE.json's
north-star prescribes.  Every batch comes from ONE aspect-ratio bucket, so shapes are static per
batch; contents are seeded standard-normal tensors generated on the target device."""
from __future__ import annotations

import numpy as np
import torch
import torch.utils.data as data

from .aspect_ratio_bucket import AspectRatioBucketConfig, generate_buckets
from .bucket import Bucket, BucketDataset


def headline_buckets(base: int = 1024, step: int = 128, min_size: int = 384, lo: int = 640, hi: int = 1536,
                     min_area_frac: float = 0.9) -> list[tuple[int, int]]:
    """The {640x1536 ... 1536x640} set of BASELINE config 5: unique rows of generate_buckets with
    area >= 0.9 * base^2 and both sides in [lo, hi], sorted by width."""
    rows = generate_buckets(base * base, base, step, min_size)
    keep = {(int(w), int(h)) for w, h in rows if w * h >= min_area_frac * base * base and lo <= w <= hi and lo <= h <= hi}
    return sorted(keep)


class SyntheticLatentBucket(Bucket):
    """`num_items` virtual samples of one (width, height) bucket; items are just integer seeds."""

    def __init__(self, width: int, height: int, num_items: int, batch_size: int, num_repeats: int = 1):
        super().__init__(np.arange(num_items), batch_size, num_repeats)
        self.width, self.height = width, height


class SyntheticLatentDatasetConfig(AspectRatioBucketConfig):
    """YAML-selectable dataset producing {"bucket": (w,h), "seeds": [...]} batches."""

    samples_per_bucket: int = 64
    buckets_wh: list[tuple[int, int]] | None = None
    num_repeats: int = 1

    def get_buckets(self) -> list[SyntheticLatentBucket]:
        wh = self.buckets_wh or headline_buckets(self.bucket_base_size, self.step, self.min_size)
        return [SyntheticLatentBucket(w, h, self.samples_per_bucket, self.batch_size, self.num_repeats) for w, h in wh]

    def get_dataset(self) -> data.Dataset:
        parts = []
        for b in self.get_buckets():
            parts.append(_Tagged(BucketDataset(b), b.width, b.height))
        return data.ConcatDataset(parts)


class _Tagged(data.Dataset):
    def __init__(self, inner: BucketDataset, width: int, height: int):
        self.inner, self.width, self.height = inner, width, height

    def __len__(self) -> int:
        return len(self.inner)

    def __getitem__(self, idx: int) -> dict:
        seeds = self.inner[idx]
        # every value is a list so the bucketing collate (list concatenation) applies, as for the reference's items
        return {"width": [self.width], "height": [self.height], "seeds": list(map(int, seeds))}


def make_latents(width: int, height: int, batch: int, channels: int, device, seed: int, dtype=torch.bfloat16,
                 compression: int = 8) -> torch.Tensor:
    g = torch.Generator(device=device).manual_seed(seed)
    return torch.randn(batch, channels, height // compression, width // compression, generator=g, device=device,
                       dtype=torch.float32).to(dtype)
