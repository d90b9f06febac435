"""DataLoader factories (reference: src/dataloader.py:18-108): the bucket datasets emit whole
batches, so the loader runs with batch_size=1 and *concatenates* instead of stacking."""
from __future__ import annotations

from collections import defaultdict
from typing import Callable, Iterable

import torch
import torch.utils.data as data


def concatnate_collate_fn(batch: Iterable[dict]) -> dict:  # (sic) reference spelling
    merged: dict[str, list] = defaultdict(list)
    for item in batch:
        for k, v in item.items():
            merged[k].append(v)
    out = {}
    for k, vs in merged.items():
        out[k] = torch.cat(vs, dim=0) if isinstance(vs[0], torch.Tensor) else sum(vs, [])
    return out


def preview_batch_collate_fn(batch: Iterable[dict]) -> dict:
    merged: dict[str, list] = defaultdict(list)
    for item in batch:
        for k, v in item.items():
            merged[k].append(v)
    for k, vs in merged.items():
        assert len(vs) == 1, "Preview batch size must be 1"
    return {k: vs[0] for k, vs in merged.items()}


def get_dataloader(dataset: data.Dataset, batch_size: int, shuffle: bool = True, num_workers: int = 0,
                   drop_last: bool = False, generator: torch.Generator | None = None,
                   collate_fn: Callable | None = None) -> data.DataLoader:
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                           drop_last=drop_last, generator=generator, collate_fn=collate_fn)


def get_dataloader_for_bucketing(dataset: data.Dataset, shuffle: bool = True, num_workers: int = 0,
                                 drop_last: bool = False, generator: torch.Generator | None = None) -> data.DataLoader:
    return get_dataloader(dataset, 1, shuffle, num_workers, drop_last, generator, concatnate_collate_fn)


def get_dataloader_for_preview(dataset: data.Dataset, num_workers: int = 0, drop_last: bool = False,
                               generator: torch.Generator | None = None) -> data.DataLoader:
    return get_dataloader(dataset, 1, False, num_workers, drop_last, generator, preview_batch_collate_fn)
