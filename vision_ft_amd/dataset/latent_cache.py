"""Latent / text-embedding cache: the real-image front half of `train_step` with the frozen encoders taken out of the
training process ("needs torchvision/diffusers on the box, or latent caching").

The reference encodes captions and pixels inside every `train_step` (train/sdxl/text_to_image.py:101-113:
`text_encoder.encode_prompts` -> `prepare_encoder_hidden_states`, `encode_image` = VAE.encode().sample() * scaling,
src/models/sdxl/pipeline.py:189-205) on what `TextToImageBucket.__getitem__` (src/dataset/text_to_image.py:182-230)
hands it.  Both encoders are frozen and run under no_grad, so their outputs per (image file, crop window) can be written
once — by `LatentCacheWriter`, wherever a VAE / text encoder exists — and the MI355X box then trains from the cache with
no diffusers / transformers installed:

    TextToImageDatasetConfig (folder scan -> bucket -> cover-resize -> crop -> size conditions)
        -> LatentCacheWriter(image_encoder, prompt_encoder)     one safetensors file per (file, crop) + index.json
        -> CachedLatentDatasetConfig                            the same bucket batches, "latents" / "text.*" instead of
                                                                "image" / "caption"; size conditions as recorded

A cache entry is keyed by the image's path relative to the dataset folder and its crop window (the reference draws a new
random crop per epoch; `crops_per_image` windows are drawn per image at cache time, seeded per file, and one of them is
picked per visit).  Batch dict keys match `TextToImageBucket.__getitem__` except for the encoded pair."""
from __future__ import annotations

import hashlib
import json
import os
from typing import Callable

import torch
import torch.utils.data as data
from PIL import Image
from safetensors.torch import load_file, save_file

from .aspect_ratio_bucket import AspectRatioBucket, AspectRatioBucketConfig
from .bucket import BucketDataset
from .text_to_image import TextToImageDatasetConfig
from .transform import random_crop_params

INDEX_NAME = "index.json"
TEXT_PREFIX = "text."  # text-condition tensors travel as top-level "text.<name>" keys (the bucket collate concatenates values)


def entry_name(rel_path: str, width: int, height: int, top: int, left: int) -> str:
    """File name of one cache entry: image file (relative path, hashed) + bucket + crop window."""
    h = hashlib.sha1(rel_path.encode()).hexdigest()[:16]
    return f"{h}_{width}x{height}_t{top}_l{left}.safetensors"


def _file_seed(rel_path: str, seed: int) -> int:
    return (int.from_bytes(hashlib.sha1(rel_path.encode()).digest()[:4], "little") ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


class LatentCacheWriter:
    """Runs the frozen encoders over a `TextToImageDatasetConfig` once.

    image_encoder: (1,3,H,W) fp16 in -1..1 -> (1,C,H/8,W/8) latents with the scaling factor applied (`encode_image`);
    prompt_encoder: list[str] -> dict of tensors with a leading batch dimension (SDXL: "ehs", "pooled")."""

    def __init__(self, cache_folder: str, image_encoder: Callable, prompt_encoder: Callable, crops_per_image: int = 1,
                 seed: int = 0, latent_dtype: torch.dtype = torch.bfloat16):
        self.cache_folder, self.image_encoder, self.prompt_encoder = cache_folder, image_encoder, prompt_encoder
        self.crops_per_image, self.seed, self.latent_dtype = int(crops_per_image), int(seed), latent_dtype

    @torch.no_grad()
    def build(self, cfg: TextToImageDatasetConfig) -> dict:
        os.makedirs(self.cache_folder, exist_ok=True)
        entries = []
        for bucket in cfg.generate_buckets():
            for row in bucket.items:
                rel = os.path.relpath(row["image"], cfg.folder)
                with Image.open(row["image"]) as pil:
                    resized = bucket.resize_transform(pil)
                caption = row["caption"]
                for proc in bucket.caption_processors:
                    caption = proc(caption)
                text = {k: v.detach().cpu() for k, v in self.prompt_encoder([caption]).items()}
                gen = torch.Generator().manual_seed(_file_seed(rel, self.seed))
                seen = set()
                for _ in range(self.crops_per_image):
                    top, left, ch, cw = random_crop_params(resized, (bucket.height, bucket.width), generator=gen)
                    if (top, left) in seen:
                        continue
                    seen.add((top, left))
                    crop = resized[..., top:top + ch, left:left + cw]
                    lat = self.image_encoder(crop[None])[0].detach().to("cpu", self.latent_dtype).contiguous()
                    name = entry_name(rel, bucket.width, bucket.height, top, left)
                    save_file({"latents": lat, **{TEXT_PREFIX + k: v[0].contiguous() for k, v in text.items()}},
                              os.path.join(self.cache_folder, name))
                    entries.append({"entry": name, "file": rel, "bucket": [bucket.width, bucket.height], "crop": [top, left],
                                    "original_size": [int(resized.shape[1]), int(resized.shape[2])], "target_size": [ch, cw],
                                    "width": row["width"], "height": row["height"], "caption": caption})
        index = {"version": 1, "crops_per_image": self.crops_per_image, "seed": self.seed, "entries": entries}
        with open(os.path.join(self.cache_folder, INDEX_NAME), "w") as f:
            json.dump(index, f)
        return index


class CachedLatentBucket(AspectRatioBucket):
    """One aspect-ratio bucket of cached samples; an item = one image file with its cached crop windows."""

    def __init__(self, cache_folder: str, items: list[list[dict]], batch_size: int, width: int, height: int, num_repeats: int):
        super().__init__(items=items, batch_size=batch_size, num_repeats=num_repeats)
        self.cache_folder, self.width, self.height = cache_folder, width, height

    def __getitem__(self, idx: int | slice):
        local = self.to_local_idx(idx)
        groups = [self.items[i] for i in local] if isinstance(local, list) else [self.items[local]]
        rows = []
        for crops in groups:  # one of the image's cached crop windows per visit (the reference: a new random crop per visit)
            rows.append(crops[int(torch.randint(0, len(crops), (1,)))] if len(crops) > 1 else crops[0])
        tensors = [load_file(os.path.join(self.cache_folder, r["entry"])) for r in rows]
        out = {k: torch.stack([t[k] for t in tensors]) for k in tensors[0]}
        out.update({"caption": [r["caption"] for r in rows], "width": [r["width"] for r in rows], "height": [r["height"] for r in rows],
                    "original_size": torch.tensor([r["original_size"] for r in rows]),
                    "target_size": torch.tensor([r["target_size"] for r in rows]),
                    "crop_coords_top_left": torch.tensor([r["crop"] for r in rows])})
        return out


class CachedLatentDatasetConfig(AspectRatioBucketConfig):
    """YAML-selectable dataset over a cache written by `LatentCacheWriter` (same `batch_size` / bucket parameters as the
    `TextToImageDatasetConfig` it was built from are NOT required: entries carry their bucket)."""

    cache_folder: str
    num_repeats: int = 1

    def generate_buckets(self) -> list[CachedLatentBucket]:  # type: ignore[override]
        with open(os.path.join(self.cache_folder, INDEX_NAME)) as f:
            index = json.load(f)
        per_bucket: dict[tuple[int, int], dict[str, list[dict]]] = {}
        for e in index["entries"]:
            per_bucket.setdefault(tuple(e["bucket"]), {}).setdefault(e["file"], []).append(e)
        return [CachedLatentBucket(self.cache_folder, list(files.values()), self.batch_size, w, h, self.num_repeats)
                for (w, h), files in per_bucket.items()]

    def get_dataset(self) -> data.Dataset:
        return data.ConcatDataset([BucketDataset(b) for b in self.generate_buckets()])


def split_text_conditions(batch: dict) -> dict[str, torch.Tensor]:
    """The "text.<name>" entries of a cached batch as the dict a prompt encoder would have returned."""
    return {k[len(TEXT_PREFIX):]: v for k, v in batch.items() if k.startswith(TEXT_PREFIX)}
