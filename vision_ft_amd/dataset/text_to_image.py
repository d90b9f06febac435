"""Real-image text-to-image dataset: the front half of `train_step` (reference: src/dataset/text_to_image.py).

Images under `folder` are paired with `<stem>.txt` captions (or `<stem>.json` metadata carrying "caption" / "tags" /
"captions"), sorted into aspect-ratio buckets by `AspectRatioBucketManager.find_nearest`, and one dataset item is one
full batch of a single bucket: bicubic cover-resize to the bucket -> random crop -> the SDXL size conditions
(`original_size` = resized (h, w), `target_size` = crop (h, w), `crop_coords_top_left`), exactly the batch the
reference's `TextToImageBucket.__getitem__` (:182-230) hands to `train_step`.

Not carried over: the HF `datasets` cache (a plain list serves), `imagesize` (PIL reads the header) and the
booru-tag caption formatting of `ImageCaptionPair.read_caption` (:46-66, string work upstream of the text encoder,
which is out of scope)."""
from __future__ import annotations

import json
import os
import random
import warnings
from collections import defaultdict
from functools import reduce
from pathlib import Path
from typing import Callable, NamedTuple, Sequence

import torch
import torch.utils.data as data
from PIL import Image
from pydantic import BaseModel

from .aspect_ratio_bucket import AspectRatioBucket, AspectRatioBucketConfig, AspectRatioBucketManager
from .bucket import BucketDataset
from .transform import ObjectCoverResize, pil_to_tensor, random_crop_params, to_normalised_half


class ImageCaptionPair(BaseModel):
    image: Path
    width: int
    height: int
    caption: Path | None
    metadata: Path | None = None

    def read_caption(self) -> str:
        if self.metadata is not None:
            with open(self.metadata, "r") as f:
                metadata = json.load(f)
            if "tags" in metadata:
                return metadata["tags"]
            if "caption" in metadata:
                return metadata["caption"]
            if "captions" in metadata:
                return random.choice(metadata["captions"])
            if self.caption is None:
                raise ValueError(f"Caption not found in metadata {self.metadata}. Available keys: {', '.join(metadata.keys())}")
        assert self.caption is not None
        return self.caption.read_text()

    @property
    def should_skip(self) -> bool:
        if self.metadata is None:
            return False
        with open(self.metadata, "r") as f:
            metadata = json.load(f)
        return bool(metadata.get("skip", False))


class RandomCropOutput(NamedTuple):
    image: torch.Tensor
    top: int
    left: int
    crop_height: int
    crop_width: int
    original_height: int
    original_width: int


class TextToImageBucket(AspectRatioBucket):
    def __init__(self, items: list[ImageCaptionPair], batch_size: int, width: int, height: int, do_upscale: bool,
                 num_repeats: int, caption_processors: Sequence[Callable[[str], str]] = ()):
        rows = [{"image": str(p.image), "caption": p.read_caption(), "width": p.width, "height": p.height} for p in items]
        super().__init__(items=rows, batch_size=batch_size, num_repeats=num_repeats)
        self.cover_resize = ObjectCoverResize(width, height, do_upscale=do_upscale)
        self.width = width
        self.height = height
        self.do_upscale = do_upscale
        self.caption_processors = list(caption_processors)

    def resize_transform(self, image) -> torch.Tensor:
        # PILToTensor -> ToDtype(fp16, scale) -> Normalize(0.5, 0.5) -> ObjectCoverResize (reference :140-155)
        return self.cover_resize(to_normalised_half(pil_to_tensor(image)))

    def random_crop(self, image: torch.Tensor) -> RandomCropOutput:
        top, left, ch, cw = random_crop_params(image, (self.height, self.width))
        return RandomCropOutput(image=image[..., top:top + ch, left:left + cw], top=top, left=left, crop_height=ch,
                                crop_width=cw, original_height=image.shape[1], original_width=image.shape[2])

    def __getitem__(self, idx: int | slice):
        local = self.to_local_idx(idx)
        rows = [self.items[i] for i in local] if isinstance(local, list) else [self.items[local]]
        images, original_size, target_size, crop = [], [], [], []
        for row in rows:
            with Image.open(row["image"]) as pil:
                resized = self.resize_transform(pil)
            c = self.random_crop(resized)
            images.append(c.image)
            original_size.append(torch.tensor([c.original_height, c.original_width]))
            target_size.append(torch.tensor([c.crop_height, c.crop_width]))
            crop.append(torch.tensor([c.top, c.left]))
        captions = [reduce(lambda cap, proc: proc(cap), self.caption_processors, row["caption"]) for row in rows]
        return {"image": torch.stack(images), "caption": captions,
                "width": [row["width"] for row in rows], "height": [row["height"] for row in rows],
                "original_size": torch.stack(original_size), "target_size": torch.stack(target_size),
                "crop_coords_top_left": torch.stack(crop)}


class TextToImageDatasetConfig(AspectRatioBucketConfig):
    supported_extensions: list[str] = [".png", ".jpg", ".jpeg", ".webp", ".avif"]
    caption_extension: str = ".txt"
    metadata_extension: str = ".json"
    folder: str
    do_upscale: bool = False
    num_repeats: int = 1
    caption_processors: list = []

    def _retrive_images(self) -> list[ImageCaptionPair]:
        pairs: list[ImageCaptionPair] = []
        for root, _, files in os.walk(self.folder):
            for file in files:
                file = Path(file)
                if file.suffix not in self.supported_extensions:
                    continue
                image_path = Path(root) / file
                caption_path = Path(root) / (file.stem + self.caption_extension)
                metadata_path = Path(root) / (file.stem + self.metadata_extension)
                caption_path = caption_path if caption_path.exists() else None
                metadata_path = metadata_path if metadata_path.exists() else None
                if caption_path is None and metadata_path is None:
                    raise FileNotFoundError(f"Caption file or metadata file not found for image {image_path}")
                with Image.open(image_path) as im:
                    width, height = im.size  # header only
                pair = ImageCaptionPair(image=image_path, width=width, height=height, caption=caption_path, metadata=metadata_path)
                if not pair.should_skip:
                    pairs.append(pair)
        return pairs

    def generate_buckets(self) -> list[TextToImageBucket]:  # type: ignore[override]
        ar_buckets = self.buckets
        manager = AspectRatioBucketManager(ar_buckets)
        subsets: dict[int, list[ImageCaptionPair]] = defaultdict(list)
        for pair in self._retrive_images():
            try:
                subsets[manager.find_nearest(pair.width, pair.height)].append(pair)
            except Exception as e:  # too small for every bucket and do_upscale is off (reference :303-311)
                warnings.warn(f"Image size {pair.width}x{pair.height} is too small, and `do_upscale` is set False. "
                              f"Skipping... \n{e}", UserWarning)
        buckets = []
        for bucket_idx, pairs in subsets.items():
            if not pairs:
                continue
            width, height = (int(v) for v in ar_buckets[bucket_idx])
            buckets.append(TextToImageBucket(items=pairs, batch_size=self.batch_size, width=width, height=height,
                                             do_upscale=self.do_upscale, num_repeats=self.num_repeats,
                                             caption_processors=self.caption_processors))
        return buckets

    def get_dataset(self) -> data.Dataset:
        return data.ConcatDataset([BucketDataset(bucket) for bucket in self.generate_buckets()])
