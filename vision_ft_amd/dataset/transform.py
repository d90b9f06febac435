"""Image transforms of the real-image front half (reference: src/dataset/transform.py:11-58 and the `v2.Compose`
in src/dataset/text_to_image.py:140-155).

torchvision is not a dependency here: its tensor resize with `antialias=True` *is*
`torch.nn.functional.interpolate(..., mode="bicubic", align_corners=False, antialias=True)` computed in fp32 and cast
back (torchvision/transforms/_functional_tensor.py `resize`), and `PILToTensor` / `ToDtype(scale=True)` / `Normalize`
are one-liners, so they are written out on plain tensors."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def pil_to_tensor(image) -> torch.Tensor:
    """PIL image -> uint8 (C,H,W) (v2.PILToTensor); palette / grey / alpha images are converted to RGB first, which is
    what the 3-channel Normalize of the reference requires."""
    if image.mode != "RGB":
        image = image.convert("RGB")
    return torch.from_numpy(np.asarray(image).copy()).permute(2, 0, 1).contiguous()


def to_normalised_half(image_u8: torch.Tensor) -> torch.Tensor:
    """uint8 0..255 -> fp16 -1..1: ToDtype(float16, scale=True) then Normalize(0.5, 0.5), both in fp16 as the
    reference runs them (text_to_image.py:142-148)."""
    x = image_u8.to(torch.float16) / 255.0
    return (x - 0.5) / 0.5


class ObjectCoverResize:
    """Resize so the image covers (width, height) keeping its aspect ratio (CSS `object-fit: cover`), bicubic with
    antialiasing; the random crop to the bucket size follows.  reference: transform.py:11-58."""

    def __init__(self, width: int, height: int, do_upscale: bool = False, antialias: bool = True):
        self.target_width = width
        self.target_height = height
        self.do_upscale = do_upscale
        self.antialias = antialias

    def scaled_size(self, w: int, h: int) -> tuple[int, int]:
        if (w < self.target_width or h < self.target_height) and not self.do_upscale:
            raise ValueError(f"Image is too small to crop to {self.target_width}x{self.target_height}")
        factor = max(self.target_width / w, self.target_height / h)
        return math.ceil(h * factor), math.ceil(w * factor)

    def __call__(self, img: torch.Tensor) -> torch.Tensor:
        h, w = img.shape[-2:]
        sh, sw = self.scaled_size(w, h)
        out = F.interpolate(img.unsqueeze(0).float(), size=[sh, sw], mode="bicubic", align_corners=False,
                            antialias=self.antialias)
        return out.squeeze(0).to(img.dtype)


def random_crop_params(image: torch.Tensor, output_size: tuple[int, int],
                       generator: torch.Generator | None = None) -> tuple[int, int, int, int]:
    """v2.RandomCrop.get_params: (top, left, height, width); two `torch.randint` draws from the global generator (or
    `generator`: the latent cache seeds one per file), top first — the order matters for seeded reproducibility."""
    h, w = image.shape[-2:]
    th, tw = output_size
    if h < th or w < tw:
        raise ValueError(f"Required crop size {(th, tw)} is larger than input image size {(h, w)}")
    if w == tw and h == th:
        return 0, 0, h, w
    top = int(torch.randint(0, h - th + 1, size=(1,), generator=generator).item())
    left = int(torch.randint(0, w - tw + 1, size=(1,), generator=generator).item())
    return top, left, th, tw
