"""Checkpoint key naming for the SDXL denoiser (reference: src/models/sdxl/util.py:6-39, :212-227).

Original (sgm / ComfyUI) checkpoints index UNet stages as ``input_blocks.<i>.<j>``; the module tree
here (like the reference's) nests them as ``input_blocks.blocks.<i>.<j>``."""
from __future__ import annotations

import re

_TO_NESTED = re.compile(r"(input|output)_blocks\.")
_FROM_NESTED = re.compile(r"(input|output)_blocks\.blocks\.")


def unet_block_convert_from_original_key(key: str) -> str:
    key = _TO_NESTED.sub(lambda m: f"{m.group(1)}_blocks.blocks.", key)
    return key.replace("middle_block.", "middle_block.blocks.", 1)


def unet_block_convert_to_original_key(key: str) -> str:
    key = _FROM_NESTED.sub(lambda m: f"{m.group(1)}_blocks.", key)
    return key.replace("middle_block.blocks.", "middle_block.", 1)


denoiser_convert_from_original_key = unet_block_convert_from_original_key
denoiser_convert_to_original_key = unet_block_convert_to_original_key


def convert_to_comfy_key(key: str) -> str:
    """Pipeline-level key (``denoiser.<...>``) -> ComfyUI LoRA/checkpoint key (``diffusion_model.<...>``)."""
    key = key.replace("text_encoder.text_encoder_1.", "clip_l.", 1)
    key = key.replace("text_encoder.text_encoder_2.", "clip_g.", 1)
    if key.startswith("denoiser."):
        key = denoiser_convert_to_original_key(key).replace("denoiser.", "diffusion_model.", 1)
    return key
