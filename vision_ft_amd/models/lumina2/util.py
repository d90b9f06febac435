"""Checkpoint key naming for Lumina-Image-2.0 (reference: src/models/lumina2/util.py:5-168).  The denoiser keeps the
original names; only the top-level prefixes differ (ComfyUI-style single file: ``model.diffusion_model.``,
``text_encoders.gemma2_2b.transformer.``, ``vae.``)."""
from __future__ import annotations

from ..sdxl.util import vae_convert_from_original_key, vae_convert_to_original_key

_TE = ("text_encoders.gemma2_2b.transformer.", "text_encoder.")


def root_convert_from_original_key(key: str) -> str:
    key = key.replace("model.diffusion_model.", "diffusion_model.", 1).replace("diffusion_model.", "denoiser.", 1)
    return key.replace(_TE[0], _TE[1], 1)


def root_convert_to_original_key(key: str) -> str:
    return key.replace("denoiser.", "model.diffusion_model.", 1).replace(_TE[1], _TE[0], 1)


def denoiser_convert_from_original_key(key: str) -> str:
    return key


def denoiser_convert_to_original_key(key: str) -> str:
    return key


def convert_from_original_key(key: str) -> str:
    key = root_convert_from_original_key(key)
    return vae_convert_from_original_key(key) if key.startswith("vae.") else key


def convert_to_original_key(key: str) -> str:
    # (the reference renames the root first, so its `startswith("denoiser.")` branch never fires; a no-op either way)
    key = root_convert_to_original_key(key)
    return vae_convert_to_original_key(key) if key.startswith("vae.") else key


def convert_to_comfy_key(key: str) -> str:
    return convert_to_original_key(key).replace("model.diffusion_model.", "diffusion_model.", 1)
