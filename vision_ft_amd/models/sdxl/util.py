"""Checkpoint key naming for SDXL (reference: src/models/sdxl/util.py:6-227).

Original (sgm / ComfyUI) checkpoints index UNet stages as ``input_blocks.<i>.<j>``; the module tree here (like the
reference's) nests them as ``input_blocks.blocks.<i>.<j>``.  The VAE / text-encoder renames are kept as pure string
functions because ``convert_from_original_key`` is what classifies every checkpoint key (those modules themselves are
outside the hot path and are never instantiated)."""
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

# ---- VAE (ldm naming <-> diffusers AutoencoderKL naming): ordered first-occurrence substitutions around the block
#      re-indexing; (ldm, diffusers)
_VAE_HEAD = ((".attn_1.", ".attentions.0."), (".q.", ".to_q."), (".k.", ".to_k."), (".v.", ".to_v."), (".proj_out.", ".to_out.0."),
             (".norm.", ".group_norm."), (".nin_shortcut.", ".conv_shortcut."), (".mid.", ".mid_block."))
_VAE_TAIL = ((".upsample.conv.", ".upsamplers.0.conv."), (".downsample.conv.", ".downsamplers.0.conv."), (".block.", ".resnets."),
             (".norm_out.", ".conv_norm_out."))


def _vae_rename(key: str, forward: bool, num_blocks: int) -> str:
    src, dst = (0, 1) if forward else (1, 0)
    mid_marker = (".mid.", ".mid_block.")[src]
    if mid_marker in key:  # mid-block resnets are numbered from 1 in ldm checkpoints
        if forward:
            key = re.sub(r"block_(\d+)", lambda m: f"resnets.{int(m.group(1)) - 1}", key)
        else:
            key = re.sub(r"resnets\.(\d+)", lambda m: f"block_{int(m.group(1)) + 1}", key)
    for pair in _VAE_HEAD:
        key = key.replace(pair[src], pair[dst], 1)
    up, down = (("up", "up_blocks"), ("down", "down_blocks"))
    m_up = re.search(rf"\.{up[src]}\.(\d+)\.", key)
    m_down = re.search(rf"\.{down[src]}\.(\d+)\.", key)
    if m_up:  # decoder stages are stored in reverse order
        key = re.sub(rf"\.{up[src]}\.\d+\.", f".{up[dst]}.{num_blocks - 1 - int(m_up.group(1))}.", key)
    elif m_down:
        key = re.sub(rf"\.{down[src]}\.\d+\.", f".{down[dst]}.{int(m_down.group(1))}.", key)
    for pair in _VAE_TAIL:
        key = key.replace(pair[src], pair[dst], 1)
    return key


def vae_convert_from_original_key(key: str, num_blocks: int = 4) -> str:
    return _vae_rename(key, True, num_blocks)


def vae_convert_to_original_key(key: str, num_blocks: int = 4) -> str:
    return _vae_rename(key, False, num_blocks)


# ---- whole-checkpoint prefixes
_TE2_PROJ = ("conditioner.embedders.1.model.text_projection", "text_encoder.text_encoder_2.text_projection.weight")
_ROOT = (("conditioner.embedders.0.transformer.", "text_encoder.text_encoder_1."), _TE2_PROJ,
         ("conditioner.embedders.1.model.", "text_encoder.text_encoder_2.text_model."), ("first_stage_model.", "vae."))


def root_convert_from_original_key(key: str) -> str:
    key = key.replace("model.diffusion_model.", "diffusion_model.", 1).replace("diffusion_model.", "denoiser.", 1)
    for orig, ours in _ROOT:
        key = key.replace(orig, ours, 1)
    return key


def root_convert_to_original_key(key: str) -> str:
    key = key.replace("denoiser.", "model.diffusion_model.", 1)
    for orig, ours in _ROOT:
        key = key.replace(ours, orig, 1)
    return key


def convert_from_original_key(key: str) -> str:
    key = root_convert_from_original_key(key)
    if key.startswith("denoiser."):
        return denoiser_convert_from_original_key(key)
    if key.startswith("vae."):
        return vae_convert_from_original_key(key)
    return key


def convert_to_original_key(key: str) -> str:
    if key.startswith("denoiser."):
        key = denoiser_convert_to_original_key(key)
    elif key.startswith("vae."):
        key = vae_convert_to_original_key(key)
    return root_convert_to_original_key(key)


def convert_to_comfy_key(key: str) -> str:
    """Pipeline-level key (``denoiser.<...>``) -> ComfyUI LoRA/checkpoint key (``diffusion_model.<...>``)."""
    key = key.replace("text_encoder.text_encoder_1.", "clip_l.", 1)
    key = key.replace("text_encoder.text_encoder_2.", "clip_g.", 1)
    if key.startswith("denoiser."):
        key = denoiser_convert_to_original_key(key).replace("denoiser.", "diffusion_model.", 1)
    return key
