"""Checkpoint loading for the three denoiser families: original key names -> this module tree, pre-quantised (NF4)
weights recognised by their extra keys, strict load with ``assign=True``.

Counterpart of the ``_from_checkpoint`` / ``_load_original_weights`` methods of the reference pipelines
(src/models/sdxl/pipeline.py:49-102, src/models/lumina2/pipeline.py:47-86, src/models/auraflow/pipeline.py:85-136).
Only the denoiser is instantiated here — the VAE and the text encoders are outside the hot path
— so converted keys under ``vae.`` / ``text_encoder.`` are dropped instead of loaded."""
from __future__ import annotations

from typing import Callable

import torch
import torch.nn as nn
from safetensors.torch import load_file

from ..modules.quant.functional import replace_by_prequantized_weights


def denoiser_state_dict(state_dict: dict[str, torch.Tensor], convert_key: Callable[[str], str]) -> dict[str, torch.Tensor]:
    """Original checkpoint dict -> ``{key without 'denoiser.': tensor}`` for the keys that land under ``denoiser.``."""
    out = {}
    for key, value in state_dict.items():
        new = convert_key(key)
        if new.startswith("denoiser."):
            out[new[len("denoiser."):]] = value
    return out


def load_denoiser_checkpoint(holder: nn.Module, path: str, convert_key: Callable[[str], str], strict: bool = True) -> None:
    """`holder.denoiser` <- checkpoint at `path`.  Same three steps as the reference: rename, swap the Linears whose
    entries carry quantisation keys (``<name>.weight.quant_state.bitsandbytes__nf4`` ...), ``load_state_dict(assign=True)``."""
    sd = denoiser_state_dict(load_file(path), convert_key)
    if not sd:
        raise ValueError(f"{path}: no denoiser tensors found after key conversion")
    # the sniffing walks module paths as they are in the holder ("denoiser.<...>"), like the reference's call on the pipeline
    replace_by_prequantized_weights(holder, {"denoiser." + k: v for k, v in sd.items()})
    holder.denoiser.load_state_dict(sd, strict=strict, assign=True)
