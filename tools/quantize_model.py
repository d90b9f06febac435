#!/usr/bin/env python
"""Quantise the denoiser Linears of a checkpoint to bitsandbytes-layout NF4 and save it — counterpart of the
reference's tools/quantize_model.py:16-58 (which loads the model, swaps the target Linears for bnb 4-bit layers and
lets bitsandbytes quantise them on `.cuda()`).  Here the weights are quantised directly in the state dict
(`quantize_state_dict`: blocksize 64, absmax fp32 or double-quantised) and written with the same key layout
(`<name>.weight`, `.absmax`, `.quant_map`, `.quant_state.bitsandbytes__nf4`, optional `.nested_*`), so the result
loads through `_from_checkpoint` (key sniffing -> NF4Linear) here and through the reference.

  python tools/quantize_model.py --family auraflow --model-path in.safetensors --save-path out.bnb_nf4.safetensors
"""
from __future__ import annotations

import os
import sys

import click
from safetensors.torch import load_file, save_file

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vision_ft_amd.modules.quant.functional import quantize_state_dict, validate_quant_type  # noqa: E402
from vision_ft_amd.utils.state_dict import get_target_keys  # noqa: E402

FAMILIES = {
    # family: (key converter module, default include, default exclude) — AuraFlow's defaults are the reference tool's
    "auraflow": ("vision_ft_amd.models.auraflow.pipeline", ["denoiser."], ["t_embedder", "final_linear", "modF"]),
    "sdxl": ("vision_ft_amd.models.sdxl.util", ["attn1", "attn2", ".ff."], ["time_embed", "label_emb"]),
    "lumina2": ("vision_ft_amd.models.lumina2.util", ["attention", "feed_forward"], ["noise_refiner", "t_embedder", "final_layer"]),
}


def quantize_checkpoint(state_dict: dict, family: str, quant_type: str = "bnb_nf4", include_keys: list | None = None,
                        exclude_keys: list | None = None, double_quant: bool = True) -> dict:
    """Returns the checkpoint dict (original key names kept) with the selected 2-D `*.weight` tensors replaced by their
    NF4 payload + quantisation keys.  Selection works on the converted names (`denoiser.<module path>.weight`), like
    the reference's include/exclude keys."""
    import importlib

    validate_quant_type(quant_type)
    mod_name, inc, exc = FAMILIES[family]
    conv = importlib.import_module(mod_name).convert_from_original_key
    include_keys = inc if include_keys is None else include_keys
    exclude_keys = exc if exclude_keys is None else exclude_keys
    by_conv = {conv(k): k for k in state_dict}
    cands = [c for c in by_conv if c.startswith("denoiser.") and c.endswith(".weight") and state_dict[by_conv[c]].dim() == 2
             and state_dict[by_conv[c]].shape[1] % 64 == 0]
    targets = get_target_keys(include_keys, exclude_keys, cands)
    orig_targets = [by_conv[c] for c in targets]
    return quantize_state_dict(state_dict, quant_type, include_keys=orig_targets, double_quant=double_quant), orig_targets


@click.command()
@click.option("--family", type=click.Choice(sorted(FAMILIES)), required=True)
@click.option("--model-path", type=str, required=True)
@click.option("--save-path", type=str, required=True)
@click.option("--quant-type", type=str, default="bnb_nf4")
@click.option("--include-keys", type=str, multiple=True)
@click.option("--exclude-keys", type=str, multiple=True)
@click.option("--double-quant/--no-double-quant", default=True)
def main(family, model_path, save_path, quant_type, include_keys, exclude_keys, double_quant):
    print("Loading model from", model_path)
    sd = load_file(model_path)
    sd, targets = quantize_checkpoint(sd, family, quant_type, list(include_keys) or None, list(exclude_keys) or None, double_quant)
    print(f"Quantised {len(targets)} weights to {quant_type}")
    print("Saving model to", save_path)
    save_file({k: v.contiguous() for k, v in sd.items()}, save_path)
    print("Done!")


if __name__ == "__main__":
    main()
