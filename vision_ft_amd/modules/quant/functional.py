"""Quantised-linear plumbing (reference: src/modules/quant/functional.py:21-371): quant-type
literal, checkpoint-key sniffing, module replacement, and NF4 quantisation of state dicts.

Only ``"bnb_nf4"`` has an MI355X handler (BASELINE config 4); the other literals are accepted by
``validate_quant_type`` for schema compatibility and rejected by the factories."""
from __future__ import annotations

import json
from typing import Literal

import torch
import torch.nn as nn

from ...utils.state_dict import get_target_keys
from .nf4 import NF4Linear

QUANT_TYPE = Literal["fp8_e4m3fn", "bnb_int8", "bnb_fp4", "bnb_nf4", "quanto_int4", "quanto_int8", "ao_nf4", "ao_fp8"]
_ALL = ("fp8_e4m3fn", "bnb_int8", "bnb_fp4", "bnb_nf4", "quanto_int4", "quanto_int8", "ao_nf4", "ao_fp8")

# QLoRA / bitsandbytes NF4 code book
NF4_CODE = torch.tensor([-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635,
                         -0.18477343022823334, -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725,
                         0.24611230194568634, 0.33791524171829224, 0.44070982933044434, 0.5626170039176941,
                         0.7229568362236023, 1.0], dtype=torch.float32)


def validate_quant_type(quant_type: str) -> None:
    if quant_type not in _ALL:
        raise ValueError(f"Unknown quant_type: {quant_type}")


def _get_quant_linear(module: nn.Linear, quant_type: QUANT_TYPE) -> nn.Module:
    validate_quant_type(quant_type)
    if quant_type == "bnb_nf4":
        return NF4Linear(module.in_features, module.out_features, bias=module.bias is not None, device=module.weight.device
                         if module.weight.device.type != "meta" else None)
    raise NotImplementedError(f"quant_type {quant_type!r} has no MI355X kernel (only 'bnb_nf4' is on the hot path)")


def collect_children_dict(prefix: str, state_dict: dict[str, torch.Tensor], remove_prefix: bool = True) -> dict[str, torch.Tensor]:
    return {(k[len(prefix):] if remove_prefix else k): v for k, v in state_dict.items() if k.startswith(prefix)}


def get_quant_type_from_children_dict(children_dict: dict[str, torch.Tensor]) -> QUANT_TYPE:
    for key, tensor in children_dict.items():
        if "quant_state" in key:
            kind = key[len("quant_state.bitsandbytes__"):]
            if kind == "nf4":
                return "bnb_nf4"
            if kind == "fp4":
                return "bnb_fp4"
        elif "weight_format" in key:
            return "bnb_int8"
        elif "_data" in key:
            if tensor.dtype == torch.int8:
                return "quanto_int8"
            if tensor.dtype == torch.uint8:
                return "quanto_int4"
    raise ValueError("quant_type not found")


def replace_by_prequantized_weights(model: nn.Module, state_dict: dict[str, torch.Tensor], prefix: str = "") -> None:
    """Swap every nn.Linear whose checkpoint entry carries quantisation keys (``<name>.weight.<...>``)."""
    for name, layer in model.named_children():
        path = prefix + name
        if isinstance(layer, nn.Linear):
            children = collect_children_dict(f"{path}.weight.", state_dict)
            if children:
                q = _get_quant_linear(layer, get_quant_type_from_children_dict(children))
                q.requires_grad_(False)
                setattr(model, name, q)
        else:
            replace_by_prequantized_weights(layer, state_dict, path + ".")


def replace_to_quant_linear(model: nn.Module, quant_type: QUANT_TYPE, include_keys: list, exclude_keys: list = []) -> None:
    targets = set(get_target_keys(include_keys, exclude_keys, [n for n, _ in model.named_modules()]))

    def walk(parent: nn.Module, prefix: str) -> None:
        for name, layer in parent.named_children():
            path = prefix + name
            if isinstance(layer, nn.Linear) and not isinstance(layer, NF4Linear):
                if path in targets:
                    setattr(parent, name, _get_quant_linear(layer, quant_type))
            else:
                walk(layer, path + ".")

    walk(model, "")


def _dynamic_map_8bit() -> torch.Tensor:
    """bitsandbytes create_dynamic_map(signed=True, max_exponent_bits=7, total_bits=8)."""
    data: list[float] = []
    for i in range(7):
        n = int(2 ** i + 1)
        b = torch.linspace(0.1, 1, n)
        means = (b[:-1] + b[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data += [0.0, 1.0]
    data += [0.0] * (256 - len(data))
    return torch.tensor(sorted(data), dtype=torch.float32)


def _nearest(x: torch.Tensor, code: torch.Tensor) -> torch.Tensor:
    mid = (code[:-1].double() + code[1:].double()) / 2
    return torch.bucketize(x.double(), mid.to(x.device), right=False).to(torch.uint8)


def quantize_nf4(weight: torch.Tensor, double_quant: bool = True) -> dict[str, torch.Tensor]:
    """bitsandbytes-compatible NF4 quantisation of a 2-D weight (blocksize 64); returns the tensors keyed
    by the suffix after ``<name>.weight`` ('' = packed payload).  Offline tooling, any device."""
    out_f, in_f = weight.shape
    flat = weight.detach().float().reshape(-1, 64)
    absmax = flat.abs().amax(dim=1)
    safe = torch.where(absmax == 0, torch.ones_like(absmax), absmax)
    q = _nearest(flat / safe[:, None], NF4_CODE.to(flat.device)).reshape(-1)
    packed = ((q[0::2] << 4) | q[1::2]).to(torch.uint8).reshape(-1, 1)
    out = {"": packed, ".quant_map": NF4_CODE.clone().to(weight.device)}
    meta = {"quant_type": "nf4", "blocksize": 64, "dtype": "bfloat16", "shape": [out_f, in_f]}
    if double_quant:
        offset = absmax.mean()
        resid = absmax - offset
        nb = resid.numel()
        pad = (-nb) % 256
        blocks = torch.nn.functional.pad(resid, (0, pad)).reshape(-1, 256)
        am2 = blocks.abs().amax(dim=1)
        safe2 = torch.where(am2 == 0, torch.ones_like(am2), am2)
        code2 = _dynamic_map_8bit().to(weight.device)
        out[".absmax"] = _nearest(blocks / safe2[:, None], code2).reshape(-1)[:nb]
        out[".nested_absmax"] = am2.float()
        out[".nested_quant_map"] = code2
        meta.update({"nested_blocksize": 256, "nested_dtype": "float32", "nested_offset": float(offset)})
    else:
        out[".absmax"] = absmax.float()
    out[".quant_state.bitsandbytes__nf4"] = torch.tensor(list(json.dumps(meta).encode()), dtype=torch.uint8)
    return out


def quantize_state_dict(state_dict: dict[str, torch.Tensor], quant_type: QUANT_TYPE, include_keys: list,
                        exclude_keys: list = [], double_quant: bool = False) -> dict[str, torch.Tensor]:
    """NF4-quantise the matching 2-D weights in place (reference: functional.py:342-371, which calls
    bitsandbytes' ``quantize_4bit`` with its default ``compress_statistics=False``)."""
    if quant_type != "bnb_nf4":
        raise NotImplementedError("only 'bnb_nf4' quantisation is implemented for MI355X")
    for key in get_target_keys(include_keys, exclude_keys, list(state_dict.keys())):
        if key not in state_dict or state_dict[key].dim() != 2:
            continue
        for suffix, t in quantize_nf4(state_dict[key], double_quant=double_quant).items():
            state_dict[key + suffix] = t.cpu()
    return state_dict
