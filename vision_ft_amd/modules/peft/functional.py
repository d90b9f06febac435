"""Adapter injection / extraction (reference: src/modules/peft/functional.py:16-360).

`replace_to_peft_layer` walks `named_children` recursively and swaps every targeted nn.Linear /
nn.Conv2d whose *full dotted name* is in the target list; `get_adapter_parameters` returns the
adapter-only state dict with the reference's key names.
"""
from __future__ import annotations

import warnings
from contextlib import contextmanager
from typing import Callable, NamedTuple

import torch
from torch import nn

from ...utils.state_dict import RegexMatch, get_target_keys
from ...utils.tensor import remove_orig_mod_prefix
from .config import PEFT_TYPE, PeftConfigMixin
from .lora import LoRAConfig, LoRAConv2d, LoRALinear
from .util import PeftLayer


def _make_adapter(layer: nn.Module, config: PeftConfigMixin) -> nn.Module:
    if config.type == "none":
        raise ValueError("peft type 'none' is not parameter efficient training")
    if config.type == "loha":
        if isinstance(layer, nn.Conv2d):  # the reference maps loha conv targets onto LoRAConv2d
            return LoRAConv2d(LoRAConfig.model_validate(config.model_dump() | {"type": "lora"}), layer)
        raise NotImplementedError("LoHa is outside the MI355X hot path (SURVEY.md §2 row 9)")
    if config.type != "lora":
        raise ValueError(f"Unknown peft type: {config.type}")
    cfg = LoRAConfig.model_validate(config.model_dump())
    return LoRAConv2d(cfg, layer) if isinstance(layer, nn.Conv2d) else LoRALinear(cfg, layer)


def _inject(parent: nn.Module, config: PeftConfigMixin, targets: set[str], prefix: str) -> None:
    for name, child in parent.named_children():
        path = prefix + name
        if isinstance(child, PeftLayer):
            continue
        if isinstance(child, (nn.Linear, nn.Conv2d)):
            if path in targets:
                setattr(parent, name, _make_adapter(child, config))
        else:
            _inject(child, config, targets, path + ".")


def replace_to_peft_layer(
    model: nn.Module,
    include_keys: list[str | RegexMatch],
    exclude_keys: list[str | RegexMatch],
    config: PeftConfigMixin,
    freeze_base: bool = False,
) -> None:
    targets = set(get_target_keys(include_keys, exclude_keys, [n for n, _ in model.named_modules()]))
    if freeze_base:
        for _, m in model.named_modules():
            m.requires_grad_(False)
    _inject(model, config, targets, "")


def get_adapter_parameters(model: nn.Module) -> dict[str, torch.Tensor]:
    out: dict[str, torch.Tensor] = {}
    for mod_name, module in model.named_modules():
        names = getattr(module, "adapter_param_names", None)
        if names is None:
            continue
        for key, value in module.state_dict().items():
            if any(key.startswith(n) for n in names):
                out[remove_orig_mod_prefix(f"{mod_name}.{key}")] = value
    return out


def extract_peft_layers(model: nn.Module) -> dict[str, PeftLayer]:
    return {n: m for n, m in model.named_modules() if isinstance(m, PeftLayer)}


def detect_peft_method(state_dict: dict[str, torch.Tensor]) -> PEFT_TYPE:
    return "lora" if any(k.endswith(".lora_up.weight") for k in state_dict) else "none"


def _load_into(parent: nn.Module, state_dict: dict[str, torch.Tensor], prefix: str) -> None:
    for name, child in parent.named_children():
        path = prefix + name
        weights = {w: state_dict.get(f"{path}.{w}") for w in LoRALinear.adapter_weight_names}
        complete = all(v is not None for k, v in weights.items() if "bias" not in k)
        if isinstance(child, PeftLayer):
            if complete:
                child.load_weights(weights)
        elif isinstance(child, nn.Linear):
            if complete:
                setattr(parent, name, LoRALinear.from_weights(weights, child))  # type: ignore[arg-type]
        else:
            _load_into(child, state_dict, path + ".")


def load_peft_weight(model: nn.Module, state_dict: dict[str, torch.Tensor]) -> None:
    if detect_peft_method(state_dict) == "none":
        raise ValueError("Failed to detect peft method from state_dict")
    _load_into(model, state_dict, "")


class TrainableParameters(NamedTuple):
    trainable_params: int
    all_param: int
    trainable_percent: float


def calculate_trainable_parameters(model: nn.Module) -> TrainableParameters:
    total = sum(p.numel() for p in model.parameters())
    train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return TrainableParameters(train, total, 100 * train / max(total, 1))


def _human(n: int) -> str:
    for unit, v in (("T", 10**12), ("B", 10**9), ("M", 10**6), ("K", 10**3)):
        if n >= v:
            return f"{n / v:.2f}{unit}"
    return str(n)


def print_trainable_parameters(model: nn.Module, print_fn: Callable = print) -> None:
    t, a, pct = calculate_trainable_parameters(model)
    print_fn(f"Trainable params: {_human(t)}, All params: {_human(a)}, Trainable%: {pct:.4f}%")
    if t == 0:
        warnings.warn("No trainable parameters found: check your peft config")


def set_peft_layer_enabled(model: nn.Module, enabled: bool) -> None:
    for _, m in model.named_modules():
        if hasattr(m, "set_enabled"):
            m.set_enabled(enabled)


@contextmanager
def while_peft_disabled(model: nn.Module):
    set_peft_layer_enabled(model, False)
    try:
        yield
    finally:
        set_peft_layer_enabled(model, True)


@contextmanager
def while_peft_enabled(model: nn.Module):
    set_peft_layer_enabled(model, True)
    try:
        yield
    finally:
        set_peft_layer_enabled(model, False)
