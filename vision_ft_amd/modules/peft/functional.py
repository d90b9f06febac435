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
    install_adapter_pack(model)


def build_adapter_pack(model: nn.Module):
    """One `kernels.AdapterPack` for every bf16 LoRALinear of `model` that runs on the fused projection GEMM, plus one
    grouped entry per attention module whose q/k/v projections are fused (duck-typed: `to_q`, `to_k`, `to_v` LoRALinears
    and a `_fused_qkv_weight` method — SDXL SelfAttention).  Entries are attached to the modules (`_vft_pack`,
    `_vft_qkv_pack`), which pass them to the kernels."""
    from ... import kernels

    loras = [m for m in model.modules() if isinstance(m, LoRALinear)]
    dev = next((m.lora_down.weight.device for m in loras), None)
    if dev is None or dev.type != "cuda":
        return None
    pack = kernels.AdapterPack(dev)
    for m in loras:
        base = m.linear
        bf16_base = type(base) is nn.Linear and kernels.fused_lora_eligible(base.weight, m.rank)
        nf4_base = (hasattr(base, "quant_state") and m.rank in (4, 8, 16) and base.in_features % 64 == 0
                    and base.out_features % 64 == 0)  # the two cases kernels.lora_linear runs on a fused GEMM
        if ((bf16_base or nf4_base) and m.lora_up.bias is None and m.lora_down.weight.device == dev
                and kernels._adapters_ok((m.lora_down.weight, m.lora_up.weight))):
            m._vft_pack = pack.add([(m.lora_down.weight, m.lora_up.weight)], kernels._ext_rank(m.rank))
    for mod in model.modules():
        ps = [getattr(mod, n, None) for n in ("to_q", "to_k", "to_v")]
        if callable(getattr(mod, "_fused_qkv_weight", None)) and all(isinstance(p_, LoRALinear) and hasattr(p_, "_vft_pack") for p_ in ps):
            if len({p_.rank for p_ in ps}) == 1 and len({p_.lora_up.weight.shape for p_ in ps}) == 1 and 3 * ps[0].rank <= 16:
                mod._vft_qkv_pack = pack.add([(p_.lora_down.weight, p_.lora_up.weight) for p_ in ps], kernels._ext_rank(3 * ps[0].rank))
        kv = [getattr(mod, n, None) for n in ("to_k", "to_v")]
        if callable(getattr(mod, "_fused_kv_weight", None)) and all(isinstance(p_, LoRALinear) and hasattr(p_, "_vft_pack") for p_ in kv):
            if kv[0].rank == kv[1].rank and kv[0].lora_up.weight.shape == kv[1].lora_up.weight.shape and 2 * kv[0].rank <= 16:
                mod._vft_kv_pack = pack.add([(p_.lora_down.weight, p_.lora_up.weight) for p_ in kv], kernels._ext_rank(2 * kv[0].rank))
    if not pack.entries:
        return None
    pack.finalize()
    return pack


def _pack_pre_forward(model: nn.Module, args, kwargs=None):
    state = model.__dict__.setdefault("_vft_pack_state", {"pack": None, "tried": False})
    if state["pack"] is None and not state["tried"]:
        if torch.cuda.is_available() and not torch.cuda.is_current_stream_capturing():
            state["pack"] = build_adapter_pack(model)
            state["tried"] = state["pack"] is not None or any(
                isinstance(m, LoRALinear) and m.lora_down.weight.is_cuda for m in model.modules())
    if state["pack"] is not None:
        state["pack"].refresh()


def install_adapter_pack(model: nn.Module) -> None:
    """Refresh the adapter operand pack at the start of every forward of the smallest submodule that contains all LoRA
    layers (the denoiser: the trainers call it directly — also from inside captured HIP graphs — not through the
    task object that `replace_to_peft_layer` receives).  The pack is built lazily at the first CUDA forward."""
    names = [n.split(".") for n, m in model.named_modules() if isinstance(m, LoRALinear)]
    if not names:
        return
    common = names[0][:-1]
    for parts in names[1:]:
        k = 0
        while k < len(common) and k < len(parts) - 1 and common[k] == parts[k]:
            k += 1
        common = common[:k]
    root = model.get_submodule(".".join(common)) if common else model
    root.__dict__["_vft_pack_state"] = {"pack": None, "tried": False}  # adapters changed: rebuild
    if not root.__dict__.get("_vft_pack_hooked", False):
        root.register_forward_pre_hook(_pack_pre_forward, with_kwargs=True)
        root.__dict__["_vft_pack_hooked"] = True


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
    install_adapter_pack(model)  # parameters were replaced


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
