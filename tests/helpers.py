"""Shared test helpers: golden-fixture loading and synthetic weights."""
from __future__ import annotations

import json
import os

import torch
from safetensors import safe_open

from oracle import fixture_weights as fw

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str) -> tuple[dict[str, torch.Tensor], dict[str, str]]:
    path = os.path.join(GOLDEN, name)
    with safe_open(path, "pt") as f:
        tensors = {k: f.get_tensor(k) for k in f.keys()}
        meta = f.metadata() or {}
    return tensors, meta


def load_index() -> dict:
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


def module_shapes(module: torch.nn.Module) -> dict[str, tuple]:
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def synth_params(module_on_meta: torch.nn.Module, seed: int, expect_checksum: str | None = None) -> dict[str, torch.Tensor]:
    """fp32 synthetic state dict for a module built on the meta device; verifies the checksum that
    the golden generator recorded for the *reference* module (=> identical names and shapes)."""
    sd = fw.synth_state_dict(module_shapes(module_on_meta), seed)
    if expect_checksum is not None:
        got = fw.checksum(sd)
        # (relative 1e-12: the fp64 sum over the larger fixtures differs in its last digit between hosts — thread count of the
        # reduction — 461389.18600037094 vs ...3709 on the GPU box; a wrong name, shape or value moves it by far more)
        assert abs(got - float(expect_checksum)) <= 1e-12 * abs(float(expect_checksum)), f"state-dict mismatch vs reference: {got!r} != {expect_checksum}"
    return sd


def autocast_cpu():
    return torch.autocast("cpu", dtype=torch.bfloat16)


def add_lora_params(P: dict[str, torch.Tensor], lora_keys: list[str], seed: int = 11, dtype=torch.bfloat16, rank: int = 4) -> dict[str, torch.Tensor]:
    """Rewrite a flat base state dict into the reference's LoRA-wrapped layout for `lora_keys`
    (the sorted adapter key list stored in the fixture): `<n>.weight` -> `<n>.linear.weight`, plus
    synthetic `<n>.lora_down.weight` (bf16), zero `<n>.lora_up.weight`, `<n>.alpha` = 1."""
    P = dict(P)
    wrapped = sorted({k.rsplit(".lora_", 1)[0] for k in lora_keys if ".lora_" in k})
    shapes = {}
    for n in wrapped:
        w = P.pop(f"{n}.weight")
        P[f"{n}.linear.weight"] = w
        if f"{n}.bias" in P:
            P[f"{n}.linear.bias"] = P.pop(f"{n}.bias")
        shapes[f"{n}.lora_down.weight"] = (rank, w.shape[1])
        shapes[f"{n}.lora_up.weight"] = (w.shape[0], rank)
    synth = fw.synth_state_dict(shapes, seed=seed)
    for k, v in synth.items():
        P[k] = (torch.zeros_like(v) if "lora_up" in k else v).to(dtype)
    for n in wrapped:
        P[f"{n}.alpha"] = torch.tensor(1.0, dtype=dtype)
    return P


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()
