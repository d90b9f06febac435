"""NF4 linear layer consuming the bitsandbytes checkpoint layout — the MI355X handler of the
reference's ``"bnb_nf4"`` quant type (src/modules/quant/bnb.py:37-129 ``BnbLinear4bit``,
src/modules/quant/functional.py:47-58).

State-dict keys under ``<name>.`` (what ``quantize_4bit(...).as_dict(packed=True)`` emits and what the
reference sniffs for, functional.py:277-300):
    weight                                   uint8 (out*in/2, 1)  element 2i in the high nibble
    weight.absmax                            fp32 (nblocks,) | uint8 (nblocks,) when double-quantised
    weight.quant_map                         fp32 (16,)
    weight.nested_absmax / nested_quant_map  fp32 (nblocks/256,) / (256,)      [double quant]
    weight.quant_state.bitsandbytes__nf4     uint8 JSON {quant_type, blocksize, dtype, shape,
                                             nested_blocksize, nested_dtype, nested_offset}
    bias                                     optional

The forward/backward run ``vft_linear_nf4``: the packed codes are dequantised inside the GEMM launch (once per
launch into a transient scratch; no bf16 copy of the weight persists in HBM).  It *is* an ``nn.Linear`` subclass so LoRA targeting
(`isinstance(layer, nn.Linear)`, peft/functional.py:72) wraps it like the reference's bnb layer.
"""
from __future__ import annotations

import json
import os

import torch
import torch.nn as nn

from ... import kernels

# Default: the packed 4-bit codes are the ONLY copy of the weight in HBM and are dequantised inside the projection GEMM
# (vft_linear_nf4) — BASELINE config 4 as written ("fuses NF4->bf16 dequant into the down/up matmul in-kernel",
# reference: src/modules/quant/bnb.py:37-129 -> bnb.matmul_4bit).
# MI355X option VFT_NF4_CACHE=1: additionally keep a dequantised bf16 copy of every NF4 weight for COMPUTE (the packed codes
# stay the module's state and the checkpoint format).  With 288 GB per GPU the copy (2 + 2 bytes per parameter with the
# transposed one for the data gradient: ~10 GB for Lumina2) is affordable and the adapted layer then runs on the bf16
# projection GEMM with a bit-identical GEMM input (the same code x absmax products rounded to bf16) — a memory-for-speed
# trade that bench.py reports as its own, labelled leg, never as config 4.
NF4_COMPUTE_CACHE = os.environ.get("VFT_NF4_CACHE", "0") != "0"

_SUFFIXES = ("absmax", "quant_map", "nested_absmax", "nested_quant_map", "quant_state.bitsandbytes__nf4")


class NF4Linear(nn.Linear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, device=None):
        nn.Module.__init__(self)
        self.in_features, self.out_features = in_features, out_features
        self.blocksize = 64
        n = in_features * out_features
        if in_features % 64 != 0:
            raise ValueError("NF4Linear needs in_features % 64 == 0 (one quant block never straddles two rows)")
        nblocks = n // 64
        mk = lambda *s, dt: nn.Parameter(torch.zeros(*s, dtype=dt, device=device), requires_grad=False)  # noqa: E731
        self.weight = mk(n // 2, 1, dt=torch.uint8)
        self.register_buffer("absmax", torch.zeros(nblocks, dtype=torch.float32, device=device), persistent=False)
        self.register_buffer("quant_map", torch.zeros(16, dtype=torch.float32, device=device), persistent=False)
        self.register_buffer("nested_absmax", None, persistent=False)
        self.register_buffer("nested_quant_map", None, persistent=False)
        self.nested_offset = 0.0
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=torch.bfloat16, device=device), requires_grad=False) if bias else None
        self._state: kernels.NF4State | None = None
        self._dense: torch.Tensor | None = None

    # ---- state dict in the bitsandbytes layout -------------------------------------------------
    def _save_to_state_dict(self, destination, prefix, keep_vars):
        destination[prefix + "weight"] = self.weight if keep_vars else self.weight.detach()
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()
        destination[prefix + "weight.absmax"] = self.absmax
        destination[prefix + "weight.quant_map"] = self.quant_map
        meta = {"quant_type": "nf4", "blocksize": self.blocksize, "dtype": "bfloat16",
                "shape": [self.out_features, self.in_features]}
        if self.nested_absmax is not None:
            destination[prefix + "weight.nested_absmax"] = self.nested_absmax
            destination[prefix + "weight.nested_quant_map"] = self.nested_quant_map
            meta.update({"nested_blocksize": 256, "nested_dtype": "float32", "nested_offset": self.nested_offset})
        destination[prefix + "weight.quant_state.bitsandbytes__nf4"] = torch.tensor(list(json.dumps(meta).encode()), dtype=torch.uint8)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        qkey = prefix + "weight.quant_state.bitsandbytes__nf4"
        if qkey not in state_dict:
            error_msgs.append(f"{prefix}weight: NF4Linear expects a pre-quantised bitsandbytes NF4 weight ({qkey} missing); "
                              "quantise with vision_ft_amd.modules.quant.quantize_state_dict")
            return
        meta = json.loads(bytes(state_dict[qkey].tolist()).decode())
        if meta.get("quant_type") != "nf4" or meta.get("blocksize") != 64:
            error_msgs.append(f"{prefix}weight: only NF4 with blocksize 64 is supported, got {meta}")
            return
        if list(meta["shape"]) != [self.out_features, self.in_features]:
            error_msgs.append(f"{prefix}weight: shape {meta['shape']} != {[self.out_features, self.in_features]}")
            return
        dev = self.weight.device
        self.weight = nn.Parameter(state_dict[prefix + "weight"].to(dev).reshape(-1, 1).contiguous(), requires_grad=False)
        self.absmax = state_dict[prefix + "weight.absmax"].to(dev).contiguous()
        self.quant_map = state_dict[prefix + "weight.quant_map"].to(dev, torch.float32).contiguous()
        if prefix + "weight.nested_absmax" in state_dict:
            self.nested_absmax = state_dict[prefix + "weight.nested_absmax"].to(dev, torch.float32).contiguous()
            self.nested_quant_map = state_dict[prefix + "weight.nested_quant_map"].to(dev, torch.float32).contiguous()
            self.nested_offset = float(meta["nested_offset"])
        else:
            self.nested_absmax = self.nested_quant_map = None
            self.absmax = self.absmax.float()
        if self.bias is not None:
            if prefix + "bias" in state_dict:
                self.bias = nn.Parameter(state_dict[prefix + "bias"].to(dev, torch.bfloat16), requires_grad=False)
            elif strict:
                missing_keys.append(prefix + "bias")
        self._state = None
        self._dense = None
        for k in list(state_dict):
            if k.startswith(prefix) and k[len(prefix):] not in ("weight", "bias") and not any(k == prefix + "weight." + s for s in _SUFFIXES):
                unexpected_keys.append(k)

    def _apply(self, fn, recurse=True):
        # keep the integer payload intact under .to(dtype): only move devices
        self._state = None
        self._dense = None
        # probe on the payload's CURRENT device: a dtype-only conversion (.to(torch.bfloat16), .float()) then leaves
        # the device alone instead of dragging the packed weight to the CPU next to a bias that stayed on the GPU
        cur = self.weight.device
        dev = cur if cur.type == "meta" else fn(torch.zeros(1, device=cur)).device
        self.weight = nn.Parameter(self.weight.data.to(dev), requires_grad=False)
        for name in ("absmax", "quant_map", "nested_absmax", "nested_quant_map"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(dev))
        if self.bias is not None:
            self.bias = nn.Parameter(fn(self.bias.data), requires_grad=False)
        return self

    def quant_state(self) -> kernels.NF4State:
        if self._state is None:
            self._state = kernels.NF4State(self.weight.view(-1), self.absmax, self.nested_absmax, self.nested_quant_map,
                                           self.quant_map, self.nested_offset, self.out_features, self.in_features)
        return self._state

    def dequantize(self) -> torch.Tensor:
        return kernels.nf4_dequantize(self.quant_state())

    def dense_weight(self) -> torch.Tensor | None:
        """The cached bf16 copy (out, in) when NF4_COMPUTE_CACHE is on (built at first use, dropped by `.to()` / loading)."""
        if not NF4_COMPUTE_CACHE or self.weight.device.type != "cuda":
            return None
        if self._state is None or getattr(self, "_dense", None) is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            self._dense = self.dequantize().contiguous()
        return self._dense

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        w = self.dense_weight()
        if w is not None and kernels.linear_eligible(x, w):
            return kernels.linear(x, w, self.bias)
        return kernels.nf4_linear(x, self.quant_state(), self.bias)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bias={self.bias is not None}, quant=nf4/64"
