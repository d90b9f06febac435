"""Task protocol (reference: src/models/for_training.py:14-216): every training task subclasses
``ModelForTraining`` and implements ``setup_model`` / ``train_step`` / ...; the trainer drives the
hooks.  ``accelerator`` is replaced by the lightweight ``Runtime`` below (one process per GPU,
torch.distributed over RCCL) — same attribute names where the tasks use them."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import torch
import torch.distributed as dist
import torch.nn as nn
from pydantic import BaseModel

from ..config import TrainConfig


class Runtime:
    """What the tasks need from `accelerate.Accelerator`: device, rank info, print, gather."""

    def __init__(self) -> None:
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.sync_gradients = True

    @property
    def is_main_process(self) -> bool:
        return self.rank == 0

    def print(self, *a, **k) -> None:
        if self.is_main_process:
            print(*a, **k)

    def gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return t.reshape(1) if t.dim() == 0 else t
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t)
        return torch.stack(out)

    def wait_for_everyone(self) -> None:
        if self.world > 1:
            dist.barrier()


class ModelForTraining(ABC, nn.Module):
    model_config_class: type[BaseModel]
    model: nn.Module

    def __init__(self, accelerator: Runtime, config: TrainConfig, *args, **kwargs) -> None:
        super().__init__()
        self.config = config
        self.accelerator = accelerator
        self._current_step = 0
        self._logs_at_step: dict = {}
        self._logs_at_epoch: dict[str, list] = {}
        self._is_peft = False
        self.log_every = 1  # device->host transfers for logging happen every N steps (MI355X: keep the queue full)
        self.validate_config()

    def validate_config(self) -> None:
        self.model_config = self.model_config_class.model_validate(self.config.model)

    def _set_is_peft(self, is_peft: bool) -> None:
        self._is_peft = is_peft

    def load_peft_weights(self) -> None:
        pass

    def before_setup_model(self) -> None:
        pass

    @abstractmethod
    def setup_model(self) -> None: ...

    def after_setup_model(self) -> None:
        self.accelerator.wait_for_everyone()

    @abstractmethod
    def sanity_check(self) -> None: ...

    @abstractmethod
    def train_step(self, batch) -> torch.Tensor: ...

    def eval_step(self, batch) -> torch.Tensor:
        raise NotImplementedError

    def before_train_step(self) -> None:
        self._current_step += 1

    def after_train_step(self) -> None:
        self._logs_at_step = {}

    def before_backward(self) -> None:
        pass

    def after_backward(self) -> None:
        cfg = self.config.trainer
        params = [p for p in self.model.parameters() if p.requires_grad]
        if cfg.clip_grad_norm is not None:
            flat = getattr(self, "_flat_grad_buffers", None)
            if flat and all(b.is_cuda and b.dtype == torch.bfloat16 for b in flat):
                # every trainable gradient is a view into these buffers (trainer/dp.py; the Trainer hands them over): one
                # pass of the own sum-of-squares kernel per buffer instead of a multi-tensor pass over every view
                from .. import kernels

                kernels.clip_flat_grad_norm_(list(flat), cfg.clip_grad_norm)
            else:
                torch.nn.utils.clip_grad_norm_(params, cfg.clip_grad_norm)
        if cfg.clip_grad_value is not None:
            torch.nn.utils.clip_grad_value_(params, cfg.clip_grad_value)

    def before_train_epoch(self) -> None:
        self.model.train()

    def after_train_epoch(self) -> None:
        self.model.eval()

    def get_state_dict_to_save(self) -> dict[str, torch.Tensor]:
        return self.model.state_dict()

    def get_metadata_to_save(self) -> dict[str, str]:
        return {}

    def preview_step(self, batch, preview_index: int) -> Any:
        raise NotImplementedError("preview generation is outside the MI355X hot path")

    def print(self, *args, **kwargs) -> None:
        self.accelerator.print(*args, **kwargs)

    def log(self, name: str, value, on_step: bool = True, on_epoch: bool = False) -> None:
        """Tensors are kept on the device; they are only reduced/transferred when read
        (``logs_as_floats``), so logging never stalls the launch queue (the reference all-gathers and
        `.item()`s on every call, for_training.py:171-189)."""
        if isinstance(value, torch.Tensor):
            value = value.detach()
        if on_step:
            self._logs_at_step[name] = value
        if on_epoch:
            self._logs_at_epoch.setdefault(name, []).append(value)

    def logs_as_floats(self) -> dict[str, float]:
        out = {}
        for k, v in self._logs_at_step.items():
            if isinstance(v, torch.Tensor):
                v = self.accelerator.gather(v.float()).mean().item()
            out[k] = v
        return out

    def forward(self, *args, **kwargs):
        return self.train_step(*args, **kwargs)
