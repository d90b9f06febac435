"""Checkpoint cadence + safetensors callback (reference: src/saving/util.py, src/saving/safetensors.py).
File name template and cadence semantics are the reference's; the HF-hub callback is out of scope."""
from __future__ import annotations

from pathlib import Path

import torch
from pydantic import BaseModel
from safetensors.torch import save_file


class ModelSavingStrategyConfig(BaseModel):
    per_epochs: int | float | None = 1
    per_steps: int | None = None
    save_last: bool = True


class ModelSavingStrategy:
    def __init__(self, total_epochs: int, steps_per_epoch: int, per_epochs: int | float | None, per_steps: int | None,
                 save_last: bool):
        self.per_epochs, self.per_steps, self.save_last = per_epochs, per_steps, save_last
        self._total_epochs, self._steps_per_epoch = total_epochs, steps_per_epoch
        self.sanity_check()

    @classmethod
    def from_config(cls, config: ModelSavingStrategyConfig, total_epochs: int, steps_per_epoch: int) -> "ModelSavingStrategy":
        return cls(total_epochs=total_epochs, steps_per_epoch=steps_per_epoch, **config.model_dump())

    def sanity_check(self) -> None:
        pe, ps = self.per_epochs, self.per_steps
        if pe is not None:
            if pe <= 0:
                raise ValueError("per_epochs must be greater than 0")
            if isinstance(pe, float):
                if pe >= 1:
                    raise ValueError("per_epochs must be less than 1 if float")
                if ps is not None:
                    raise ValueError("per_epochs and per_steps cannot be set together")
            elif pe > self._total_epochs:
                raise ValueError("per_epochs must be less than or equal to total_epochs")
        if ps is not None:
            if ps <= 0:
                raise ValueError("per_steps must be greater than 0")
            if ps > self._total_epochs * self._steps_per_epoch:
                raise ValueError("per_steps must be less than or equal to total_steps")

    @property
    def _per_steps(self) -> int | None:
        if isinstance(self.per_epochs, float):
            return int(self.per_epochs * self._steps_per_epoch)
        return self.per_steps

    def should_save(self, epoch: int, steps: int) -> bool:
        if self._steps_per_epoch is None and self._total_epochs is None:
            return False
        if epoch == 0 and steps == 0:
            return False
        if isinstance(self.per_epochs, int) and epoch != 0 and steps % (self._steps_per_epoch * self.per_epochs) == 0:
            return True
        if self._per_steps is not None and steps != 0 and steps % self._per_steps == 0:
            return True
        return False


class SafetensorsSavingCallbackConfig(BaseModel):
    type: str = "safetensors"
    name: str
    save_dir: str | Path


class SafetensorsSavingCallback:
    save_name_template = "{name}_{epoch:05}e_{steps:06}s.safetensors"

    def __init__(self, name: str, save_dir: str | Path):
        self.name, self.save_dir = name, Path(save_dir)

    @classmethod
    def from_config(cls, config: SafetensorsSavingCallbackConfig, **kwargs) -> "SafetensorsSavingCallback":
        return cls(name=config.name, save_dir=config.save_dir)

    def save_state_dict(self, state_dict: dict[str, torch.Tensor], epoch: int, steps: int, metadata: dict | None = None) -> Path:
        path = self.save_dir / self.save_name_template.format(name=self.name, epoch=epoch, steps=steps)
        path.parent.mkdir(parents=True, exist_ok=True)
        save_file({k: v.detach().contiguous().cpu() for k, v in state_dict.items()}, path, metadata)
        return path


ModelSavingCallbackConfgiAlias = SafetensorsSavingCallbackConfig  # (sic)


def get_saving_callback(config: SafetensorsSavingCallbackConfig, **kwargs) -> SafetensorsSavingCallback:
    return SafetensorsSavingCallback.from_config(config, **kwargs)
