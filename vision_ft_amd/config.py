"""Training configuration: the YAML schema of the reference (src/config.py:21-115), validated with
pydantic in strict mode; ``model`` and ``dataset`` stay raw dicts that the task re-validates with its
own config class (src/models/for_training.py:41-42, src/trainer/common.py:78-81).  Unknown keys are
ignored, as in the reference.  Preview generation is outside the hot path: a ``preview`` section is
accepted and ignored."""
from __future__ import annotations

from pathlib import Path
from typing import Literal

import yaml
from pydantic import BaseModel

from .modules.peft import PeftTargetConfig
from .saving import ModelSavingCallbackConfgiAlias, ModelSavingStrategyConfig, SafetensorsSavingCallbackConfig


class OptimizerConfig(BaseModel):
    name: str = "torch.optim.AdamW"
    args: dict = {"lr": 1e-3}


class SchedulerConfig(BaseModel):
    name: str = "torch.optim.lr_scheduler.ConstantLR"
    args: dict = {}


class SavingConfig(BaseModel):
    strategy: ModelSavingStrategyConfig = ModelSavingStrategyConfig()
    callbacks: list[ModelSavingCallbackConfgiAlias] = [SafetensorsSavingCallbackConfig(name="model", save_dir="./output")]
    rename_key_map: dict[str, str] = {}


class TrackerConfig(BaseModel):
    project_name: str
    loggers: list[Literal["wandb", "tensorboard"]]


DEBUG_MODE_TYPE = Literal[False, "sanity_check", "1step", "dataset"]


class TrainerConfig(BaseModel):
    debug_mode: DEBUG_MODE_TYPE = False
    torch_compile: bool = False  # accepted for schema compatibility; the MI355X path uses HIP graphs instead
    torch_compile_args: dict = {}
    gradient_checkpointing: bool = False
    gradient_accumulation_steps: int = 1
    clip_grad_norm: float | None = None
    clip_grad_value: float | None = None
    fp32_matmul_precision: Literal["highest", "high", "medium"] | None = None
    allow_tf32: bool = False
    # MI355X extension: capture each bucket shape's step into HIP graphs (vision_ft_amd/trainer/graph.py)
    hip_graph: bool = True
    # MI355X extension: pick library GEMM solutions from the committed gfx950 table (vision_ft_amd/tuning.py)
    tuned_gemms: bool = True


class TrainConfig(BaseModel):
    model: dict | BaseModel
    dataset: dict | BaseModel
    peft: PeftTargetConfig | list[PeftTargetConfig] | None = None
    optimizer: OptimizerConfig = OptimizerConfig()
    scheduler: SchedulerConfig | None = None
    saving: SavingConfig | None = SavingConfig()
    preview: dict | None = None
    tracker: TrackerConfig | None = None
    trainer: TrainerConfig = TrainerConfig()
    seed: int = 42
    num_train_epochs: int = 1

    def to_dict(self) -> dict:
        return self.model_dump()

    def save_to(self, dir: Path | str, filename: str = "config.yaml") -> None:
        d = Path(dir)
        d.mkdir(parents=True, exist_ok=True)
        with open(d / filename, "w") as f:
            yaml.dump(self.to_dict(), f)

    @staticmethod
    def from_config_file(path: str) -> "TrainConfig":
        with open(path) as f:
            return TrainConfig.model_validate(yaml.safe_load(f), strict=True)
