"""Training loop (reference: src/trainer/common.py:31-469) without accelerate: one process per GPU,
gradients exchanged by ``DataParallelGrads`` (flat buffer, RCCL on a side stream), step optionally
replayed from HIP graphs.  Hook order, loss/accumulation scaling (:355-365), debug modes (:241-244,
:324-325, :440-446), checkpoint cadence and file naming follow the reference."""
from __future__ import annotations

import importlib
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from .. import kernels
from ..config import DEBUG_MODE_TYPE, TrainConfig
from ..dataloader import get_dataloader_for_bucketing
from ..dataset.util import DatasetConfig
from ..models.for_training import ModelForTraining, Runtime
from ..modules.peft import print_trainable_parameters
from ..saving import ModelSavingStrategy, get_saving_callback
from .dp import DataParallelGrads, shard_indices


def get_optimizer(optimizer_name: str, params, **kwargs) -> torch.optim.Optimizer:
    """"pkg.mod.Class" string -> optimizer (reference: src/optimizer/util.py:6-31)."""
    module_name, class_name = optimizer_name.rsplit(".", 1) if "." in optimizer_name else ("torch.optim", optimizer_name)
    try:
        cls = getattr(importlib.import_module(module_name), class_name)
    except (ImportError, AttributeError) as e:
        raise ValueError(f"Optimizer {class_name} not found in module {module_name}") from e
    lr = kwargs.pop("lr", None)
    assert lr is not None, "Learning rate must be provided"
    return cls(params, lr=lr, **kwargs)


def get_scheduler(optimizer, scheduler_name: str, **kwargs):
    module_name, class_name = scheduler_name.rsplit(".", 1) if "." in scheduler_name else ("torch.optim.lr_scheduler", scheduler_name)
    try:
        cls = getattr(importlib.import_module(module_name), class_name)
    except (ImportError, AttributeError) as e:
        raise ValueError(f"Scheduler {class_name} not found in module {module_name}") from e
    return cls(optimizer, **kwargs)


def _accepts_capturable(optimizer_name: str) -> bool:
    """True when the optimizer class takes `capturable=` (its step can then be captured into a HIP graph)."""
    import inspect

    module_name, class_name = optimizer_name.rsplit(".", 1) if "." in optimizer_name else ("torch.optim", optimizer_name)
    try:
        cls = getattr(importlib.import_module(module_name), class_name)
        return "capturable" in inspect.signature(cls.__init__).parameters
    except (ImportError, AttributeError, TypeError, ValueError):
        return False


def set_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def init_distributed() -> None:
    """torchrun / torch.distributed.run environment -> RCCL process group on this rank's GPU.
    Rehearsal hooks (tests, bench.py): VFT_FORCE_DEVICE=<i> puts every rank on GPU i and VFT_DIST_BACKEND=gloo
    exchanges through the host, so the whole N>1 path (sharding, side stream, split graphs) runs on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("VFT_FORCE_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(os.environ.get("VFT_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"))
    elif world == 1 and os.environ.get("VFT_DP_ALWAYS_REDUCE", "0") == "1" and not dist.is_initialized():
        # rehearsal hook (trainer/dp.py): a one-rank group, so that the gradient exchange really calls the backend
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(os.environ.get("VFT_DIST_BACKEND", "nccl" if torch.cuda.is_available() else "gloo"), rank=0, world_size=1)


class Trainer:
    def __init__(self, config: TrainConfig, seed: int | None = None) -> None:
        init_distributed()
        self.config = config
        self.peft_config = config.peft
        self.seed = config.seed if seed is None else seed
        self.debug_mode: DEBUG_MODE_TYPE = config.trainer.debug_mode
        self.gradient_accumulation_steps = config.trainer.gradient_accumulation_steps
        self.accelerator = Runtime()
        self.losses: list[float] = []

    def register_model_class(self, model_cls, *args, **kwargs) -> None:
        self.model_cls = model_cls
        self.model: ModelForTraining = model_cls(self.accelerator, self.config, *args, **kwargs)

    def register_train_dataset_class(self, dataset_config_class: type[DatasetConfig], *args, **kwargs) -> None:
        self.dataset_config = dataset_config_class.model_validate(self.config.dataset)

    def register_preview_dataset_class(self, dataset_config_class, *args, **kwargs) -> None:
        pass  # previews are outside the hot path

    @property
    def raw_model(self) -> ModelForTraining:
        return self.model

    def print(self, *a, **k) -> None:
        self.accelerator.print(*a, **k)

    def prepare_dataloaders(self) -> None:
        ds = self.dataset_config.get_dataset()
        rt = self.accelerator
        if rt.world > 1:  # rank r takes batches r, r+W, ... of the (seeded) order
            ds = torch.utils.data.Subset(ds, shard_indices(len(ds), rt.rank, rt.world))
        g = torch.Generator().manual_seed(self.seed)
        self.train_dataloader = get_dataloader_for_bucketing(ds, shuffle=self.dataset_config.shuffle, num_workers=0, generator=g)

    def prepare_saving_strategy(self) -> None:
        steps = len(self.train_dataloader)
        if (saving := self.config.saving) is not None:
            self.saving_strategy = ModelSavingStrategy.from_config(saving.strategy, self.config.num_train_epochs, steps)
            self.saving_callbacks = [get_saving_callback(c) for c in saving.callbacks]
        else:
            self.saving_strategy = ModelSavingStrategy(self.config.num_train_epochs, steps, None, None, False)
            self.saving_callbacks = []

    def setup_peft_if_needed(self) -> None:
        if self.peft_config is None:
            self.model._set_is_peft(False)
            return
        self.model._set_is_peft(True)
        cfgs = self.peft_config if isinstance(self.peft_config, list) else [self.peft_config]
        self.model.requires_grad_(False)  # freeze the base before wrapping (reference :179-180)
        for c in cfgs:
            c.replace_to_peft_layer(self.model)
        self.model.load_peft_weights()

    def prepare_model(self) -> None:
        self.model.before_setup_model()
        self.model.setup_model()
        self.setup_peft_if_needed()
        self.model.after_setup_model()
        print_trainable_parameters(self.model, self.print)

    def prepare_optimizer(self) -> None:
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.dp = DataParallelGrads(params)
        # gradient-norm clipping reduces over these buffers directly (ModelForTraining.after_backward)
        getattr(self, "raw_model", self.model)._flat_grad_buffers = list(self.dp.flat.buffers.values())
        args = dict(self.config.optimizer.args)
        if torch.cuda.is_available() and _accepts_capturable(self.config.optimizer.name):
            # the same optimizer kernel whether the step is replayed from HIP graphs or launched eagerly (`hip_graph: false`):
            # with the step itself bit-reproducible (round 5) the two modes then take bit-identical steps — graph replay is an
            # execution detail, not a numerical variant (tests/test_trainer_gpu.py)
            args.setdefault("capturable", True)
            if self.config.optimizer.name.rsplit(".", 1)[-1] in ("Adam", "AdamW", "SGD", "Adagrad"):
                args.setdefault("fused", True)
            # a device-resident learning rate: LR schedulers then reach the captured optimizer kernel
            args["lr"] = torch.tensor(float(args["lr"]), dtype=torch.float32, device=params[0].device)
        elif self.config.trainer.hip_graph and torch.cuda.is_available():
            self.print(f"optimizer {self.config.optimizer.name} has no capturable mode: forward/backward replay from HIP "
                       "graphs, the optimizer step runs eagerly")
        self.optimizer = get_optimizer(self.config.optimizer.name, params, **args)
        sc = self.config.scheduler
        self.scheduler = get_scheduler(self.optimizer, sc.name, **sc.args) if sc is not None else None

    def before_train(self) -> None:
        if self.config.trainer.fp32_matmul_precision is not None:
            torch.set_float32_matmul_precision(self.config.trainer.fp32_matmul_precision)
        if self.config.trainer.tuned_gemms and torch.cuda.is_available():
            from .. import tuning

            tuning.enable()
        set_seed(self.seed)
        self.prepare_dataloaders()
        self.prepare_saving_strategy()
        if self.debug_mode == "dataset":
            for batch in self.train_dataloader:
                self.print(batch)
            return
        self.prepare_model()
        self.prepare_optimizer()

    def backward(self, loss: torch.Tensor, current_step: int | None = None) -> None:
        """reference :355-359.  There DDP all-reduces inside `accelerator.backward` on boundary micro-steps, so
        `after_backward` (gradient clipping) sees the AVERAGED gradient on boundaries and the local one under
        `no_sync`; the same order here: exchange first on boundaries, then clip.  The exchange is only STARTED here
        (side stream); `apply_gradients` waits for it, so whatever the loop does in between — reading the loss back
        for the log, staging the next batch — overlaps with the collective."""
        self.raw_model.before_backward()
        (loss / self.gradient_accumulation_steps).backward()
        # `current_step=None` is the reference's own hook signature (backward(loss)): the boundary is then unknown here
        # and `apply_gradients` runs the exchange itself, un-overlapped but never skipped.
        self._boundary = current_step is not None and current_step % self.gradient_accumulation_steps == 0
        self._clipped = False
        if self._boundary:
            self.dp.reduce_async()  # skipped on non-boundary micro-steps == the reference's no_sync (:301-308)
        elif current_step is not None:
            self.raw_model.after_backward()
            self._clipped = True

    _boundary = False
    _clipped = False

    def apply_gradients(self, current_step: int) -> None:
        if current_step % self.gradient_accumulation_steps == 0:
            if not self._boundary and not self._clipped:
                # a caller on the old backward(loss) / apply_gradients(step) protocol: the exchange was not started in
                # backward, so do it here — otherwise the ranks would step on local gradients and diverge silently
                self.dp.reduce_async()
                self._boundary = True
            if self._boundary:
                self.dp.wait()
                self.raw_model.after_backward()  # clipping on the averaged gradient
                self._boundary = False
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            self.dp.zero_grad()

    def call_saving_callbacks(self, epoch: int, steps: int) -> None:
        if not self.saving_strategy.should_save(epoch, steps):
            return
        self.accelerator.wait_for_everyone()
        kernels.nf4_check_barrier()  # never checkpoint weights trained through a timed-out NF4 launch
        if self.saving_callbacks and self.accelerator.is_main_process:
            sd, meta = self.raw_model.get_state_dict_to_save(), self.raw_model.get_metadata_to_save()
            for cb in self.saving_callbacks:
                cb.save_state_dict(sd, epoch, steps, metadata=meta)
        self.accelerator.wait_for_everyone()

    def training_loop(self) -> None:
        """One optimizer step per accumulation boundary.  Order on a boundary: forward + backward (graph A) ->
        gradient exchange STARTS on the side stream -> host work that does not need the update (loss read-back for the
        log, the next batch staged into its static tensors) -> wait -> clip -> optimizer (graph B) -> LR scheduler ->
        checkpoint cadence.  The collective therefore overlaps with the logging and the next batch's staging; the next
        forward needs the updated weights and cannot start earlier."""
        step = 0
        graphs: dict = {}
        use_graph = (self.config.trainer.hip_graph and torch.cuda.is_available()
                     and getattr(self.raw_model, "graph_safe", False))
        ga = self.gradient_accumulation_steps
        opt_graph = None  # shared optimizer graph when the step is split (DP or accumulation)
        for epoch in range(1, self.config.num_train_epochs + 1):
            self.raw_model.before_train_epoch()
            it = iter(self.train_dataloader)
            batch = next(it, None)
            staged = False
            while batch is not None:
                step += 1
                self.raw_model.before_train_step()
                boundary = step % ga == 0
                nxt = None
                if use_graph:
                    key = self.raw_model._key if staged else self.raw_model.stage_batch(batch)  # batch -> static tensors
                    if key not in graphs:
                        from .graph import GraphedTrainStep

                        first = next(iter(graphs.values()), None)
                        graphs[key] = GraphedTrainStep(self.raw_model.train_step_static, self.optimizer, self.dp,
                                                       after_backward=self.raw_model.after_backward, split=ga > 1 or None,
                                                       opt_graph=opt_graph, pool=first.pool() if first is not None else None,
                                                       loss_scale=1.0 / ga)
                        graphs[key].defer_finish = True
                        opt_graph = graphs[key].graph_b
                    loss = graphs[key](boundary)
                    if step % self.raw_model.log_every == 0 or self.debug_mode == "1step":
                        self.losses.append(float(loss.detach()))
                        kernels.nf4_check_barrier()  # the host has just synchronised for the loss: read the NF4 error word too
                    nxt = next(it, None)
                    staged = nxt is not None and self.debug_mode != "1step"
                    if staged:
                        self.raw_model.stage_batch(nxt)  # queued behind graph A; the exchange runs beside it
                    graphs[key].finish()
                    if boundary and self.scheduler is not None:
                        self.scheduler.step()
                else:
                    loss = self.model(batch)
                    self.backward(loss, step)
                    if step % self.raw_model.log_every == 0 or self.debug_mode == "1step":
                        self.losses.append(float(loss.detach()))
                        kernels.nf4_check_barrier()
                    nxt = next(it, None)
                    self.apply_gradients(step)
                self.call_saving_callbacks(epoch, step)
                self.raw_model.after_train_step()
                if self.debug_mode == "1step":
                    break
                batch = nxt
            self.raw_model.after_train_epoch()
            if self.debug_mode == "1step":
                break

    def train(self) -> None:
        self.before_train()
        if self.debug_mode == "dataset":
            return
        self.raw_model.sanity_check()
        if self.debug_mode == "sanity_check":
            self.print("Sanity check done. Exiting...")
            return
        self.training_loop()
