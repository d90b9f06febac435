"""HIP-graph capture of the training step.

A LoRA step of the SDXL UNet is ~8 000 kernel launches, most of them a few microseconds long, so an
eagerly launched step is bound by the host (Python + launch latency), not by the GPU.  The step is
therefore captured once per batch shape (= per aspect-ratio bucket) into HIP graphs and replayed:

    graph A:  noising -> denoiser forward -> loss -> backward      (grads land in the flat buffer)
    ---- RCCL all-reduce of the flat gradient buffer on the side stream (DataParallelGrads) ----
    graph B:  optimizer.step() + zero the flat buffer

With one GPU, A and B are captured as a single graph.  Inputs live in static tensors that the
caller fills (``copy_``) before each replay; random draws inside the graph (timesteps, noise) use
PyTorch's graph-safe Philox generator, so every replay sees fresh randomness.
"""
from __future__ import annotations

from typing import Callable

import torch

from .. import kernels as _kernels
from .dp import DataParallelGrads


def _init_optimizer_state(optimizer: torch.optim.Optimizer, dp: DataParallelGrads) -> None:
    """Allocate the optimizer's lazy state with a no-op step: zero gradients and zero learning rate (no update,
    no weight decay, moments stay zero), then rewind the step counters."""
    dp.zero_grad()
    lrs = []
    for g in optimizer.param_groups:
        lrs.append(g["lr"].clone() if torch.is_tensor(g["lr"]) else g["lr"])
        if torch.is_tensor(g["lr"]):
            g["lr"].zero_()
        else:
            g["lr"] = 0.0
    optimizer.step()
    for g, lr in zip(optimizer.param_groups, lrs):
        if torch.is_tensor(g["lr"]):
            g["lr"].copy_(lr)
        else:
            g["lr"] = lr
    for st in optimizer.state.values():
        if "step" in st:
            if torch.is_tensor(st["step"]):
                st["step"].zero_()
            else:
                st["step"] = 0


class GraphedOptimizerStep:
    """optimizer.step() + zeroing of the flat gradient buffer as one graph (shape-independent: shared by
    every bucket's forward/backward graph when the step is split)."""

    def __init__(self, optimizer: torch.optim.Optimizer, dp: DataParallelGrads, pool=None):
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            optimizer.step()
            dp.zero_grad()

    def replay(self) -> None:
        self.graph.replay()


class EagerOptimizerStep:
    """Stand-in for `GraphedOptimizerStep` when the optimizer cannot be captured (no `capturable` mode: host-side
    step counters / learning rates would be baked into the graph): forward + backward still replay from a graph,
    the optimizer runs eagerly on accumulation boundaries."""

    def __init__(self, optimizer: torch.optim.Optimizer, dp: DataParallelGrads):
        self.optimizer, self.dp = optimizer, dp

    def replay(self) -> None:
        self.optimizer.step()
        self.dp.zero_grad()


def optimizer_is_capturable(optimizer: torch.optim.Optimizer) -> bool:
    return all(bool(g.get("capturable", False)) for g in optimizer.param_groups)


class GraphedTrainStep:
    """One bucket shape's training step as HIP graphs.

    split=False (single GPU, no accumulation): forward + backward + optimizer in one graph.
    split=True  (DP or gradient accumulation): graph A = forward + backward (gradients accumulate in the flat
    buffer); `after_backward` (gradient clipping) is its own small graph C, replayed after A on non-boundary
    micro-steps (local gradients, the reference under `no_sync`) and AFTER the all-reduce on boundaries — the
    reference clips the averaged gradient there (DDP reduces inside backward, src/trainer/common.py:355-359);
    the optimizer graph B (`opt_graph`, shared between bucket shapes) is replayed only on boundaries.
    `pool` lets all bucket graphs share one private memory pool — they never replay concurrently, so the
    activation memory is paid once, not once per bucket.  An optimizer without a capturable mode runs eagerly
    in place of graph B."""

    def __init__(self, forward_loss: Callable[[], torch.Tensor], optimizer: torch.optim.Optimizer,
                 dp: DataParallelGrads, warmup: int = 3, after_backward: Callable[[], None] | None = None,
                 split: bool | None = None, opt_graph: GraphedOptimizerStep | None = None, pool=None,
                 loss_scale: float = 1.0):
        self.forward_loss = forward_loss
        self.optimizer = optimizer
        self.dp = dp
        self.after_backward = after_backward
        self.loss_scale = float(loss_scale)  # 1 / gradient_accumulation_steps; the reported loss stays unscaled
        self.capturable = optimizer_is_capturable(optimizer)
        self.single = not dp.collective and not split and self.capturable
        self.loss: torch.Tensor | None = None
        self._eager(warmup)  # allocator, MIOpen find, lazy kernel attributes: everything un-capturable happens here
        self._capture(pool, opt_graph)

    def _eager(self, n: int) -> None:
        """Eager passes that leave the run exactly where it was: everything un-capturable (allocator growth,
        MIOpen find, lazy kernel attributes, lazy optimizer state) happens here, but gradients already
        accumulated in the flat buffer, the weights, the moments and the step counts are preserved — a graph
        run takes the same optimizer steps as an eager run."""
        saved = {dt: b.clone() for dt, b in self.dp.flat.buffers.items()}
        rng = torch.cuda.get_rng_state()  # the warm-up's timestep / noise draws are not part of the run
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n):
                loss = self.forward_loss()
                (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
                if self.after_backward is not None:
                    self.after_backward()
            if not any(len(st) for st in self.optimizer.state.values()):
                _init_optimizer_state(self.optimizer, self.dp)
            for dt, b in self.dp.flat.buffers.items():
                b.copy_(saved[dt])
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        torch.cuda.set_rng_state(rng)

    def _capture(self, pool, opt_graph) -> None:
        # the warm-up's activations are cached blocks of the default pool; the capture allocates the same amount again in
        # the graph's private pool — hand the cached blocks back first, or every step costs its memory twice
        import gc

        gc.collect()
        torch.cuda.empty_cache()
        self.graph_a = torch.cuda.CUDAGraph()
        # The reported loss lives in its own one-element buffer allocated BEFORE the capture (default pool, never recycled
        # by the graph's private pool) and is copied there right behind the forward pass: F.mse_loss used to return a 0-dim
        # tensor that owns the whole elementwise buffer (6 MB at the bench size), and round 4 found element 0 of that
        # buffer — "the loss" — overwritten by the time a replay ends once the process had called torch.cuda.synchronize()
        # (the reduction itself is our own kernel since then).
        static_loss = torch.zeros((), dtype=torch.float32, device=next(iter(self.dp.flat.buffers.values())).device)
        with torch.cuda.graph(self.graph_a, pool=pool):
            loss = self.forward_loss()
            static_loss.copy_(loss.detach())
            (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
            if self.single:
                if self.after_backward is not None:
                    self.after_backward()
                self.optimizer.step()
                self.dp.zero_grad()
            self.loss = static_loss
            self._raw_loss = loss.detach()  # (diagnostics: tools/exp/graph_loss_debug.py)
        self.graph_b = self.graph_c = None
        if not self.single:
            if self.after_backward is not None:
                self.graph_c = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_c, pool=self.graph_a.pool()):
                    self.after_backward()
            if opt_graph is not None:
                self.graph_b = opt_graph
            elif self.capturable:
                self.graph_b = GraphedOptimizerStep(self.optimizer, self.dp, pool=self.graph_a.pool())
            else:
                self.graph_b = EagerOptimizerStep(self.optimizer, self.dp)

    def pool(self):
        return self.graph_a.pool()

    def __call__(self, boundary: bool = True) -> torch.Tensor:
        """One micro-step (forward + backward); on an accumulation boundary also the gradient exchange and the
        optimizer step.  Returns the (device-resident, static) loss tensor of this micro-step."""
        _kernels.nf4_note_replay()  # the replaying stream takes over the per-device NF4 scratch (no-op without NF4 layers)
        self.graph_a.replay()
        if self.graph_b is None:
            return self.loss
        if boundary:
            self.dp.reduce_async()  # side stream; whatever the host queues before finish() overlaps with it
            self._open = True
            if not self.defer_finish:
                self.finish()
        elif self.graph_c is not None:
            self.graph_c.replay()
        return self.loss

    defer_finish = False  # True: the caller runs finish() itself (after staging the next batch, logging, ...)
    _open = False

    def finish(self) -> None:
        """Second half of a boundary step: wait for the gradient exchange, clip the averaged gradient, update."""
        if not self._open:
            return
        self._open = False
        self.dp.wait()
        if self.graph_c is not None:
            self.graph_c.replay()
        self.graph_b.replay()
