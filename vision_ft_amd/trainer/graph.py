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

from .dp import DataParallelGrads


class GraphedTrainStep:
    def __init__(self, forward_loss: Callable[[], torch.Tensor], optimizer: torch.optim.Optimizer,
                 dp: DataParallelGrads, warmup: int = 3, after_backward: Callable[[], None] | None = None):
        self.forward_loss = forward_loss
        self.optimizer = optimizer
        self.dp = dp
        self.after_backward = after_backward
        self.single = dp.world == 1
        self.loss: torch.Tensor | None = None
        self._eager(warmup)  # allocator, MIOpen find, lazy kernel attributes: everything un-capturable happens here
        self._capture()

    def _eager_once(self) -> torch.Tensor:
        loss = self.forward_loss()
        loss.backward()
        if self.after_backward is not None:
            self.after_backward()
        self.dp.reduce_async()
        self.dp.wait()
        self.optimizer.step()
        self.dp.zero_grad()
        return loss

    def _eager(self, n: int) -> None:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(n):
                self._eager_once()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def _capture(self) -> None:
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a):
            loss = self.forward_loss()
            loss.backward()
            if self.after_backward is not None:
                self.after_backward()
            if self.single:
                self.optimizer.step()
                self.dp.zero_grad()
            self.loss = loss.detach()
        self.graph_b = None
        if not self.single:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool()):
                self.optimizer.step()
                self.dp.zero_grad()

    def __call__(self) -> torch.Tensor:
        """One optimizer step; returns the (device-resident, static) loss tensor of this step."""
        self.graph_a.replay()
        if self.graph_b is not None:
            self.dp.reduce_async()
            self.dp.wait()
            self.graph_b.replay()
        return self.loss
