"""The mean-squared-error reduction shared by the losses (reference: F.mse_loss(..., reduction="mean") in
src/modules/loss/diffusion.py:62-68, src/modules/loss/flow_match.py:85-96, train/sdxl/flow_match.py:130-150; under the
trainer's autocast mse_loss computes in fp32).

GPU tensors take `kernels.mse_loss` (fp32 difference by ATen, mean of squares on the two-launch HIP kernel `vft_mean_sq`): ATen's
one-launch reduction (workgroups meeting through a semaphore buffer that a memset node zeroes) returned garbage inside the
captured training step once the process had synchronised an idle stream — found in round 4, see csrc/eltwise.hip.  CPU tensors
(host-side tests of the loss arithmetic) keep the torch op."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def mse_mean(prediction: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    if prediction.is_cuda:
        from ... import kernels

        return kernels.mse_loss(prediction, target)
    return F.mse_loss(prediction.float(), target.float(), reduction="mean")
