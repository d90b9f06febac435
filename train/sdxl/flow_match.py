"""SDXL flow-matching fine-tuning on MI355X — entrypoint compatible with the reference's
``train/sdxl/flow_match.py`` (``--config <yaml>``; same TrainConfig / model-config schema).

The step is the reference's `train_step` (:109-174) from the timestep draw on: `sample_timestep(...) * 1000`,
`prepare_scaled_noised_latents(latents, t / 1000, noise_scale, clean_at_zero)`, the denoiser called with the
0..1000 timestep, and `treat_loss` (:51-107) with its velocity / image prediction branches.

    python train/sdxl/flow_match.py --config configs/sdxl/flow_match_lora.yml
"""
import os
import sys

import click
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from vision_ft_amd.config import TrainConfig  # noqa: E402
from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig  # noqa: E402
from vision_ft_amd.models.sdxl.adapter.flow_match import SDXLFlowMatch, SDXLFlowMatchConfig  # noqa: E402
from vision_ft_amd.modules.loss.flow_match import (ModelPredictionType, convert_x0_to_velocity,  # noqa: E402
                                                   loss_with_predicted_velocity, prepare_scaled_noised_latents)
from vision_ft_amd.modules.loss.reduce import mse_mean  # noqa: E402
from vision_ft_amd.modules.timestep.sampling import TimestepSamplingType, sample_timestep  # noqa: E402
from vision_ft_amd.trainer.common import Trainer  # noqa: E402

from text_to_image import SDXLForTextToImageTraining  # noqa: E402


class SDXLForFlowMatchingTrainingConfig(SDXLFlowMatchConfig):
    max_token_length: int = 225  # 75 * 3

    loss_type: ModelPredictionType = "velocity"

    timestep_sampling: TimestepSamplingType = "scale_shift_sigmoid"
    timestep_std: float = 0.8
    timestep_mean: float = -0.8


class SDXLForFlowMatchingTraining(SDXLForTextToImageTraining):
    model_config_class = SDXLForFlowMatchingTrainingConfig
    pipeline_class = SDXLFlowMatch

    def treat_loss(self, model_pred: torch.Tensor, latents: torch.Tensor, random_noise: torch.Tensor,
                   noisy_latents: torch.Tensor, timestep: torch.Tensor) -> torch.Tensor:
        cfg = self.model_config
        if cfg.model_prediction == "velocity":
            if cfg.loss_type == "velocity":
                return loss_with_predicted_velocity(latents=latents, random_noise=random_noise, predicted_velocity=model_pred)
            raise NotImplementedError(f"loss_type {cfg.loss_type} not implemented for velocity prediction")
        if cfg.model_prediction == "image":
            if cfg.loss_type == "velocity":
                target_v = convert_x0_to_velocity(x0=latents, noisy_latents=noisy_latents, timestep=timestep,
                                                  eps=cfg.timestep_eps, clean_at_zero=cfg.clean_at_zero)
                v_pred = convert_x0_to_velocity(x0=model_pred, noisy_latents=noisy_latents, timestep=timestep,
                                                eps=cfg.timestep_eps, clean_at_zero=cfg.clean_at_zero)
                return mse_mean(v_pred, target_v)
            if cfg.loss_type == "image":
                return mse_mean(model_pred, latents.clone().detach())
            raise NotImplementedError(f"loss_type {cfg.loss_type} not implemented for image prediction")
        raise ValueError(f"Unknown model_prediction: {cfg.model_prediction}")

    def denoise_loss(self, latents, cond):
        cfg = self.model_config
        with torch.no_grad():
            # jit sampling arguments, then `shift` / `sigmoid_scale` for the flux-style samplers; like the reference,
            # every keyword goes to the selected sampler, so only samplers that accept them all can be configured
            timesteps = sample_timestep(latents_shape=latents.shape, device=latents.device,
                                        sampling_type=cfg.timestep_sampling, std=cfg.timestep_std,
                                        mean=cfg.timestep_mean, shift=3.1825, sigmoid_scale=1) * 1000
        noisy, noise = prepare_scaled_noised_latents(latents=latents, timestep=timesteps / 1000,
                                                     noise_scale=cfg.noise_scale, clean_at_zero=cfg.clean_at_zero)
        pred = self.model(latents=noisy, timestep=timesteps, encoder_hidden_states=cond["ehs"],
                          encoder_pooler_output=cond["pooled"], original_size=cond["original_size"],
                          target_size=cond["target_size"], crop_coords_top_left=cond["crop"])
        return self.treat_loss(model_pred=pred, latents=latents, random_noise=noise, noisy_latents=noisy,
                               timestep=timesteps / 1000)


@click.command()
@click.option("--config", type=str, required=True)
def main(config: str):
    cfg = TrainConfig.from_config_file(config)
    trainer = Trainer(cfg)
    trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
    trainer.register_model_class(SDXLForFlowMatchingTraining)
    trainer.train()
    trainer.print(f"done: {len(trainer.losses)} logged losses, last {trainer.losses[-1] if trainer.losses else None}")


if __name__ == "__main__":
    main()
