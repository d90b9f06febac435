"""AuraFlow v0.3 text-to-image fine-tuning on MI355X — counterpart of the reference's
``train/auraflow/text_to_image.py`` (``--config <yaml>``)."""
import os
import sys

import click
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from vision_ft_amd.config import TrainConfig  # noqa: E402
from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig  # noqa: E402
from vision_ft_amd.models.auraflow import AuraFlowConig, Denoiser  # noqa: E402
from vision_ft_amd.modules.loss.flow_match import loss_with_predicted_velocity, prepare_noised_latents  # noqa: E402
from vision_ft_amd.modules.timestep.sampling import sigmoid_randn  # noqa: E402
from vision_ft_amd.models.auraflow.pipeline import AuraFlowModel  # noqa: E402
from vision_ft_amd.trainer.common import Trainer  # noqa: E402
from vision_ft_amd.trainer.tasks import DenoiserTask  # noqa: E402


class AuraFlowForTraining(DenoiserTask):
    model_config_class = AuraFlowConig
    latent_channels = 4
    pipeline_class = AuraFlowModel


    def make_conditions(self, batch, width, height, seed):
        dev = self.accelerator.device
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        d = self.model_config.denoiser.joint_attention_dim
        return {"ehs": torch.randn(batch, 256, d, generator=g, device=dev).to(torch.bfloat16)}

    def denoise_loss(self, latents, cond):
        # reference train_step :73-96: t ~ sigmoid(N(0,1)); x_t = (1-t) x + t eps; target eps - x
        t = sigmoid_randn(latents.shape, latents.device)
        noisy, noise = prepare_noised_latents(latents, t.to(latents.dtype))
        pred = self.model(latent=noisy, encoder_hidden_states=cond["ehs"], timestep=t)
        return loss_with_predicted_velocity(latents, noise, pred)

    def convert_adapter_key(self, key: str) -> str:
        return "diffusion_model." + key  # src/models/auraflow/pipeline.py:42-46


@click.command()
@click.option("--config", type=str, required=True)
def main(config: str):
    cfg = TrainConfig.from_config_file(config)
    trainer = Trainer(cfg)
    trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
    trainer.register_model_class(AuraFlowForTraining)
    trainer.train()
    trainer.print(f"done: {len(trainer.losses)} logged losses, last {trainer.losses[-1] if trainer.losses else None}")


if __name__ == "__main__":
    main()
