"""SDXL text-to-image fine-tuning on MI355X — entrypoint compatible with the reference's
``train/sdxl/text_to_image.py`` (``--config <yaml>``; same TrainConfig schema).

    python train/sdxl/text_to_image.py --config configs/sdxl/text_to_image_lora.yml
    python -m torch.distributed.run --nproc-per-node 8 train/sdxl/text_to_image.py --config ...
"""
import os
import sys

import click
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from vision_ft_amd.config import TrainConfig  # noqa: E402
from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig  # noqa: E402
from vision_ft_amd.models.sdxl import Denoiser, SDXLConfig  # noqa: E402
from vision_ft_amd.models.sdxl.util import convert_to_comfy_key  # noqa: E402
from vision_ft_amd.modules.loss.diffusion import loss_with_predicted_noise, prepare_noised_latents  # noqa: E402
from vision_ft_amd.modules.timestep.sampling import uniform_randint  # noqa: E402
from vision_ft_amd.models.sdxl.pipeline import SDXLModel  # noqa: E402
from vision_ft_amd.trainer.common import Trainer  # noqa: E402
from vision_ft_amd.trainer.tasks import DenoiserTask  # noqa: E402


class SDXLForTextToImageTrainingConfig(SDXLConfig):
    max_token_length: int = 225  # 75 * 3 -> 77 * 3 context tokens


class SDXLForTextToImageTraining(DenoiserTask):
    model_config_class = SDXLForTextToImageTrainingConfig
    latent_channels = 4
    pipeline_class = SDXLModel


    def make_conditions(self, batch, width, height, seed):
        dev = self.accelerator.device
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        n_ctx = 77 * ((self.model_config.max_token_length + 74) // 75)
        size = torch.tensor([[height, width]] * batch, device=dev)
        return {
            "ehs": torch.randn(batch, n_ctx, self.model_config.denoiser.context_dim, generator=g, device=dev).to(torch.bfloat16),
            "pooled": torch.randn(batch, 1280, generator=g, device=dev).to(torch.bfloat16),
            "original_size": size, "target_size": size.clone(),
            "crop": torch.zeros(batch, 2, dtype=torch.long, device=dev),
        }

    def conditions_from_batch(self, batch, text_cond):
        # real-image batches bring their own size conditions (dataset/text_to_image.py; reference train_step :106-112)
        return {"ehs": text_cond["ehs"].to(torch.bfloat16), "pooled": text_cond["pooled"].to(torch.bfloat16),
                "original_size": batch["original_size"], "target_size": batch["target_size"],
                "crop": batch["crop_coords_top_left"]}

    def denoise_loss(self, latents, cond):
        # train_step of the reference, :114-143
        timesteps = uniform_randint(latents.shape, latents.device, 0, 1000)
        noisy, noise = prepare_noised_latents(latents, timesteps)
        pred = self.model(latents=noisy, timestep=timesteps, encoder_hidden_states=cond["ehs"],
                          encoder_pooler_output=cond["pooled"], original_size=cond["original_size"],
                          target_size=cond["target_size"], crop_coords_top_left=cond["crop"])
        return loss_with_predicted_noise(latents, noise, pred)

    def convert_adapter_key(self, key: str) -> str:
        return convert_to_comfy_key("denoiser." + key)  # the task holds the denoiser itself


@click.command()
@click.option("--config", type=str, required=True)
def main(config: str):
    cfg = TrainConfig.from_config_file(config)
    trainer = Trainer(cfg)
    trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
    trainer.register_model_class(SDXLForTextToImageTraining)
    trainer.train()
    trainer.print(f"done: {len(trainer.losses)} logged losses, last {trainer.losses[-1] if trainer.losses else None}")


if __name__ == "__main__":
    main()
