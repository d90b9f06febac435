"""Lumina-Image-2.0 text-to-image fine-tuning on MI355X — counterpart of the reference's
``train/lumina2/text_to_image.py`` (``--config <yaml>``)."""
import os
import sys
from typing import Literal

import click
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from vision_ft_amd.config import TrainConfig  # noqa: E402
from vision_ft_amd.dataset.synthetic import SyntheticLatentDatasetConfig  # noqa: E402
from vision_ft_amd.models.lumina2 import Denoiser, Lumina2Config, Scheduler  # noqa: E402
from vision_ft_amd.modules.loss.flow_match import loss_with_predicted_velocity, prepare_noised_latents  # noqa: E402
from vision_ft_amd.modules.loss.reduce import mse_mean  # noqa: E402
from vision_ft_amd.modules.timestep.sampling import shift_fraction_uniform_rand, uniform_rand  # noqa: E402
from vision_ft_amd.models.lumina2.pipeline import Lumina2  # noqa: E402
from vision_ft_amd.trainer.common import Trainer  # noqa: E402
from vision_ft_amd.trainer.tasks import DenoiserTask  # noqa: E402


class Lumina2ForTextToImageTrainingConfig(Lumina2Config):
    max_token_length: int = 256
    timestep_sampling: Literal["uniform", "lognorm", "shift_fraction_uniform"] = "uniform"
    timestep_fraction_divisible: list[int] = [20, 25, 30, 32]
    # the reference's default `use_lowres_loss: true` crashes (tuple added to a tensor,
    # train/lumina2/text_to_image.py:207-216); its shipped YAML sets false, which is the only mode here
    use_lowres_loss: bool = False
    use_downsampled_velocity_loss: bool = False


class Lumina2ForTextToImageTraining(DenoiserTask):
    model_config_class = Lumina2ForTextToImageTrainingConfig
    latent_channels = 16
    pipeline_class = Lumina2
    # Round 4: the step runs the shape-static forward (`NextDiT._forward_static`): caption lengths never leave the device, every
    # batch of a bucket has the same tensor shapes, so the trainer replays it from one HIP graph per bucket like the other two
    # families (rounds 1-3: eager, one host read of the lengths per step).  `static_layout = False` restores that path.
    static_layout = True

    @property
    def graph_safe(self) -> bool:  # (the `shift_fraction_uniform` sampler draws on the host with numpy: not capturable)
        return self.static_layout and self.model_config.timestep_sampling != "shift_fraction_uniform"


    def make_conditions(self, batch, width, height, seed):
        dev = self.accelerator.device
        g = torch.Generator(device=dev).manual_seed(seed + 1)
        L = self.model_config.max_token_length
        lens = torch.randint(32, L + 1, (batch,), generator=g, device=dev)
        return {"caption": torch.randn(batch, L, self.model_config.denoiser.caption_dim, generator=g, device=dev).to(torch.bfloat16),
                "mask": torch.arange(L, device=dev).unsqueeze(0) < lens.unsqueeze(1)}

    def sample_timesteps(self, latents_shape: torch.Size) -> torch.Tensor:
        """train/lumina2/text_to_image.py:144-171 of the reference: the `timestep_sampling` switch."""
        kind, dev = self.model_config.timestep_sampling, self.accelerator.device
        if kind == "uniform":
            return uniform_rand(latents_shape, dev)
        if kind == "lognorm":
            return self.scheduler.sample_sigmoid_randn(latents_shape, dev, patch_size=self.model_config.denoiser.patch_size)
        if kind == "shift_fraction_uniform":  # host-side numpy draw: this mode cannot be replayed from a HIP graph
            return 1 - shift_fraction_uniform_rand(latents_shape, dev, shift=self.scheduler.shift,
                                                   divisible=self.model_config.timestep_fraction_divisible)
        raise ValueError(f"Unknown timestep sampling method: {kind}. ")

    scheduler = Scheduler()

    def denoise_loss(self, latents, cond):
        # reference forward_and_loss :100-142: Lumina2 time runs 0 (noise) -> 1 (clean); prediction sign flipped
        t = self.sample_timesteps(latents.shape)
        noisy, noise = prepare_noised_latents(latents, (1 - t).to(latents.dtype))
        static_ok = self.static_layout and cond.get("caption_lengths") is None and noisy.dim() == 4
        if not static_ok and noisy.is_cuda and torch.cuda.is_current_stream_capturing():
            # `graph_safe` promises the shape-static forward; the ragged one reads lengths on the host (fails under capture) or,
            # given host integers, would bake THIS batch's lengths and layout into the graph and replay them for every later
            # batch.  A producer that hands over `caption_lengths` or a list of latents has to run eagerly.
            raise RuntimeError("Lumina2 step captured into a HIP graph needs the static layout: stacked (B,C,H,W) latents and no "
                               "`caption_lengths` in the conditions (set trainer.hip_graph: false for ragged batches)")
        if static_ok:
            out, _, _ = self.model(latents=noisy, timestep=t, caption_features=cond["caption"], caption_mask=cond["mask"],
                                   static_layout=True)
        else:
            out, _, _ = self.model(latents=noisy, timestep=t, caption_features=cond["caption"], caption_mask=cond["mask"],
                                   caption_lengths=cond.get("caption_lengths"),  # host integers when the producer has them
                                   padded_output=True)  # = to_padded_tensor(nested output, 0.0), capturable
        pred = -out
        loss = loss_with_predicted_velocity(latents, noise, pred)
        if self.model_config.use_downsampled_velocity_loss:
            loss = loss + mse_mean(F.avg_pool2d(pred.float(), 4, 4), F.avg_pool2d((noise - latents).float(), 4, 4))
        return loss

    def convert_adapter_key(self, key: str) -> str:
        return "diffusion_model." + key  # src/models/lumina2/util.py:162-168


@click.command()
@click.option("--config", type=str, required=True)
def main(config: str):
    cfg = TrainConfig.from_config_file(config)
    trainer = Trainer(cfg)
    trainer.register_train_dataset_class(SyntheticLatentDatasetConfig)
    trainer.register_model_class(Lumina2ForTextToImageTraining)
    trainer.train()
    trainer.print(f"done: {len(trainer.losses)} logged losses, last {trainer.losses[-1] if trainer.losses else None}")


if __name__ == "__main__":
    main()
