"""Shared base of the three text-to-image training tasks (train/*/text_to_image.py).

The reference tasks (train/sdxl/text_to_image.py:28-210 etc.) encode captions and images with the
frozen text encoders / VAE inside ``train_step``; those are out of the hot path (SURVEY.md §2 rows
19-20), so a batch here already carries latents and text embeddings — drawn synthetically, per
aspect-ratio bucket, by ``SyntheticLatentDatasetConfig`` — and everything from the timestep draw on
is the reference's step.  Batches are staged into static device tensors per bucket shape so the
step can be replayed from a HIP graph."""
from __future__ import annotations

import torch

from ..dataset.synthetic import make_latents
from ..models.for_training import ModelForTraining
from ..modules.peft import get_adapter_parameters
from ..utils.dtype import str_to_dtype


class DenoiserTask(ModelForTraining):
    graph_safe = True
    latent_channels = 4
    pipeline_class: type[torch.nn.Module]  # models/<family>/pipeline.py: holds `.denoiser`, loads checkpoints

    def make_conditions(self, batch: int, width: int, height: int, seed: int) -> dict[str, torch.Tensor]:
        raise NotImplementedError

    def denoise_loss(self, latents: torch.Tensor, cond: dict[str, torch.Tensor]) -> torch.Tensor:
        raise NotImplementedError

    # ---- ModelForTraining protocol -----------------------------------------------------------
    def setup_model(self) -> None:
        """Reference tasks: `init_empty_weights()` -> pipeline -> `_from_checkpoint()` (train/sdxl/text_to_image.py:34-42):
        original key names are converted, pre-quantised (NF4) Linears are recognised by their checkpoint keys, the
        state dict is assigned strictly.  `checkpoint_path: random` keeps the default initialisation (benchmarks)."""
        dev = self.accelerator.device
        dtype = str_to_dtype(self.model_config.dtype)
        path = self.model_config.checkpoint_path
        if path in ("", "random", None):
            with torch.device(dev):
                pipe = self.pipeline_class(self.model_config)
            pipe.denoiser.to(dtype)
        else:
            with torch.device("meta"):
                pipe = self.pipeline_class(self.model_config)
            pipe._from_checkpoint()
            pipe.denoiser.to(device=dev)
            for mod in pipe.denoiser.modules():  # floating-point weights in the run's dtype; packed NF4 payloads stay as they are
                if not hasattr(mod, "quant_state"):
                    for p_ in mod.parameters(recurse=False):
                        if p_.is_floating_point():
                            p_.data = p_.data.to(dtype)
                    for b_name, b_ in list(mod.named_buffers(recurse=False)):
                        if b_ is not None and b_.is_floating_point():
                            setattr(mod, b_name, b_.to(dtype))
        self.model = pipe.denoiser
        self._static: dict = {}

    def after_setup_model(self) -> None:
        if self.config.trainer.gradient_checkpointing:
            for name in ("set_gradient_checkpointing", "_set_gradient_checkpointing"):
                if hasattr(self.model, name):
                    getattr(self.model, name)(True)
        super().after_setup_model()

    def sanity_check(self) -> None:
        with torch.no_grad():
            cond = self.make_conditions(1, 768, 768, seed=0)
            lat = make_latents(768, 768, 1, self.latent_channels, self.accelerator.device, seed=0)
            self.denoise_loss(lat, cond)

    # ---- real-image front half (reference train_step: text encode + VAE encode under no_grad, train/sdxl/text_to_image.py:
    # 114-135).  The frozen encoders themselves are outside the hot path: they are attached as callables.
    image_encoder = None   # (B,3,H,W) fp16 in -1..1 -> latents (B,C,H/8,W/8), scaling factor applied (`encode_image`)
    prompt_encoder = None  # list[str] -> dict of text-condition tensors (the keys `make_conditions` produces)

    def attach_encoders(self, image_encoder=None, prompt_encoder=None) -> None:
        self.image_encoder = image_encoder or self.image_encoder
        self.prompt_encoder = prompt_encoder or self.prompt_encoder

    def conditions_from_batch(self, batch: dict, text_cond: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """Task hook: merge the encoded captions with the batch's own conditions (SDXL: the three size conditions)."""
        return text_cond

    @torch.no_grad()
    def _stage_image_batch(self, batch: dict) -> tuple:
        if self.image_encoder is None or self.prompt_encoder is None:
            raise RuntimeError(
                "this batch carries pixels and captions (TextToImageDatasetConfig), but no frozen VAE / text encoder is "
                "attached: call task.attach_encoders(image_encoder=..., prompt_encoder=...) — they are outside the "
                "MI355X hot path and are not built by vision_ft_amd — or train from SyntheticLatentDatasetConfig")
        dev = self.accelerator.device
        images = batch["image"].to(dev, non_blocking=True)
        lat = self.image_encoder(images)
        cond = self.conditions_from_batch(batch, self.prompt_encoder(list(batch["caption"])))
        return self._stage_static("image", lat, {k: v.to(dev) for k, v in cond.items()})

    @torch.no_grad()
    def _stage_cached_batch(self, batch: dict) -> tuple:
        """A batch of `CachedLatentDatasetConfig` (dataset/latent_cache.py): latents and text conditions were encoded once,
        off this box; the size conditions are the ones the dataset recorded with each crop."""
        from ..dataset.latent_cache import split_text_conditions

        dev = self.accelerator.device
        lat = batch["latents"].to(dev, str_to_dtype(self.model_config.dtype), non_blocking=True)
        cond = self.conditions_from_batch(batch, split_text_conditions(batch))
        return self._stage_static("cached", lat, {k: v.to(dev) for k, v in cond.items()})

    def _stage_static(self, tag: str, lat: torch.Tensor, cond: dict[str, torch.Tensor]) -> tuple:
        key = (tag, tuple(lat.shape), tuple((k, tuple(v.shape)) for k, v in sorted(cond.items())))
        if key not in self._static:
            self._static[key] = (lat.clone(), {k: v.clone() for k, v in cond.items()})
        else:
            s_lat, s_cond = self._static[key]
            s_lat.copy_(lat)
            for k, v in cond.items():
                s_cond[k].copy_(v)
        self._key = key
        return key

    def stage_batch(self, batch: dict) -> tuple:
        """Materialise the batch on the device into per-shape static tensors; returns the graph key."""
        if "image" in batch:
            return self._stage_image_batch(batch)
        if "latents" in batch:
            return self._stage_cached_batch(batch)
        w, h, seeds = batch["width"], batch["height"], batch["seeds"]
        w, h = (int(w[0]) if isinstance(w, (list, tuple)) else int(w)), (int(h[0]) if isinstance(h, (list, tuple)) else int(h))
        key = (w, h, len(seeds))
        lat = make_latents(w, h, len(seeds), self.latent_channels, self.accelerator.device, seed=int(seeds[0]))
        cond = self.make_conditions(len(seeds), w, h, seed=int(seeds[0]))
        if key not in self._static:
            self._static[key] = (lat, cond)
        else:
            s_lat, s_cond = self._static[key]
            s_lat.copy_(lat)
            for k, v in cond.items():
                s_cond[k].copy_(v)
        self._key = key
        return key

    def train_step_static(self) -> torch.Tensor:
        lat, cond = self._static[self._key]
        loss = self.denoise_loss(lat, cond)
        self.log("train/loss", loss, on_step=True, on_epoch=True)
        return loss

    def train_step(self, batch: dict) -> torch.Tensor:
        self.stage_batch(batch)
        return self.train_step_static()

    def get_state_dict_to_save(self) -> dict[str, torch.Tensor]:
        if not self._is_peft:
            return self.model.state_dict()
        return {self.convert_adapter_key(k): v for k, v in get_adapter_parameters(self.model).items()}

    def convert_adapter_key(self, key: str) -> str:
        return key
