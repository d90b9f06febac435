"""Lumina-Image-2.0 configuration schema (reference: src/models/lumina2/config.py:9-41)."""
import torch
from pydantic import BaseModel

from ...utils.dtype import str_to_dtype


class DenoiserConfig(BaseModel):  # NextDiT_2B_GQA_patch2_Adaln_Refiner
    in_channels: int = 16
    out_channels: int = 16
    hidden_dim: int = 2304
    caption_dim: int = 2304
    timestep_embed_dim: int = 256
    norm_eps: float = 1e-5
    depth: int = 26
    num_heads: int = 24
    num_kv_heads: int = 8
    refiner_depth: int = 2
    multiple_of: int = 256
    axes_dims: list[int] = [32, 32, 32]
    axes_lens: list[int] = [300, 512, 512]
    theta: int = 10_000
    qkv_bias: bool = True
    patch_size: int = 2
    vae_channels: int = 16


class Lumina2Config(BaseModel):
    checkpoint_path: str
    dtype: str = "bfloat16"
    denoiser: DenoiserConfig = DenoiserConfig()

    def get_dtype(self) -> torch.dtype:
        return str_to_dtype(self.dtype)
