from .config import DenoiserConfig, Lumina2Config
from .denoiser import Denoiser

__all__ = ["DenoiserConfig", "Lumina2Config", "Denoiser"]
