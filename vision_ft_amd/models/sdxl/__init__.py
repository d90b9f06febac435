from .config import DenoiserConfig, SDXLConfig
from .denoiser import Denoiser

__all__ = ["DenoiserConfig", "SDXLConfig", "Denoiser"]
