from .config import DenoiserConfig, Lumina2Config
from .denoiser import Denoiser
from .scheduler import Scheduler

__all__ = ["DenoiserConfig", "Lumina2Config", "Denoiser", "Scheduler"]
