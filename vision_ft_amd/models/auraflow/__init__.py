from .config import AuraFlowConig, DenoiserConfig
from .denoiser import Denoiser

__all__ = ["AuraFlowConig", "DenoiserConfig", "Denoiser"]
