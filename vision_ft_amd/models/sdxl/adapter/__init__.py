"""SDXL variants built on the substitution hooks of the denoiser (reference: src/models/sdxl/adapter/)."""
