from .embedding import get_timestep_embedding
from .sampling import sample_timestep, sigmoid_randn, uniform_rand, uniform_randint

__all__ = ["get_timestep_embedding", "sample_timestep", "sigmoid_randn", "uniform_rand", "uniform_randint"]
