"""`from Gpt.diffusion import create_diffusion` (main.py:18, playground.py:8)."""
from gpt_amd.ddpm import create_diffusion  # noqa: F401
from . import gaussian_diffusion, respace, timestep_sampler  # noqa: F401
