"""`from Gpt.diffusion.timestep_sampler import UniformSampler` (main.py:19)."""
from gpt_amd.ddpm import ScheduleSampler, UniformSampler, create_named_schedule_sampler  # noqa: F401
