from gpt_amd.ddpm import SpacedDiffusion, space_timesteps  # noqa: F401
