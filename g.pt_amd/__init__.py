"""MI355X-native G.pt denoiser hot path (importable as `gpt_amd`).

Public surface = the reference's names on this path:
  Gpt, GPT, FrequencyEmbedder                      (Gpt/models/transformer.py)
  create_diffusion, GaussianDiffusion, SpacedDiffusion, UniformSampler   (Gpt/diffusion/*)
  is_main_proc, scaled_all_reduce, get_rank, get_world_size, synchronize, all_gather, rank0_to_all  (Gpt/distributed.py)
"""
from .denoiser import Gpt, GPT, FrequencyEmbedder, Block, SelfAttention, build_splits  # noqa: F401
from .ddpm import (create_diffusion, GaussianDiffusion, SpacedDiffusion, UniformSampler, StaticThreshold,  # noqa: F401
                   ModelMeanType, ModelVarType, LossType, space_timesteps, get_named_beta_schedule,
                   create_named_schedule_sampler)
from .dp import (is_main_proc, scaled_all_reduce, get_rank, get_world_size, synchronize, all_gather,  # noqa: F401
                 rank0_to_all)
from . import permute  # noqa: F401
from .evaluate import (synth, create_thresholding_fn, mlp_test_losses, cnn_test_losses, one_step_prompting,  # noqa: F401,E402
                       recursive_prompting, aggregated_evolve, probe_G, batched_test_fn, batch_linspace, r_squared)
from .data import CheckpointStore, write_flat_db  # noqa: F401,E402
