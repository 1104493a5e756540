from gpt_amd.ddpm import (GaussianDiffusion, LossType, ModelMeanType, ModelVarType, get_beta_schedule,  # noqa: F401
                          get_named_beta_schedule)
