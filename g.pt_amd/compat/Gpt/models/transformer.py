"""`from Gpt.models.transformer import Gpt` (main.py:23, playground.py:10)."""
from gpt_amd.denoiser import GPT, Block, FrequencyEmbedder, Gpt, SelfAttention  # noqa: F401
