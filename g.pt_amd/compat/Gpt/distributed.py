"""`from Gpt.distributed import scaled_all_reduce, get_rank, get_world_size, is_main_proc, synchronize` (main.py:22,26)."""
from gpt_amd.dp import (all_gather, get_rank, get_world_size, is_main_proc, rank0_to_all, scaled_all_reduce,  # noqa: F401
                        synchronize)
