"""`from Gpt.diffusion import create_diffusion` (main.py:18, playground.py:9).  Submodules that are not part of the hot path
(`Gpt.diffusion.diffusion_utils`, ...) fall through to the reference checkout, see Gpt/__init__.py."""
import os

from .. import _extend_path

__path__ = _extend_path(list(__path__), os.path.dirname(os.path.abspath(__file__)), "diffusion")

from gpt_amd.ddpm import create_diffusion  # noqa: F401,E402
from . import gaussian_diffusion, respace, timestep_sampler  # noqa: F401,E402
