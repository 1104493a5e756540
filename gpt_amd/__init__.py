"""Import alias: the package lives in the directory `g.pt_amd/` (not a valid Python identifier), this
shim makes it importable as `gpt_amd` (`import gpt_amd`, `from gpt_amd.ddpm import create_diffusion`)."""
import os

__path__ = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "g.pt_amd")]
with open(os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), os.path.join(__path__[0], "__init__.py"), "exec"))
