"""`Gpt.models`: `transformer` is the HIP host; anything else falls through to the reference checkout (Gpt/__init__.py)."""
import os

from .. import _extend_path

__path__ = _extend_path(list(__path__), os.path.dirname(os.path.abspath(__file__)), "models")
