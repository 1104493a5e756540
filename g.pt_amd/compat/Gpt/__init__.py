"""Import shim: put g.pt_amd/compat on PYTHONPATH (ahead of the reference checkout) and the reference's callers
(main.py:18-26, playground.py:7-14, Gpt/vis.py:10-12) resolve their `Gpt.*` hot-path imports to the HIP path.
Only the modules of the hot path exist here; everything else of the reference's `Gpt` package stays the reference's."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
