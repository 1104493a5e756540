"""Import shim: put g.pt_amd/compat on PYTHONPATH (ahead of the reference checkout) and the reference's callers
(main.py:18-29, playground.py:7-14, Gpt/vis.py:10-12) resolve their `Gpt.*` hot-path imports to the HIP path.

Only the modules of the hot path exist here.  Everything else of the reference's `Gpt` package (`Gpt.data.*`, `Gpt.utils`,
`Gpt.vis`, `Gpt.meters`, `Gpt.tasks`, `Gpt.download`, `Gpt.latent_walk_helpers`) stays the reference's: this package's
`__path__` is extended with every other `Gpt/` directory on `sys.path`, so a submodule that is not found here is looked up
in the reference checkout (this directory comes first, so the hot-path modules always win)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(_HERE)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)


def _extend_path(path, here, *sub):
    """Appends <entry>/Gpt/<sub...> of every other sys.path entry to a package __path__ (what pkgutil.extend_path does for
    namespace-style packages, restricted to directories, evaluated once at import)."""
    seen = {os.path.realpath(p) for p in path}
    for entry in list(sys.path):
        cand = os.path.join(entry or os.getcwd(), "Gpt", *sub)
        real = os.path.realpath(cand)
        if os.path.isdir(cand) and real not in seen and real != os.path.realpath(here):
            path.append(cand)
            seen.add(real)
    return path


__path__ = _extend_path(list(__path__), _HERE)
