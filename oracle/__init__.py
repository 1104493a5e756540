"""CPU oracle for the G.pt denoiser hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``g.pt_amd/`` (the product) imports this
package.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / reported baseline.

The oracle is a functional (state_dict in, tensor out) restatement, in plain
PyTorch-CPU fp32 + numpy float64, of the arithmetic in the reference's
``Gpt/models/transformer.py``, ``Gpt/diffusion/*`` and ``Gpt/data/augment.py``.
It is pinned against outputs of the reference itself, imported in the build
container by ``tools/make_golden.py`` (fixtures committed under ``tests/golden/``).
The reference has no tests or golden vectors of its own.
"""
