"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header declares,
and the host logic (token plans, schedule tables, module/state_dict layout, optimizer groups,
permutation draws) matches the golden vectors produced by the reference.  No GPU, no compute calls."""
import os
import re

import numpy as np
import pytest
import torch

import gpt_amd
from gpt_amd import _lib as L
from tests.cases import gpt_kwargs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "gpt_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(gpt_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = L.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/gpt_hip.h but not exported"
    assert declared == set(L.exported_symbols()), declared ^ set(L.exported_symbols())
    assert lib.gpt_abi_version() == 1


def test_token_plans_match_reference(golden):
    g = golden("g1_splits")
    tasks = {"mnist": ([7840, 10, 100, 10], 1000), "cifar": ([432, 4608, 320, 10], 576)}
    for tname, (sizes, chunk) in tasks.items():
        nested = [sizes, sizes, [257], [257], [257], [257]]
        for pol in ("chunk", "layer_by_layer", "chunk_within_layer", "chunk_within_input"):
            assert gpt_amd.GPT.build_splits(nested, pol, chunk) == g[f"{tname}_{pol}_in"].tolist()
            assert gpt_amd.GPT.build_splits([sizes], pol, chunk) == g[f"{tname}_{pol}_out"].tolist()
    with pytest.raises(NotImplementedError):
        gpt_amd.GPT.build_splits([1], "nope", 1)


def test_schedule_tables_match_reference(golden):
    g = golden("g5_tables")
    for tag, resp in (("full", ""), ("r50", "50"), ("ddim25", "ddim25")):
        d = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                     timestep_respacing=resp)
        assert d.timestep_map == g[f"{tag}_timestep_map"].tolist()
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            assert np.array_equal(getattr(d, k), g[f"{tag}_{k}"]), (tag, k)
    d = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=False)
    assert d.model_mean_type == gpt_amd.ModelMeanType.EPSILON and d.model_var_type == gpt_amd.ModelVarType.FIXED_LARGE
    with pytest.raises(NotImplementedError):
        gpt_amd.create_diffusion(learn_sigma=True).coef_table("cpu")


def test_state_dict_layout_and_init(golden):
    m = gpt_amd.Gpt(**gpt_kwargs("mnist_small"))
    sd = m.state_dict()
    from oracle import denoiser as od
    from tests.test_oracle_golden import plan_of
    ref = od.random_state_dict(plan_of("mnist_small"))
    assert list(sd.keys()) == list(ref.keys())
    assert all(sd[k].shape == ref[k].shape for k in ref)
    # fresh model: identity decoder and zero position embedding (transformer.py:101, 363-367)
    assert float(sd["decoder.pos_emb"].abs().max()) == 0
    assert all(float(sd[f"decoder.output_parameter_projections.{j}.0.weight"].abs().max()) == 0 for j in range(11))
    assert abs(float(sd["decoder.blocks.0.mlp.0.weight"].std()) - 0.02) < 2e-3
    m.load_state_dict(ref)
    g9 = golden("g9_trainstep_mnist_small")
    opt = m.configure_optimizers(lr=4e-4, wd=0.1, betas=(0.9, 0.999))
    assert len(opt.param_groups[0]["params"]) == int(g9["n_decay"])
    assert len(opt.param_groups[1]["params"]) == int(g9["n_nodecay"])
    assert opt.param_groups[0]["weight_decay"] == 0.1 and opt.param_groups[1]["weight_decay"] == 0.0
    from copy import deepcopy
    ema = deepcopy(m)
    assert ema._engines == {} and list(ema.state_dict()) == list(sd)
    full = gpt_amd.Gpt(**{**gpt_kwargs("mnist"), "n_layer": 1})
    assert len(full.state_dict()) == 235 - 11 * 16        # 12-layer model: 235 entries


def test_no_cpu_fallback():
    m = gpt_amd.Gpt(**gpt_kwargs("mnist_small")).eval()
    x = torch.zeros(2, 7960)
    with pytest.raises(L.GptHipError), torch.no_grad():
        m(x, torch.zeros(2, dtype=torch.long), torch.zeros(2), loss_prev=torch.zeros(2), x_prev=x)
    with pytest.raises(NotImplementedError):
        gpt_amd.Gpt(**{**gpt_kwargs("mnist_small"), "encoder_depth": 2})


def test_permutation_draws_match_reference_rng():
    from oracle import permute as op
    P = gpt_amd.permute
    perms = P.draw_permutations(P.CNN_ARCH, [3, 4])
    p1, p2 = op.draw_cnn(3)
    assert perms.dtype == torch.int32 and perms[0].tolist() == p1.tolist() + p2.tolist()
    assert P.arch_dim(P.MLP_ARCH) == 7960 and P.arch_dim(P.CNN_ARCH) == 5370


def test_uniform_sampler():
    d = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True)
    np.random.seed(5)
    t, w = gpt_amd.UniformSampler(d).sample(16, "cpu")
    np.random.seed(5)
    assert t.tolist() == np.random.choice(1000, size=(16,), p=np.ones(1000) / 1000).tolist()
    assert t.dtype == torch.int64 and w.dtype == torch.float32 and torch.all(w == 1)


def test_reference_import_paths_resolve_to_the_hip_host():
    """The module paths the reference's callers import (main.py:18-26, playground.py:7-14) exist in the shim."""
    import importlib
    import os
    import sys
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "g.pt_amd", "compat")
    saved = {k: v for k, v in sys.modules.items() if k == "Gpt" or k.startswith("Gpt.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, compat)
    try:
        import gpt_amd
        tr = importlib.import_module("Gpt.models.transformer")
        df = importlib.import_module("Gpt.diffusion")
        tsamp = importlib.import_module("Gpt.diffusion.timestep_sampler")
        dd = importlib.import_module("Gpt.distributed")
        assert tr.Gpt is gpt_amd.Gpt and df.create_diffusion is gpt_amd.create_diffusion
        assert tsamp.UniformSampler is gpt_amd.UniformSampler
        for name in ("scaled_all_reduce", "get_rank", "get_world_size", "is_main_proc", "synchronize", "all_gather", "rank0_to_all"):
            assert callable(getattr(dd, name))
        gd = importlib.import_module("Gpt.diffusion.gaussian_diffusion")
        assert hasattr(gd, "GaussianDiffusion") and hasattr(gd, "ModelMeanType")
    finally:
        sys.path.remove(compat)
        for k in [k for k in sys.modules if k == "Gpt" or k.startswith("Gpt.")]:
            del sys.modules[k]
        sys.modules.update(saved)
