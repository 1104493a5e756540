"""Shared test-case definitions: model shapes, seeded synthetic inputs and the per-step noise
convention.  Used by tools/make_golden.py (which runs the reference) and by the tests (which run
the oracle and the HIP path), so that all three see identical inputs.  Input distributions follow
SURVEY.md section 8(d)."""
import torch

from gpt_amd.configs import (CASES, CIFAR_NAMES, CIFAR_SIZES, MNIST_NAMES, MNIST_SIZES,  # noqa: F401
                             gpt_kwargs)


def make_inputs(case, B, seed):
    """(x, t, loss_target, loss_prev, x_prev), all CPU; x/x_prev fp32 (B,D), t int64, losses fp32 (B,)."""
    c = CASES[case]
    D = sum(c["parameter_sizes"])
    g = torch.Generator().manual_seed(seed)
    x_prev = torch.randn(B, D, generator=g) * c["w_std"]
    x = torch.randn(B, D, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    t[0] = 999
    if B > 1:
        t[1] = 0
    loss_prev = c["loss_lo"] + (c["loss_hi"] - c["loss_lo"]) * torch.rand(B, generator=g)
    loss_target = torch.full((B,), float(c["loss_target"]))
    return x, t, loss_target, loss_prev, x_prev


def step_noise(base_seed, i, B, D):
    """N(0,1) noise drawn at diffusion index i (i = -1: the start noise); seed = base_seed + 1 + i."""
    g = torch.Generator().manual_seed(base_seed + 1 + i)
    return torch.randn(B, D, generator=g)


def slim(a, stride=37):
    """Same strided sampling of large tensors as tools/make_golden.py uses when storing them."""
    import numpy as np
    a = np.asarray(a)
    return a.reshape(-1)[::stride].copy() if a.size > 20000 else a


def block_case(d, seed=5):
    """Weights and input of the full-width single-Block fixture (g4_block), in the parameter order
    ln1.w, ln1.b, ln2.w, ln2.b, attn.{key,query,value,proj}.{w,b}, mlp.0.{w,b}, mlp.2.{w,b}."""
    g = torch.Generator().manual_seed(seed)
    names = ["ln1.weight", "ln1.bias", "ln2.weight", "ln2.bias"]
    shapes = [(d,)] * 4
    for nm in ("key", "query", "value", "proj"):
        names += [f"attn.{nm}.weight", f"attn.{nm}.bias"]
        shapes += [(d, d), (d,)]
    names += ["mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias"]
    shapes += [(4 * d, d), (4 * d,), (d, 4 * d), (d,)]
    sd = {}
    for n, s in zip(names, shapes):
        if len(s) == 2:
            sd[n] = torch.randn(s, generator=g) * 0.02
        else:
            sd[n] = torch.randn(s, generator=g) * 0.1 + (1.0 if n in ("ln1.weight", "ln2.weight") else 0.0)
    xin = torch.randn(2, 26, d, generator=g)
    return sd, xin


# ------------------------------------------------------------------ synthetic checkpoint datasets (fixture g13)
PAIR_CASES = {
    "mnist": dict(dataset_name="mnist_loss", D=7960, sizes=[7840, 10, 100, 10], train_metric="avg_test_loss",
                  dirs=["r00_r01_r02", "r03", "r04", "r05", "r06"], n_ckpt=[7, 5, 9, 6, 8, 4, 6], openai_coeff=4.185,
                  num_test_runs=2, augment=True, items=[0, 5, 3, 12, 7, 1, 4, 9], seed=123),
    "cifar": dict(dataset_name="cifar10_error", D=5370, sizes=[432, 4608, 320, 10], train_metric="avg_test_err",
                  dirs=["a0", "a1_a2", "a3"], n_ckpt=[6, 6, 5, 7], openai_coeff=1.646,
                  num_test_runs=1, augment=True, items=[2, 0, 1, 5, 4, 3], seed=321),
}


def make_pair_dataset(root, case):
    """Writes runs.json files under root/<dir>/ and returns {dir path: {key: fp32 tensor}} (values regenerated from seeds:
    the fixture holds no parameter data, only what the reference returned for it)."""
    import json
    import os
    c = PAIR_CASES[case]
    g = torch.Generator().manual_seed(c["seed"])
    dbs, r = {}, 0
    for d in c["dirs"]:
        path = os.path.join(root, d)
        os.makedirs(path, exist_ok=True)
        db, fused = {}, {}
        for run in d.split("_"):
            n = c["n_ckpt"][r]
            ck, series, lo, hi = {}, [], 1e9, -1e9
            for i in range(n):
                key = f"{run}/ep{i:03d}_it{i * 50:05d}.pt"
                v = torch.randn(c["D"], generator=g) * 0.25
                db[key] = v
                db[key + "_arch"] = torch.tensor(c["sizes"], dtype=torch.float32)
                val = float(torch.rand((), generator=g).double() * 2.0 + 0.05)
                ck[key] = {c["train_metric"]: val}
                series.append(val)
                lo, hi = min(lo, float(v.min())), max(hi, float(v.max()))
            opt_key = ("optimal_" + c["train_metric"]).replace("_avg", "")
            fused[run] = {"checkpoints": ck, "metadata": {c["train_metric"]: series, opt_key: min(series),
                                                          "min_parameter_val": lo, "max_parameter_val": hi}}
            r += 1
        with open(os.path.join(path, "runs.json"), "w") as f:
            json.dump(fused if len(fused) > 1 else next(iter(fused.values())), f)
        dbs[path] = db
    return dbs
