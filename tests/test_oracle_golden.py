"""The oracle (CPU restatement) against the golden vectors produced by the reference itself
(tools/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import denoiser as od
from oracle import diffusion as odf
from oracle import permute as op
from tests.cases import CASES, block_case, make_inputs, slim, step_noise


def plan_of(case, predict_xstart=True):
    c = CASES[case]
    return od.Plan(c["parameter_sizes"], 128, 14, True, c["n_head"], c["n_embd"], c["n_layer"],
                   c["chunk_size"], c["split_policy"], predict_xstart)


def test_g1_splits(golden):
    g = golden("g1_splits")
    tasks = {"mnist": ([7840, 10, 100, 10], 1000), "cifar": ([432, 4608, 320, 10], 576)}
    for tname, (sizes, chunk) in tasks.items():
        nested = [sizes, sizes, [257], [257], [257], [257]]
        for pol in ("chunk", "layer_by_layer", "chunk_within_layer", "chunk_within_input"):
            assert od.build_splits(nested, pol, chunk) == g[f"{tname}_{pol}_in"].tolist()
            assert od.build_splits([sizes], pol, chunk) == g[f"{tname}_{pol}_out"].tolist()
    p = plan_of("mnist")
    assert p.T == 26 and p.D == 7960 and len(p.output_splits) == 11
    p = plan_of("cifar")
    assert p.T == 26 and p.D == 5370 and len(p.output_splits) == 11


def test_g2_freq(golden):
    g = golden("g2_freq")
    f = od.frequencies(128, 14)
    assert np.array_equal(f.numpy(), g["frequencies"])
    assert np.array_equal(od.freq_embed(torch.from_numpy(g["t"]), f).numpy(), g["t_emb"])
    assert np.array_equal(od.freq_embed(torch.from_numpy(g["l32"]), f).numpy(), g["l32_emb"])
    assert np.array_equal(od.freq_embed(torch.from_numpy(g["l64"]), f).numpy(), g["l64_emb"])
    assert np.array_equal(od.freq_embed(torch.from_numpy(g["l32"]).view(-1, 1), f).numpy(), g["l32_col_emb"])


@pytest.mark.parametrize("case", ["mnist_small", "cifar_small", "cartpole_small"])
def test_g3_forward(golden, case):
    g = golden(f"g3_forward_{case}")
    plan = plan_of(case)
    sd = od.random_state_dict(plan, seed=0)
    chk = [float(sd["decoder.blocks.0.attn.key.weight"].double().sum()), float(sd["decoder.pos_emb"].double().sum()),
           float(sd["decoder.output_parameter_projections.0.0.weight"].double().sum())]
    assert np.allclose(chk, g["w_checksum"], rtol=0, atol=1e-9), "seeded weights drifted"
    x, t, lt, lp, xp = make_inputs(case, B=4, seed=11)
    taps = {}
    out = od.forward(sd, plan, x, t, lt, lp, xp, taps=taps)
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(taps["block0"].numpy(), g["block0"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(taps["ln_f"].numpy(), g["ln_f"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose((taps["embed"] - sd["decoder.pos_emb"]).numpy(), g["ln_in"], rtol=1e-5, atol=1e-5)
    out2 = od.forward(sd, plan, x, t, lt.view(-1, 1).double(), lp.view(-1, 1).double(), xp)
    np.testing.assert_allclose(out2.numpy(), g["out_col64"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name,d,H", [("mnist", 1536, 16), ("cifar", 2048, 16)])
def test_g4_block(golden, name, d, H):
    g = golden("g4_block")
    sd, xin = block_case(d)
    y = od.block({"b." + k: v for k, v in sd.items()}, "b.", xin, H)
    np.testing.assert_allclose(y.numpy(), g[f"{name}_y"], rtol=1e-5, atol=1e-5)


def test_g5_tables(golden):
    g = golden("g5_tables")
    for tag, resp in (("full", ""), ("r50", "50"), ("ddim25", "ddim25")):
        tb = odf.Tables(1000, resp)
        assert tb.timestep_map == g[f"{tag}_timestep_map"].tolist()
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                  "posterior_mean_coef2"):
            assert np.array_equal(getattr(tb, k), g[f"{tag}_{k}"]), (tag, k)
    assert odf.Tables(1000).betas[0] == 9.999999999998899e-05


@pytest.mark.parametrize("case,tag", [("mnist_small", "x0"), ("mnist_small", "eps"), ("cifar_small", "x0")])
def test_g6_training_loss(golden, case, tag):
    g = golden(f"g6_training_{case}_{tag}")
    px = tag == "x0"
    plan = plan_of(case, px)
    sd = od.random_state_dict(plan, seed=0)
    _, _, lt, lp, xp = make_inputs(case, B=6, seed=21)
    tb = odf.Tables(1000)
    x_start, t, noise = (torch.from_numpy(g[k]) for k in ("x_start", "t", "noise"))
    params = {k: v.clone().requires_grad_(v.is_floating_point() and k != "scalar_embedder.frequencies") for k, v in sd.items()}
    fn = lambda x, tt: od.forward(params, plan, x, tt, lt.double(), lp.double(), xp)
    mse, out, x_t = odf.training_mse(tb, fn, x_start, t, noise, px)
    np.testing.assert_allclose(x_t.detach().numpy(), g["x_t"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(mse.detach().numpy(), g["mse"], rtol=1e-5, atol=1e-7)
    mse.mean().backward()
    for k in g.files:
        if k.startswith("grad::"):
            np.testing.assert_allclose(slim(params[k[6:]].grad.numpy()), g[k], rtol=1e-4, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("case,resp,tag,px,clip", [
    ("mnist_small", "", "full", True, False), ("mnist_small", "50", "r50", True, False),
    ("cifar_small", "50", "r50", True, False), ("mnist_small", "50", "r50_eps_clip", False, True)])
def test_g7_sampling(golden, case, resp, tag, px, clip):
    g = golden(f"g7_sampling_{case}_{tag}")
    plan = plan_of(case, px)
    sd = od.random_state_dict(plan, seed=0)
    tb = odf.Tables(1000, resp)
    assert tb.T == int(g["num_steps"])
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    with torch.no_grad():
        fn = lambda x, tt: od.forward(sd, plan, x, tt, lt, lp, xp)
        final = odf.p_sample_loop(tb, fn, step_noise(1234, -1, B, plan.D), lambda i: step_noise(1234, i, B, plan.D), px, clip)
    np.testing.assert_allclose(final.numpy(), g["sample_at_0"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case,resp,tag,eta,px,clip", [
    ("mnist_small", "ddim25", "ddim25_eta0", 0.0, True, False), ("mnist_small", "50", "r50_eta05", 0.5, True, False),
    ("cifar_small", "ddim25", "ddim25_eta1_eps_clip", 1.0, False, True)])
def test_g11_ddim(golden, case, resp, tag, eta, px, clip):
    """ddim_sample_loop against the reference's trajectory, injected noise."""
    g = golden(f"g11_ddim_{case}_{tag}")
    plan = plan_of(case, px)
    sd = od.random_state_dict(plan, seed=0)
    tb = odf.Tables(1000, resp)
    assert tb.T == int(g["num_steps"]) and float(g["eta"]) == eta
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    with torch.no_grad():
        fn = lambda x, tt: od.forward(sd, plan, x, tt, lt, lp, xp)
        final = odf.ddim_sample_loop(tb, fn, step_noise(1234, -1, B, plan.D), lambda i: step_noise(1234, i, B, plan.D), eta, px, clip)
    np.testing.assert_allclose(final.numpy(), g["sample_at_0"], rtol=1e-4, atol=1e-4)


def test_g8_permute(golden):
    g = golden("g8_permute")
    for seed in range(16):
        assert np.array_equal(op.mlp_index(*op.draw_mlp(seed)), g["mlp"][seed].astype(np.int64))
        assert np.array_equal(op.cnn_index(*op.draw_cnn(seed)), g["cnn"][seed].astype(np.int64))
        assert sorted(op.mlp_index(*op.draw_mlp(seed)).tolist()) == list(range(7960))


def test_g15_actor_permute(golden):
    """The Cartpole policy's maps (data_gen/train_rl.py:51-74 run on the reference's VisActorCritic layout, 16 seeds)."""
    g = golden("g15_actor_permute")
    hid = tuple(int(h) for h in g["hidden"])
    assert hid == (32, 32) and g["sizes"].tolist() == [1, 128, 32, 1024, 32, 32, 1]
    for seed in range(16):
        idx = op.actor_index(op.draw_actor(seed, hid))
        assert np.array_equal(idx, g["actor"][seed].astype(np.int64))
        assert sorted(idx.tolist()) == list(range(1250))


def test_g10_mlp_eval(golden):
    """The oracle's per-net functional evaluation against the reference's MLP + moduleify + cross_entropy."""
    from oracle import evaluate as oe
    g = golden("g10_mlp_eval")
    loss, err = oe.mlp_eval(torch.from_numpy(g["nets"]), torch.from_numpy(g["imgs"]), torch.from_numpy(g["labels"]))
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(err.numpy(), g["err"], rtol=0, atol=1e-4)


def test_g12_cnn_eval(golden):
    """The oracle's functional ConvNet evaluation against the reference's ConvNet + moduleify + cross_entropy."""
    from oracle import evaluate as oe
    g = golden("g12_cnn_eval")
    loss, err = oe.cnn_eval(torch.from_numpy(g["nets"]), torch.from_numpy(g["imgs"]), torch.from_numpy(g["labels"]))
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(err.numpy(), g["err"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("case", ["mnist", "cifar"])
@pytest.mark.parametrize("split", ["train", "test"])
def test_g13_pair_dataset(golden, tmp_path, case, split):
    """The oracle's checkpoint-pair dataset against the reference's ParameterDataset over the same synthetic runs: split,
    metadata reductions, the per-item RNG order (spacing, first index, permutation), normalisation and augmentation."""
    from oracle import data as odata
    from tests.cases import PAIR_CASES, make_pair_dataset
    g, c = golden("g13_pairs"), PAIR_CASES[case]
    dbs = make_pair_dataset(str(tmp_path), case)
    ds = odata.RefPairDataset(str(tmp_path), lambda p: dbs[p], split=split, num_test_runs=c["num_test_runs"],
                              train_metric=c["train_metric"], openai_coeff=c["openai_coeff"], target_epoch_size=1000,
                              augment="mlp" if case == "mnist" else "cnn")
    tag, m = f"{case}_{split}", c["train_metric"]
    assert list(g[f"{tag}_runs"]) == ds.runs and list(g[f"{tag}_sizes"]) == ds.parameter_sizes
    np.testing.assert_array_equal(g[f"{tag}_scalars"], np.array([ds.min_val, ds.max_val, ds.optimal_test_loss,
                                                                 float(ds.epoch_size), *ds.get_range()]))
    torch.manual_seed(c["seed"] + 1)
    items = [ds[i] for i in c["items"]]
    assert [it["checkpoint_key_0"] for it in items] == list(g[f"{tag}_key0"])
    assert [it["checkpoint_key_1"] for it in items] == list(g[f"{tag}_key1"])
    assert [it["run_name"] for it in items] == list(g[f"{tag}_run"])
    np.testing.assert_array_equal([it["step_spacing"] for it in items], g[f"{tag}_spacing"])
    np.testing.assert_array_equal([it[f"{m}_0"] for it in items], g[f"{tag}_m0"])
    np.testing.assert_array_equal([it[f"{m}_1"] for it in items], g[f"{tag}_m1"])
    np.testing.assert_array_equal(np.stack([it["parameters_0"].numpy()[::7] for it in items]), g[f"{tag}_p0"])
    np.testing.assert_array_equal(np.stack([it["parameters_1"].numpy()[::7] for it in items]), g[f"{tag}_p1"])


def test_flat_db_round_trip(tmp_path):
    """Host logic of the product's storage back-end: write_flat_db -> FlatDb returns every value bit for bit."""
    from gpt_amd import data as gdata
    from tests.cases import make_pair_dataset
    dbs = make_pair_dataset(str(tmp_path / "src"), "cifar")
    for path, db in dbs.items():
        dst = str(tmp_path / "flat" / path.split("/")[-1])
        gdata.write_flat_db(dst, db.items())
        flat = gdata.FlatDb(dst)
        assert sorted(flat.keys()) == sorted(db.keys())
        for k, v in db.items():
            assert k in flat and torch.equal(flat[k], v)


@pytest.mark.parametrize("tag,respacing,px", [("full_x0", "", True), ("r50_eps", "50", False)])
def test_g14_diffusion_misc(golden, tag, respacing, px):
    """ddim_reverse_sample / q_mean_variance / q_posterior_mean_variance bit for bit, _prior_bpd to the row mean's rounding."""
    g = golden("g14_diffusion_misc")
    tb = odf.Tables(1000, respacing)
    x, x0, mo, t = (torch.from_numpy(g[k]) for k in ("x", "x0", "mo", f"{tag}_t"))
    rev, rx0 = odf.ddim_reverse_step(tb, mo, x, t, predict_xstart=px, clip=(tag == "r50_eps"))
    np.testing.assert_array_equal(rev.numpy(), g[f"{tag}_rev"])
    np.testing.assert_array_equal(rx0.numpy(), g[f"{tag}_rev_x0"])
    for got, want in zip(odf.q_mean_variance(tb, x0, t), ("qm", "qv", "qlv")):
        np.testing.assert_array_equal(got.numpy(), g[f"{tag}_{want}"])
    for got, want in zip(odf.q_posterior_mean_variance(tb, x0, x, t), ("pm", "pv", "plv")):
        np.testing.assert_array_equal(got.numpy(), g[f"{tag}_{want}"])
    np.testing.assert_allclose(odf.prior_bpd(tb, x0).numpy(), g[f"{tag}_bpd"], rtol=1e-6)
