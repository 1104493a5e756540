"""GPU parity tests of the evaluation chains of BASELINE.json configs[2] (Gpt/vis.py:325-395): one-step prompting
(repeat_interleave(dvo_steps) -> one sampling loop over N * dvo_steps rows) and recursive prompting (chained loops, each sample
fed back as x_prev) on the HIP path against the oracle running the same chain with the same injected noise; and the full-size
CIFAR-10 batch (B = 1280) through size-independent properties."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import denoiser as od          # noqa: E402
from oracle import diffusion as odf        # noqa: E402
from oracle import evaluate as oe          # noqa: E402
from tests.cases import make_inputs, step_noise  # noqa: E402
from tests.test_hip_parity import DEV, build, rel_err  # noqa: E402

import gpt_amd                              # noqa: E402
from gpt_amd import evaluate as ev          # noqa: E402

CASES = {   # case -> (scorer kind, fixture with images / labels, openai_coeff, Gpt/tasks.py prompts of the *_loss task)
    "cifar_small": ("cnn", "g12_cnn_eval", 1.646, dict(minimize=True, best_prompt=1.2, recursive_prompt=1.4)),
    "mnist_small": ("mlp", "g10_mlp_eval", 4.185, dict(minimize=True, best_prompt=0.0, recursive_prompt=0.0)),
}


class _Noise:
    """The same injected noise for both paths: call c of synth draws its start / per-step noise from seed base + 1000 c."""

    def __init__(self, base, n_steps, D):
        self.base, self.n, self.D, self.calls = base, n_steps, D, 0

    def next(self, B):
        seed = self.base + 1000 * self.calls
        self.calls += 1
        return step_noise(seed, -1, B, self.D), torch.stack([step_noise(seed, i, B, self.D) for i in range(self.n)])


@pytest.mark.parametrize("case", ["cifar_small", "mnist_small"])
def test_prompting_chains_vs_oracle(golden, case):
    kind, fixture, coeff, meta = CASES[case]
    m, plan, sd = build(case, "fp32", seed=14)
    resp = "20"
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing=resp)
    tb = odf.Tables(1000, resp)
    fx = golden(fixture)
    imgs, labels = torch.from_numpy(fx["imgs"]), torch.from_numpy(fx["labels"])
    N, dvo = 3, 4
    nets = make_inputs(case, B=N, seed=90)[4] * 0.5              # normalised parameter vectors (x openai_coeff scale)

    # ---- HIP path
    test_fn = ev.batched_test_fn(kind, imgs.to(DEV), labels.to(DEV), metric="loss", unnormalize=coeff)
    nz = _Noise(4000, diff.num_timesteps, plan.D)

    def synth_fn(diffusion, G, want, cur, w_prev):
        start, steps = nz.next(w_prev.shape[0])
        return gpt_amd.synth(diffusion, G, want, cur, w_prev, noise=start.to(DEV), step_noise=steps)

    d1, o1 = ev.one_step_prompting(nets.to(DEV), synth_fn, diff, m, test_fn, meta, dvo_steps=dvo)
    d2, o2 = ev.recursive_prompting(nets.to(DEV), synth_fn, diff, m, test_fn, meta, max_steps=2)
    assert nz.calls == 3

    # ---- oracle, same chain, same noise
    scorer = oe.cnn_eval if kind == "cnn" else oe.mlp_eval
    o_test = lambda v: scorer(v / coeff, imgs, labels)[0]                            # noqa: E731
    onz = _Noise(4000, tb.T, plan.D)

    def o_synth(want, cur, w_prev):
        start, steps = onz.next(w_prev.shape[0])
        with torch.no_grad():
            return odf.p_sample_loop(tb, lambda x, t: od.forward(sd, plan, x, t, want, cur, w_prev), start, lambda i: steps[i])

    rd1, ro1, _ = oe.one_step_prompting(nets, o_synth, o_test, meta["best_prompt"], dvo_steps=dvo)
    rd2, ro2, _ = oe.recursive_prompting(nets, o_synth, o_test, meta["recursive_prompt"], max_steps=2)

    assert d1.shape == (N, dvo) and o1.shape == (N, dvo + 1) and d2.shape == (N, 2) and o2.shape == (N, 3)
    assert rel_err(d1.cpu(), rd1) < 1e-4                          # prompts are spaced from each path's own base loss
    assert torch.equal(d2.cpu(), rd2)
    assert rel_err(o1.cpu(), ro1) < 1e-3, (o1.cpu(), ro1)
    assert rel_err(o2.cpu(), ro2) < 1e-3, (o2.cpu(), ro2)
    # the recursion really feeds back: loop 2 starts from loop 1's sample and its loss
    assert float((ro2[:, 1] - ro2[:, 0]).abs().max()) > 0


def test_chain_through_aggregated_evolve_equals_direct_sampling():
    """aggregated_evolve (mini-batches of nets, vis.py:293-322) around one_step_prompting gives exactly the rows that one direct
    p_sample_loop over the repeat_interleave'd inputs gives (Philox noise is row-local only with injected noise: inject it)."""
    m, plan, sd = build("cifar_small", "bf16", seed=15)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="5")
    N, dvo, D = 6, 5, plan.D
    nets = (make_inputs("cifar_small", B=N, seed=91)[4] * 0.5).to(DEV)
    g = torch.Generator().manual_seed(5)
    imgs, labels = torch.randn(32, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 10, (32,), generator=g).to(DEV)
    test_fn = ev.batched_test_fn("cnn", imgs, labels, metric="error", unnormalize=1.646)
    meta = ev.TASK_PROMPTS["cifar10_error"]
    start = step_noise(7, -1, N * dvo, D).to(DEV)
    steps = torch.stack([step_noise(7, i, N * dvo, D) for i in range(diff.num_timesteps)]).to(DEV)
    seen = []

    def synth_fn(diffusion, G, want, cur, w_prev):
        lo = len(seen) * 2 * dvo                                   # mini-batches of 2 nets = 2 * dvo rows each
        seen.append(w_prev.shape[0])
        rows = slice(lo, lo + w_prev.shape[0])
        return gpt_amd.synth(diffusion, G, want, cur, w_prev, noise=start[rows], step_noise=steps[:, rows].contiguous())

    desired, observed = ev.aggregated_evolve(ev.one_step_prompting, nets, 2, synth_fn, diff, m, test_fn, meta, dvo_steps=dvo)
    assert seen == [2 * dvo] * 3 and desired.shape == (N, dvo) and observed.shape == (N, dvo + 1)
    base = test_fn(nets)
    want = ev.batch_linspace(torch.full((N,), 35.0, device=DEV), base, dvo)
    assert torch.equal(desired, want.cpu())
    direct = diff.p_sample_loop(m, (N * dvo, D), noise=start, clip_denoised=False, step_noise=steps,
                                model_kwargs=dict(loss_target=want.flatten(), loss_prev=base.repeat_interleave(dvo),
                                                  x_prev=nets.repeat_interleave(dvo, dim=0)))
    # bf16 rows computed inside a batch of 10 vs a batch of 30 take the same kernels here (no K spans at these sizes)
    assert rel_err(observed[:, 1:].reshape(-1), test_fn(direct).cpu()) < 2e-2
    assert torch.equal(observed[:, 0], base.cpu())


def test_cifar_full_size_b1280_short_loop_properties():
    """configs[2] at its real size: CIFAR-10 G.pt shapes (d = 2048, 12 layers), B = 64 nets x 20 prompts = 1280 rows, bf16.
    A short respaced loop: (a) the fused whole-loop call equals the per-step path fed the same noise, bit for bit; (b) rows are
    independent: a 40-row slice sampled alone agrees with the same rows inside the batch to bf16 rounding (other tile shapes);
    (c) one_step_prompting over 64 nets produces exactly that batch."""
    m, plan, sd = build("cifar", "bf16", seed=3)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="3")
    N, dvo, D = 64, 20, plan.D
    B, n = N * dvo, diff.num_timesteps
    nets = (make_inputs("cifar", B=N, seed=92)[4] * 0.5).to(DEV)
    g = torch.Generator().manual_seed(6)
    imgs, labels = torch.randn(64, 3, 32, 32, generator=g).to(DEV), torch.randint(0, 10, (64,), generator=g).to(DEV)
    test_fn = ev.batched_test_fn("cnn", imgs, labels, metric="error", unnormalize=1.646)
    gen = torch.Generator(device=DEV).manual_seed(8)
    start = torch.randn(B, D, device=DEV, generator=gen)
    steps = torch.randn(n, B, D, device=DEV, generator=gen)
    captured = {}

    def synth_fn(diffusion, G, want, cur, w_prev):
        captured.update(want=want, cur=cur, w_prev=w_prev)
        return gpt_amd.synth(diffusion, G, want, cur, w_prev, noise=start, step_noise=steps)

    desired, observed = ev.one_step_prompting(nets, synth_fn, diff, m, test_fn, ev.TASK_PROMPTS["cifar10_error"], dvo_steps=dvo)
    assert captured["w_prev"].shape == (B, D) and desired.shape == (N, dvo) and observed.shape == (N, dvo + 1)
    assert bool(torch.isfinite(observed).all())
    assert torch.equal(captured["w_prev"][dvo:2 * dvo], nets[1:2].expand(dvo, D))          # repeat_interleave order (vis.py:384-386)
    kw = dict(loss_target=captured["want"], loss_prev=captured["cur"], x_prev=captured["w_prev"])
    fused = diff.p_sample_loop(m, (B, D), noise=start, clip_denoised=False, model_kwargs=kw, step_noise=steps)
    assert torch.equal(test_fn(fused).view(N, dvo).cpu(), observed[:, 1:].cpu())            # (c)
    x = start.clone()
    tmap = torch.tensor(diff.timestep_map, device=DEV)
    for i in reversed(range(n)):
        t = torch.full((B,), i, device=DEV, dtype=torch.long)
        with torch.no_grad():
            out = m(x, tmap[t], kw["loss_target"], loss_prev=kw["loss_prev"], x_prev=kw["x_prev"])
        x, _ = diff._posterior(out, x, steps[i].contiguous(), t, None)
    assert torch.equal(fused, x)                                                            # (a)
    rows = slice(600, 640)
    part = diff.p_sample_loop(m, (40, D), noise=start[rows], clip_denoised=False, step_noise=steps[:, rows].contiguous(),
                              model_kwargs={k: v[rows] for k, v in kw.items()})
    scale = float((fused[rows] - kw["x_prev"][rows]).abs().max())
    assert float((part - fused[rows]).abs().max()) / scale < 3e-2                           # (b)
