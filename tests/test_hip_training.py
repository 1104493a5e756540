"""GPU parity tests of the training path: backward kernels against torch autograd (fp64 on the CPU), and
training_losses + backward + one optimisation step against the fixtures produced by the reference."""
from copy import deepcopy

import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import denoiser as od          # noqa: E402
from tests.cases import CASES, gpt_kwargs, make_inputs, slim  # noqa: E402
from tests.test_hip_parity import DEV, build, plan_of, rel_err  # noqa: E402

import gpt_amd                              # noqa: E402
from gpt_amd import _lib as L               # noqa: E402


def _assert_equal_up_to_adam_noise(a, b, k, lr=1e-3):
    """Two runs of the same steps: the atomically summed gradients (LayerNorm / bias sums, the norm taken from the tiles)
    differ in their last bits between ANY two runs, a weight on a bf16 rounding boundary then packs differently, and Adam's
    first steps are sign-like, so isolated elements with a near-zero gradient move by a good part of lr per step.  Hence: a
    tight bar on the mean, a bar of lr / 2 that all but a 1e-4 share of a tensor's elements (or one) must meet, and a hard bar of 3 lr
    (a run on stale or missing updates is off by lr per step in EVERY element)."""
    dk = (a - b).abs()
    scale = max(1.0, float(b.abs().max()))
    if os.environ.get("GPT_TEST_TIGHT_BAR") == "1":     # the deterministic configuration (see the _tight_bar test below)
        assert float(dk.max()) < 0.5 * lr * scale, (k, float(dk.max()))     # NO element beyond lr / 2 (measured: 2.8e-4 = 0.28 lr)
    assert float(dk.mean()) < 1e-5 * max(1.0, float(b.abs().mean())), k
    assert int((dk > 0.5 * lr * scale).sum()) <= max(1, int(1e-4 * dk.numel())), (k, float(dk.max()))
    assert float(dk.max()) < 3 * lr * scale, (k, float(dk.max()))


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("H,hd,T", [(2, 96, 26), (2, 128, 26), (3, 32, 18), (1, 64, 32)])
def test_attention_backward(dtype, tol, H, hd, T):
    g = torch.Generator().manual_seed(H * hd + T)
    B, d = 3, H * hd
    code = L.dtype_code(dtype)
    td = L.torch_dtype(code)
    qkv_d = torch.randn(B * T, 3 * d, generator=g).to(DEV).to(td)
    dy_d = torch.randn(B * T, d, generator=g).to(DEV).to(td)
    qkv = qkv_d.float().cpu().double().requires_grad_(True)
    f = qkv.view(B, T, 3, H, hd)
    q, k, v = (f[:, :, i].transpose(1, 2) for i in range(3))
    y = (torch.softmax(q @ k.transpose(-2, -1) / np.sqrt(hd), dim=-1) @ v).transpose(1, 2).reshape(B * T, d)
    y.backward(dy_d.float().cpu().double())
    dqkv = torch.empty_like(qkv_d)
    L.check(L.lib().gpt_attention_bwd(code, L.ptr(qkv_d), L.ptr(dy_d), L.ptr(dqkv), B, T, H, hd, L.stream()), "attention_bwd")
    assert rel_err(dqkv.float().cpu(), qkv.grad) < tol


@pytest.mark.parametrize("d", [192, 1536, 2048, 3072, 520])      # 3072: 96 KiB of LDS for the workgroup reduction
@pytest.mark.parametrize("dy_dtype", ["fp32", "bf16"])
def test_layernorm_backward(d, dy_dtype):
    g = torch.Generator().manual_seed(d)
    rows = 130
    x = (torch.randn(rows, d, generator=g) * 2 + 0.3)
    w = 1 + 0.1 * torch.randn(d, generator=g)
    resid = torch.randn(rows, d, generator=g)
    code = L.dtype_code(dy_dtype)
    dy_d = torch.randn(rows, d, generator=g).to(DEV).to(L.torch_dtype(code))
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = torch.zeros(d, dtype=torch.double, requires_grad=True)
    torch.nn.functional.layer_norm(xr, (d,), wr, br, 1e-5).backward(dy_d.float().cpu().double())
    xd, wd, rd = x.to(DEV), w.to(DEV), resid.to(DEV)
    mean = x.mean(1).to(DEV)
    rstd = (1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)).to(DEV)
    gout, gt = torch.empty(rows, d, device=DEV), torch.empty(rows, d, device=DEV, dtype=torch.bfloat16)
    dg, db = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV)
    L.check(L.lib().gpt_layernorm_bwd(L.ptr(dy_d), code, L.ptr(xd), L.ptr(mean), L.ptr(rstd), L.ptr(wd), L.ptr(rd), L.ptr(gout),
                                      L.ptr(gt), L.BF16, L.ptr(dg), L.ptr(db), rows, d, L.stream()), "ln_bwd")
    assert rel_err(gout.cpu() - resid, xr.grad) < 2e-5
    assert rel_err(gt.float().cpu(), gout.cpu()) < 1e-2
    assert rel_err(dg.cpu(), wr.grad) < 2e-5 and rel_err(db.cpu(), br.grad) < 2e-5
    # the variant that also accumulates the column sums of gout (the bias gradient of the Linear behind the residual add)
    gout2, gt2 = torch.empty_like(gout), torch.empty_like(gt)
    dg2, db2, gcol = torch.zeros(d, device=DEV), torch.zeros(d, device=DEV), torch.ones(d, device=DEV)
    L.check(L.lib().gpt_layernorm_bwd_colsum(L.ptr(dy_d), code, L.ptr(xd), L.ptr(mean), L.ptr(rstd), L.ptr(wd), L.ptr(rd), L.ptr(gout2),
                                             L.ptr(gt2), L.BF16, L.ptr(dg2), L.ptr(db2), L.ptr(gcol), rows, d, L.stream()), "ln_bwd_colsum")
    assert torch.equal(gout2, gout) and torch.equal(gt2, gt)
    assert rel_err(dg2.cpu(), wr.grad) < 2e-5 and rel_err(db2.cpu(), br.grad) < 2e-5
    assert rel_err(gcol.cpu() - 1, gout.cpu().double().sum(0)) < 2e-5


def test_small_gemm_colsum_transpose():
    g = torch.Generator().manual_seed(4)
    A, Bm = torch.randn(70, 33, generator=g), torch.randn(33, 90, generator=g)
    Ad, Bd = A.to(DEV), Bm.t().contiguous().to(DEV).to(torch.bfloat16)     # B stored transposed, bf16
    Cd = torch.ones(70, 90, device=DEV)
    L.check(L.lib().gpt_small_gemm(L.F32, L.ptr(Ad), 33, 1, L.BF16, L.ptr(Bd), 1, 33, L.ptr(Cd), 90, 70, 90, 33, 1, L.stream()), "sg")
    ref = 1 + A.double() @ Bd.float().cpu().t().double()
    assert rel_err(Cd.cpu(), ref) < 1e-5
    X = torch.randn(300, 500, generator=g).to(DEV)
    out = torch.zeros(500, device=DEV)
    L.check(L.lib().gpt_colsum(L.F32, L.ptr(X), 500, L.ptr(out), 300, 500, L.stream()), "colsum")
    assert rel_err(out.cpu(), X.cpu().double().sum(0)) < 1e-5
    Xt = torch.full((500, 320), 7.0, device=DEV)
    L.check(L.lib().gpt_transpose(L.F32, L.ptr(X), L.ptr(Xt), 300, 500, 320, L.stream()), "transpose")
    assert torch.equal(Xt[:, :300].cpu(), X.cpu().t()) and float(Xt[:, 300:].abs().max()) == 0
    # bf16: the 16-byte path (cols % 8 == 0, ld_dst % 8 == 0) with ragged tile counts and zero-filled padding, and the
    # 2-byte path for the other shapes
    for rows, cols, ld in ((300, 504, 320), (1536, 6144, 1536), (70, 64, 72), (300, 500, 320), (64, 72, 70)):
        Xb = torch.randn(rows, cols, generator=g).to(DEV).bfloat16()
        Xbt = torch.full((cols, ld), 7.0, device=DEV, dtype=torch.bfloat16)
        L.check(L.lib().gpt_transpose(L.BF16, L.ptr(Xb), L.ptr(Xbt), rows, cols, ld, L.stream()), "transpose")
        assert torch.equal(Xbt[:, :rows].cpu(), Xb.cpu().t())
        assert ld == rows or float(Xbt[:, rows:].float().abs().max()) == 0


@pytest.mark.parametrize("case,tag", [("mnist_small", "x0"), ("mnist_small", "eps"), ("cifar_small", "x0")])
def test_training_losses_and_grads_golden(golden, case, tag):
    """diffusion.training_losses(model, ...) + .backward() through the HIP path, fp32, against the reference."""
    g = golden(f"g6_training_{case}_{tag}")
    px = tag == "x0"
    m, plan, sd = build(case, "fp32")
    m.predict_xstart = px
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = make_inputs(case, B=6, seed=21)
    x_start, t, noise = (torch.from_numpy(g[k]).to(DEV) for k in ("x_start", "t", "noise"))
    terms = diff.training_losses(m, x_start, t, model_kwargs=dict(loss_target=lt.double().to(DEV), loss_prev=lp.double().to(DEV),
                                                                  x_prev=xp.to(DEV)), noise=noise)
    np.testing.assert_allclose(terms["mse"].detach().cpu().numpy(), g["mse"], rtol=2e-4)
    loss = terms["loss"].mean()
    assert abs(float(loss) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    loss.backward()
    named = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("grad::"):
            got = slim(named[k[6:]].grad.cpu().numpy())
            assert np.abs(got - g[k]).max() < 1e-3 * np.abs(g[k]).max() + 1e-9, k
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    assert abs(gn - float(g["grad_norm"])) < 1e-3 * float(g["grad_norm"])
    assert abs(m.gradient_norm() - gn) < 1e-4 * gn


def test_train_step_golden(golden):
    """loss -> backward -> clip 0.75 -> AdamW (2 groups) -> EMA 0.9999, the sequence of main.py:63-74."""
    g = golden("g9_trainstep_mnist_small")
    m, plan, sd = build("mnist_small", "fp32")
    m.train()
    ema = deepcopy(m)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = make_inputs("mnist_small", B=6, seed=41)
    x_start, t, noise = (torch.from_numpy(g[k]).to(DEV) for k in ("x_start", "t", "noise"))
    opt = m.configure_optimizers(lr=4e-4, wd=0.1, betas=(0.9, 0.999))
    terms = diff.training_losses(m, x_start, t, model_kwargs=dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV)),
                                 noise=noise)
    loss = (terms["loss"] * torch.ones(6, device=DEV)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.75)
    opt.step()
    with torch.no_grad():
        pe, pm = dict(ema.named_parameters()), dict(m.named_parameters())
        for k in pe:
            pe[k].mul_(0.9999).add_(pm[k], alpha=1 - 0.9999)
    assert abs(float(loss) - float(g["loss"])) < 2e-4 * float(g["loss"])
    assert abs(float(total_norm) - float(g["total_norm"])) < 1e-3 * float(g["total_norm"])
    for k in g.files:
        if k.startswith("param::"):
            np.testing.assert_allclose(slim(pm[k[7:]].detach().cpu().numpy()), g[k], rtol=0, atol=2e-6, err_msg=k)
        if k.startswith("ema::"):
            np.testing.assert_allclose(slim(pe[k[5:]].detach().cpu().numpy()), g[k], rtol=0, atol=2e-6, err_msg=k)
    # after the step the packed weights must follow the parameters (stale-copy check)
    with torch.no_grad():
        x, tt, lt2, lp2, xp2 = (v.to(DEV) for v in make_inputs("mnist_small", B=4, seed=5))
        out = m(x, tt, lt2, loss_prev=lp2, x_prev=xp2)
        ref = od.forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, plan, *make_inputs("mnist_small", B=4, seed=5))
    assert np.abs(out.cpu().numpy() - ref.numpy()).max() / np.abs((ref - xp2.cpu()).numpy()).max() < 2e-4


def test_training_bf16_close_to_fp32():
    m32, plan, sd = build("mnist_small", "fp32", seed=2)
    m16, _, _ = build("mnist_small", "bf16", seed=2)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("mnist_small", B=8, seed=3))
    g = torch.Generator().manual_seed(9)
    x_start = xp + 0.1 * torch.randn(xp.shape, generator=g).to(DEV)
    t = torch.randint(0, 1000, (8,), generator=g).to(DEV)
    noise = torch.randn(xp.shape, generator=g).to(DEV)
    grads = []
    for m in (m32, m16):
        m.train()
        terms = diff.training_losses(m, x_start, t, model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), noise=noise)
        terms["loss"].mean().backward()
        grads.append({k: p.grad.float().cpu() for k, p in m.named_parameters()})
    for k in ("decoder.blocks.1.mlp.0.weight", "decoder.blocks.0.attn.query.weight", "decoder.ln_f.weight",
              "decoder.input_parameter_projections.0.0.weight", "decoder.output_parameter_projections.2.0.weight"):
        a, b = grads[1][k], grads[0][k]
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        assert cos > 0.99, (k, cos)


@pytest.mark.parametrize("fused", [True, False])
def test_fused_train_step_golden(golden, fused):
    """TrainStep (flat arenas; fused clip + AdamW + EMA + operand re-packing in one pass) against the same fixture."""
    from gpt_amd.trainstep import TrainStep
    g = golden("g9_trainstep_mnist_small")
    m, plan, sd = build("mnist_small", "fp32")
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = make_inputs("mnist_small", B=6, seed=41)
    x_start, t, noise = (torch.from_numpy(g[k]).to(DEV) for k in ("x_start", "t", "noise"))
    ts = TrainStep(m, diff, lr=4e-4, wd=0.1, betas=(0.9, 0.999), grad_clip=0.75, ema_decay=0.9999, fused=fused)
    mse = ts.loss_and_grads(xp.to(DEV), x_start, lp.to(DEV), lt.to(DEV), t, weights=torch.ones(6, device=DEV), noise=noise)
    assert abs(float(mse.mean()) - float(g["loss"])) < 2e-4 * float(g["loss"])
    gn = float(torch.linalg.vector_norm(ts.grads.double()))
    assert abs(gn - float(g["total_norm"])) < 1e-3 * float(g["total_norm"])
    ts.exchange()
    ts.update()
    if fused:
        assert abs(float(ts.grad_norm_sq.sqrt()) - gn) < 1e-5 * gn
        assert float(ts.grads.abs().max()) == 0.0
    pm, pe = dict(m.named_parameters()), dict(ts.ema.named_parameters())
    for k in g.files:
        if k.startswith("param::"):
            np.testing.assert_allclose(slim(pm[k[7:]].detach().cpu().numpy()), g[k], rtol=0, atol=2e-6, err_msg=k)
        if k.startswith("ema::"):
            np.testing.assert_allclose(slim(pe[k[5:]].detach().cpu().numpy()), g[k], rtol=0, atol=2e-6, err_msg=k)
    # the operands the next forward reads must be the updated weights, for the model and for the EMA copy
    for net in (m, ts.ema):
        with torch.no_grad():
            x, tt, lt2, lp2, xp2 = (v.to(DEV) for v in make_inputs("mnist_small", B=4, seed=5))
            out = net(x, tt, lt2, loss_prev=lp2, x_prev=xp2)
            ref = od.forward({k: v.detach().cpu() for k, v in net.state_dict().items()}, plan, *make_inputs("mnist_small", B=4, seed=5))
        assert np.abs(out.cpu().numpy() - ref.numpy()).max() / np.abs((ref - xp2.cpu()).numpy()).max() < 2e-4


def test_fused_step_matches_torch_optimizer_over_steps():
    """Five steps, bf16 compute: the fused pass and torch's AdamW / clip / EMA see the same gradients and must agree."""
    from gpt_amd.trainstep import TrainStep
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    runs = []
    for fused in (True, False):
        m, plan, sd = build("cifar_small", "bf16", seed=4)
        m.train()
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, fused=fused)
        for i in range(5):
            _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=8, seed=50 + i))
            gen = torch.Generator().manual_seed(70 + i)
            x1 = xp + 0.1 * torch.randn(xp.shape, generator=gen).to(DEV)
            t = torch.randint(0, 1000, (8,), generator=gen).to(DEV)
            noise = torch.randn(xp.shape, generator=gen).to(DEV)
            ts.set_lr(1e-3 * (i + 1) / 5)
            ts.step(xp, x1, lp, lt, t, noise=noise)
        runs.append(({k: p.detach().float().cpu() for k, p in m.named_parameters()},
                     {k: p.detach().float().cpu() for k, p in ts.ema.named_parameters()}))
    for a, b in zip(runs[0], runs[1]):
        for k in a:
            if k.endswith("attn.key.bias"):
                continue      # its true gradient is zero (softmax is shift invariant): Adam normalises pure rounding noise
            _assert_equal_up_to_adam_noise(a[k], b[k], k)


def test_fused_step_matches_torch_optimizer_tight_bar_in_the_deterministic_configuration():
    """The same five steps with the bias column sums and the norm on their separate two-pass kernels (GPT_DGELU_COLSUM=0
    GPT_LNB_COLSUM=0 GPT_WGRAD_SUMSQ=0: no float atomics but LayerNorm's dgamma / dbeta and pos_emb), in a process of its own
    (the switches are read once per process): fused pass and torch optimizer must then agree to lr / 2 in
    EVERY element (what is left is a weight on a bf16 rounding boundary that the two AdamW formulas round apart,
    which changes the next step's gradients: 0.28 lr measured), where the default configuration's bar tolerates a 1e-4 share
    of elements up to 3 lr apart."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPT_DGELU_COLSUM="0", GPT_LNB_COLSUM="0", GPT_WGRAD_SUMSQ="0", GPT_TEST_TIGHT_BAR="1")
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          "tests/test_hip_training.py::test_fused_step_matches_torch_optimizer_over_steps"],
                         cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


@pytest.mark.parametrize("dtype,B", [("fp32", 5), ("bf16", 32)])
def test_layernorm_backward_partials_equal_atomics(monkeypatch, dtype, B):
    """gpt_backward_t.ln_partials (ABI 4): the LayerNorm backward launches store per-workgroup dgamma / dbeta / bias sums and one
    launch per call adds them up, against the float-atomics route (no buffer): every gradient equal to summation order, and the
    partials route bit-reproducible from run to run in the tensors it produces."""
    from gpt_amd import backward as gb
    m, plan, sd = build("cifar_small", dtype, seed=19)
    m.train()
    x, t, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=B, seed=61))
    xx, tt, l1, l2, xpp = m._prepare(x, t, lt, lp, xp)
    d_out = torch.randn(x.shape, generator=torch.Generator().manual_seed(3)).to(DEV) * 1e-2
    orig = gb._descriptor

    def without_partials(*a, **k):
        bw, keep = orig(*a, **k)
        bw.ln_partials, bw.ln_partials_bytes = None, 0
        return bw, keep

    bufs = []
    for variant in ("partials", "partials", "atomics"):
        monkeypatch.setattr(gb, "_descriptor", without_partials if variant == "atomics" else orig)
        eng, ws, out = gb.forward_keep(m, xx, tt, l1, l2, xpp)
        buf = torch.zeros(eng.arena.numel, device=DEV)
        gb.backward_into(eng, ws, xx, xpp, d_out, buf)
        torch.cuda.synchronize()
        bufs.append({k: v.clone() for (k, _), v in zip(m.named_parameters(), eng.arena.views(buf, m))})
    ln_like = [k for k in bufs[0] if ".ln" in k or "ln_" in k or k.endswith("mlp.2.bias") or k.endswith("attn.proj.bias")]
    assert len(ln_like) >= 2 * (2 * plan.n_layer + 2)
    for k in bufs[0]:
        if k.endswith("attn.key.bias"):
            continue              # true gradient zero: rounding noise of the atomically summed q|k|v column sums on every route
        a, b, c = bufs[0][k], bufs[1][k], bufs[2][k]
        scale = float(c.abs().max()) + 1e-30
        assert float((a - c).abs().max()) <= 2e-5 * scale + 1e-12, k
        if k in ln_like:
            assert torch.equal(a, b), k                       # fixed summation order: the same bits every run


def test_store_mode_weight_gradients_equal_accumulation(monkeypatch):
    """Where the grouped weight-gradient launch applies (bf16, d % 256 == 0) TrainStep has the Blocks' weight
    gradients STORED and stops zeroing their part of the arena: three steps must give the parameters of the accumulate-into-zero
    path (a + 0 is exact; LayerNorm / bias gradients are atomically summed, hence a tolerance), and the arena really holds stale
    block-weight gradients in between."""
    from gpt_amd.trainstep import TrainStep
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("GPT_WGRAD_OVERWRITE", flag)
        m, plan, sd = build("cifar_small", "bf16", seed=4)
        m.train()
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75)
        for i in range(3):
            _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=32, seed=150 + i))
            gen = torch.Generator().manual_seed(170 + i)
            x1 = xp + 0.1 * torch.randn(xp.shape, generator=gen).to(DEV)
            t = torch.randint(0, 1000, (32,), generator=gen).to(DEV)
            noise = torch.randn(xp.shape, generator=gen).to(DEV)
            ts.step(xp, x1, lp, lt, t, noise=noise)
        assert ts._overwrote == (flag == "1")
        cut = ts.arena.s_dec_start
        assert float(ts.grads[cut:].abs().max()) == 0.0                        # what still accumulates is zeroed
        assert (float(ts.grads[:cut].abs().max()) > 0.0) == (flag == "1")      # stored gradients are simply left behind
        runs.append({k: p.detach().float().cpu() for k, p in m.named_parameters()})
    for k in runs[0]:
        if k.endswith("attn.key.bias"):
            continue
        _assert_equal_up_to_adam_noise(runs[0][k], runs[1][k], k)
    monkeypatch.setenv("GPT_WGRAD_OVERWRITE", "1")
    # a row count that is no multiple of the TN kernel's 64-row K-tile (B * T = 130) is taken as well (zero-filled last K-tile) ...
    m, plan, sd = build("cifar_small", "bf16", seed=4)
    m.train()
    ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75)
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=5, seed=1))
    ts.step(xp, xp + 0.1, lp, lt, torch.zeros(5, dtype=torch.long, device=DEV))
    assert ts._overwrote and float(ts.grads[ts.arena.s_dec_start:].abs().max()) == 0.0
    # ... a width the grouped launch does not take (n_embd = 192) is not: the flag must stay off and the step must work
    m, plan, sd = build("mnist_small", "bf16", seed=4)
    m.train()
    ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75)
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("mnist_small", B=5, seed=1))
    ts.step(xp, xp + 0.1, lp, lt, torch.zeros(5, dtype=torch.long, device=DEV))
    assert not ts._overwrote and float(ts.grads.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,B,store", [("bf16", 32, True), ("bf16", 5, True), ("bf16", 32, False), ("fp32", 8, False)])
def test_overlapped_optimizer_pass_equals_the_in_order_one(monkeypatch, dtype, B, store):
    """overlap_update: the Blocks' share of the fused AdamW + EMA pass runs on a side stream, layer by layer, beside the next
    step's forward, which waits per block (gpt_layer_t.ready).  Four steps must give the parameters and the EMA of the in-order
    step (elementwise pass cut at layer boundaries: the only differences are those of the atomically summed gradients), read
    through state_dict(), whose hook joins the side stream; in store mode (also with a ragged row count) and in accumulate mode
    (GPT_WGRAD_OVERWRITE=0; fp32)."""
    from gpt_amd.trainstep import TrainStep
    monkeypatch.setenv("GPT_WGRAD_OVERWRITE", "1" if store else "0")
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    runs = []
    for flag in (True, False):
        m, plan, sd = build("cifar_small", dtype, seed=4)
        m.train()
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, overlap_update=flag)
        for i in range(4):
            _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=B, seed=150 + i))
            gen = torch.Generator().manual_seed(170 + i)
            x1 = xp + 0.1 * torch.randn(xp.shape, generator=gen).to(DEV)
            t = torch.randint(0, 1000, (B,), generator=gen).to(DEV)
            noise = torch.randn(xp.shape, generator=gen).to(DEV)
            ts.step(xp, x1, lp, lt, t, noise=noise)
        # store mode (the Blocks' weight gradients overwritten, nothing to zero) and accumulate mode (zeroed by the same launches)
        assert ts._overwrote == store and (ts._opt_done is not None) == flag
        assert float(ts.grads[ts.arena.s_dec_start:].abs().max()) == 0.0
        if not store:
            torch.cuda.synchronize()
            assert float(ts.grads.abs().max()) == 0.0
        got = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        got.update({"ema." + k: v.detach().float().cpu() for k, v in ts.ema.state_dict().items()})
        assert ts._opt_done is None                       # the state_dict hooks joined
        # the engine's operand copies were written by the same launches: a forward through them must see the new weights
        x, tm, lt2, lp2, xp2 = (v.to(DEV) for v in make_inputs("cifar_small", B=8, seed=9))
        with torch.no_grad():
            got["_out"] = m.eval()(x, tm, lt2, lp2, xp2).float().cpu()
        runs.append(got)
    for k in runs[0]:
        if k.endswith("attn.key.bias"):
            continue
        dk = (runs[0][k] - runs[1][k]).abs()
        if k == "_out":
            # a bf16 forward through weights that agree to the bars below (their bf16 roundings may differ by an ulp); a forward
            # through the operands of a step earlier would be off by the four steps' lr = 4e-3 in every weight
            assert float(dk.max()) < 1e-2 and float(dk.mean()) < 1e-3, (float(dk.max()), float(dk.mean()))
            continue
        _assert_equal_up_to_adam_noise(runs[0][k], runs[1][k], k)


# ---------------------------------------------------------------------------------------------- data parallel
def _ddp_rank(rank, world, port, out_path, mode="replicated", dtype="fp32", case="mnist_small", rows_total=8, stride=1):
    """One rank of a 2-process data-parallel step; both ranks use GPU 0 (the box has one), so the exchange runs on gloo.
    mode: 'replicated' (all-reduce, every rank updates everything) | 'sharded' (reduce-scatter, 1/world AdamW, all-gather of the
    operands) | 'sharded_bf16' (the same with the bf16 gradient payload)."""
    import faulthandler
    import os
    import torch.distributed as dist
    faulthandler.dump_traceback_later(90 if case == "mnist_small" else 280, exit=True)   # a stuck collective must end the rank, not the box
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpt_amd.trainstep import TrainStep
        m, plan, sd = build(case, dtype, seed=3 + rank)       # different weights: rank 0's must win the broadcast
        m.train()
        diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
        # default bucket size at the full size (226 MB buckets, the last two layers singly), one layer per bucket on the small model
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, bucket_layers=1 if case == "mnist_small" else None,
                       shard=mode != "replicated", grad_payload="bf16" if mode == "sharded_bf16" else "fp32", ema_decay=_EMA_DECAY)
        assert ts.shard == (mode != "replicated")
        for i in range(2):
            batch = _ddp_batch(i, case, rows_total)
            rows = rows_total // world
            sl = slice(rank * rows, rank * rows + rows)
            ts.step(*(v[sl] for v in batch))
        x, tt, lt2, lp2, xp2 = (v.to(DEV) for v in make_inputs(case, B=4, seed=5))
        # a forward through the EMA copy right after step() (the reference's periodic EMA evaluation): under sharding the other
        # ranks' EMA shards must have been gathered before its engine packs its operands
        with torch.no_grad():
            ema_out = ts.ema(x, tt, lt2, loss_prev=lp2, x_prev=xp2).cpu()
        if ts.shard:
            assert ts.exp_avg.numel() == ts.n_own + ts.n_rep and ts.n_own == ts.arena.s_end // world    # moments: own shard + region R
            assert not ts._ema_stale
            if dtype == "bf16":
                assert ts._params_stale                                                # fp32 copies of the other shard: on demand
        # state_dict() (what a checkpoint writes, main.py:92-97) must be whole on every rank: its hook gathers the fp32 shards
        sd_now = {k: _thin(v, stride) for k, v in m.state_dict().items()}
        ema_now = {k: _thin(v, stride) for k, v in ts.ema.state_dict().items()}
        # ... and what the next forward reads (packed operands) must be the updated weights on every rank
        with torch.no_grad():
            out = m(x, tt, lt2, loss_prev=lp2, x_prev=xp2).cpu()
        torch.save({"params": {k: v for k, v in sd_now.items() if k != "scalar_embedder.frequencies"},
                    "ema": {k: v for k, v in ema_now.items() if k != "scalar_embedder.frequencies"}, "out": out,
                    "ema_out": ema_out}, out_path + f".{rank}")
        dist.barrier()
    finally:
        dist.destroy_process_group()


_EMA_DECAY = 0.5            # far from 1, so that an EMA shard that missed its update (or its gather) is visible after two steps


def _thin(v, stride):
    """Every stride-th element of a large tensor (the full-size model's checkpoints are 2 x 1.5 GB per rank otherwise); a stride
    coprime to every chunk size, so every shard and bucket is sampled."""
    v = v.detach()
    return (v.reshape(-1)[::stride] if stride > 1 and v.numel() > (1 << 16) else v).cpu().clone()


def _ddp_batch(i, case="mnist_small", rows_total=8):
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs(case, B=rows_total, seed=80 + i))
    gen = torch.Generator().manual_seed(90 + i)
    x1 = xp + 0.1 * torch.randn(xp.shape, generator=gen).to(DEV)
    t = torch.randint(0, 1000, (rows_total,), generator=gen).to(DEV)
    noise = torch.randn(xp.shape, generator=gen).to(DEV)
    return xp, x1, lp, lt, t, torch.ones(rows_total, device=DEV), noise


@pytest.mark.parametrize("mode,dtype", [("replicated", "fp32"), ("sharded", "fp32"), ("sharded", "bf16"), ("sharded_bf16", "bf16")])
def test_two_rank_step_equals_full_batch(tmp_path, mode, dtype):
    """2 ranks x 4 rows with the staged, overlapped gradient exchange == 1 process x 8 rows (two steps): the replicated step
    (all-reduce), the sharded step (reduce-scatter -> AdamW on 1/world -> all-gather of the operands; fp32 and bf16 compute)
    and the sharded step with the bf16 gradient payload.  Parameters, EMA and the next forward's output, on BOTH ranks."""
    _ranks_equal_full_batch(tmp_path, mode, dtype, 2)


@pytest.mark.parametrize("world", [4, 8])
def test_more_ranks_sharded_step_equals_full_batch(tmp_path, world):
    """The same with 4 ranks x 2 rows and 8 ranks x 1 row (the driver's largest run): the shard boundaries of every bucket
    fall inside other tensors than with 2 ranks (a quarter of a layer's 12 d^2 weights = the fused q|k|v gradient exactly,
    then proj + half of mlp.0, ...)."""
    _ranks_equal_full_batch(tmp_path, "sharded", "fp32", world)


@pytest.mark.parametrize("mode", ["sharded", "sharded_bf16"])
def test_full_size_two_rank_sharded_step_equals_full_batch(tmp_path, mode):
    """The sharded step at the size the driver's scaling runs use: the full mnist model (d = 1536, 12 layers,
    378 M parameters: 226 MB buckets with the last two layers going out singly, chunk boundaries inside q|k|v, one ready event
    per bucket, sync_params on 1.5 GB), bf16 compute, 2 ranks x 64 rows against one process x 128 rows, two steps; fp32 and
    bf16 gradient payload.  Every 7th element of every large tensor of the parameters and the EMA, and the whole output of the
    next forward (which reads every gathered operand) and of an EMA forward, on both ranks."""
    _ranks_equal_full_batch(tmp_path, mode, "bf16", 2, case="mnist", rows_total=128, stride=7)


def _ranks_equal_full_batch(tmp_path, mode, dtype, world, case="mnist_small", rows_total=8, stride=1):
    import socket
    import torch.multiprocessing as mp
    from gpt_amd.trainstep import TrainStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "rank")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ddp_rank, args=(r, world, port, out_path, mode, dtype, case, rows_total, stride)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240 if case == "mnist_small" else 300)
        if p.exitcode is None:
            p.kill()
        assert p.exitcode == 0
    m, plan, sd = build(case, dtype, seed=3)
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, ema_decay=_EMA_DECAY)
    for i in range(2):
        ts.step(*_ddp_batch(i, case, rows_total))
    with torch.no_grad():
        x, tt, lt2, lp2, xp2 = (v.to(DEV) for v in make_inputs(case, B=4, seed=5))
        want_out = m(x, tt, lt2, loss_prev=lp2, x_prev=xp2).cpu()
        want_ema_out = ts.ema(x, tt, lt2, loss_prev=lp2, x_prev=xp2).cpu()
    # fp32 compute: the two paths differ by summation order only.  bf16 compute: the gradients of 2 x 4 rows and of 8 rows see
    # differently rounded operands; a bf16 gradient payload rounds the summed gradients once more (Adam's first steps are
    # sign-like, so single elements may move by a good part of lr = 1e-3)
    # bf16 compute: after step 1 a weight that sits on a bf16 rounding boundary may round differently on the two paths, and an
    # element whose gradient is below the bf16 noise floor gets a sign-like Adam update: single elements may differ by up to
    # 2 lr = 2e-3 after two steps, so the bar is on the maximum (< 2 lr) AND on the mean (< lr / 20; lr / 10 at the full size,
    # where an element's gradient is smaller against the bf16 noise floor; a shard that missed its update or its gather is off
    # by about lr in EVERY element)
    tol = 5e-5 if dtype == "fp32" else 2e-3
    mean_bar = 5e-5 if case == "mnist_small" else 1e-4
    worst = [0.0, 0.0, 0.0, 0.0]
    for rank in range(world):
        got = torch.load(out_path + f".{rank}", weights_only=True)
        for name, ref_mod in (("params", m), ("ema", ts.ema)):
            for k, p in ref_mod.named_parameters():
                if k.endswith("attn.key.bias"):
                    continue
                diff_k = (_thin(p, stride) - got[name][k]).abs()
                err = float(diff_k.max())
                worst[:2] = [max(worst[0], err), max(worst[1], float(diff_k.mean()))]
                scale_k = max(1.0, float(got[name][k].abs().max()))
                if case == "mnist_small":
                    assert err < tol * scale_k, (rank, name, k, err)
                else:
                    # 378 M elements: the tail of the sign-like first Adam steps is longer (a handful of elements move by lr per
                    # step in opposite directions on the two paths; measured: 4e-4 of pos_emb, 2.7e-3 of block 0's key weight, whose gradient is all but zero under near-uniform attention): hard
                    # bar 5 lr, and all but 1 % within lr / 2 - a chunk that missed its update or its gather is off by about lr in
                    # ALL its elements, i.e. in whole tensors or in the larger part of one
                    assert err < 5e-3 * scale_k, (rank, name, k, err)     # (two sign-like steps apart + bf16 roundings: measured 3.05e-3)
                    far = int((diff_k > 0.5e-3 * scale_k).sum())
                    worst[2] = max(worst[2], far / diff_k.numel())
                    assert far <= max(1, int(1e-2 * diff_k.numel())), (rank, name, k, far, diff_k.numel())
                assert float(diff_k.mean()) < mean_bar, (rank, name, k, float(diff_k.mean()))
        # the full-size model in bf16: 12 layers of weights that differ by the Adam noise above (measured 3.2e-2 of the largest
        # output delta with the bf16 payload); operands of a stale shard are off by lr / 0.02 = 5 % in EVERY weight of half the model
        out_bar = 1e-3 if dtype == "fp32" else 3e-2 if case == "mnist_small" else 6e-2
        for key, want in (("out", want_out), ("ema_out", want_ema_out)):
            scale = float((want - xp2.cpu()).abs().max())
            d_out = (got[key] - want).abs()
            worst[3] = max(worst[3], float(d_out.max()) / scale)
            assert float(d_out.max()) / scale < out_bar, (rank, key, float(d_out.max()) / scale)
            assert float(d_out.mean()) / scale < out_bar / 10, (rank, key, float(d_out.mean()) / scale)
    print(f"[{case} {mode} {dtype} world {world}] worst |diff|: max {worst[0]:.2e}, mean {worst[1]:.2e}, share beyond lr/2 {worst[2]:.1e}; next forward {worst[3]:.2e}")


def test_autograd_route_delivers_gradients_stage_by_stage():
    """loss.backward() through the bridge (main.py:47-67): every parameter's gradient reaches its hooks (where DDP's reducer
    listens) as soon as ITS backward stage is enqueued, not after the whole backward; the arena is reused across iterations
    unless a .grad still aliases it; results equal the one-call backward."""
    from gpt_amd import backward as gb
    m, plan, sd = build("mnist_small", "fp32", seed=21)
    m.train()
    x, t, lt, lp, xp = (v.to(DEV) for v in make_inputs("mnist_small", B=4, seed=34))
    n_stages = m.decoder.n_layer + 2
    seen, ran = {}, []
    orig = gb._PendingBackward.run_stage

    def counting(self, stage):
        ran.append(stage)
        return orig(self, stage)

    gb._PendingBackward.run_stage = counting
    try:
        hooks = [p.register_hook(lambda g, k=k: seen.__setitem__(k, len(ran))) for k, p in m.named_parameters()]
        out = m(x, t, lt, loss_prev=lp, x_prev=xp)
        out.square().mean().backward()
    finally:
        gb._PendingBackward.run_stage = orig
        for h in hooks:
            h.remove()
    assert ran == list(range(n_stages))
    stages = gb.stage_parameter_names(m)
    for s_idx, names in enumerate(stages):
        for k in names:
            assert seen[k] == s_idx + 1, (k, seen[k], s_idx)          # delivered right after its own stage
    g1 = {k: p.grad.clone() for k, p in m.named_parameters()}
    # one-call backward of the same forward into a fresh arena
    xx, tt, l1, l2, xpp = m._prepare(x, t, lt, lp, xp)
    eng, ws, out2 = gb.forward_keep(m, xx, tt, l1, l2, xpp)
    buf = eng.arena.new_buffer()
    gb.backward_into(eng, ws, xx, xpp, (2.0 / out2.numel()) * out2, buf)
    for (k, p), v in zip(m.named_parameters(), eng.arena.views(buf, m)):
        assert torch.allclose(g1[k], v, rtol=1e-4, atol=1e-6 * float(v.abs().max()) + 1e-12), k
    # second iteration: zero_grad(set_to_none) as main.py:66 does -> the arena is reused; without it (.grad may alias) it is not
    first = eng._grad_buf.data_ptr()
    m.zero_grad(set_to_none=True)
    m(x, t, lt, loss_prev=lp, x_prev=xp).square().mean().backward()
    assert eng._grad_buf.data_ptr() == first
    for k, p in m.named_parameters():
        assert torch.allclose(p.grad, g1[k], rtol=1e-4, atol=1e-6 * float(g1[k].abs().max()) + 1e-12), k
    m(x, t, lt, loss_prev=lp, x_prev=xp).square().mean().backward()      # accumulation on top of held gradients
    for k, p in m.named_parameters():
        if not k.endswith("attn.key.bias"):
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-3, atol=1e-5 * float(g1[k].abs().max()) + 1e-12), k


def _ddp_autograd_rank(rank, world, port, out_path):
    """The reference's own training iteration (main.py:229-240 + 63-74) on 2 ranks: DistributedDataParallel(static_graph=True)
    around the HIP Gpt, training_losses(...).backward(), clip, torch AdamW."""
    import faulthandler
    import os
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    faulthandler.dump_traceback_later(100, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, plan, sd = build("mnist_small", "fp32", seed=3 + rank)          # DDP broadcasts rank 0's weights at construction
        m.train()
        diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
        ddp = DistributedDataParallel(m, device_ids=[0], output_device=0, static_graph=True)
        opt = m.configure_optimizers(lr=1e-3, wd=0.1, betas=(0.9, 0.999))
        for i in range(2):
            xp, x1, lp, lt, t, w, noise = (v[rank * 4: rank * 4 + 4] for v in _ddp_batch(i))
            terms = diff.training_losses(ddp, x1, t, model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), noise=noise)
            loss = (terms["loss"] * w).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.75)
            opt.step()
        if rank == 0:
            torch.save({k: p.detach().cpu() for k, p in m.named_parameters()}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_ddp_static_graph_through_autograd_equals_full_batch(tmp_path):
    """main.py unchanged: DDP(static_graph=True) over the autograd bridge on 2 ranks == the full batch on one process."""
    import socket
    import torch.multiprocessing as mp
    from gpt_amd.trainstep import TrainStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "ddp.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_ddp_autograd_rank, args=(r, 2, port, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        if p.exitcode is None:
            p.kill()
        assert p.exitcode == 0
    got = torch.load(out_path, weights_only=True)
    m, plan, sd = build("mnist_small", "fp32", seed=3)
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, use_ema=False)
    for i in range(2):
        ts.step(*_ddp_batch(i))
    for k, p in m.named_parameters():
        if k.endswith("attn.key.bias"):
            continue
        assert float((p.detach().cpu() - got[k]).abs().max()) < 5e-5 * max(1.0, float(got[k].abs().max())), k


def _rccl_rank(port, out_path):
    """The multi-GPU code path on RCCL with the one rank a one-GPU box allows (GPT_DP_REHEARSE=1, dp.exchanging()):
    parameter broadcast, bucketed asynchronous gradient all-reduces on RCCL's stream overlapped with the backward stages,
    the final row gather and the helper collectives all really run; with one rank each is the identity."""
    import faulthandler
    import os
    import torch.distributed as dist
    faulthandler.dump_traceback_later(100, exit=True)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", GPT_DP_REHEARSE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from gpt_amd import dp
        from gpt_amd.trainstep import TrainStep
        assert dp.exchanging() and dp.get_world_size() == 1
        m, plan, sd = build("mnist_small", "fp32", seed=3)
        m.train()
        dp.broadcast_parameters(m)
        diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, bucket_layers=1)
        assert ts.shard                       # the default under a process group: reduce_scatter / all_gather really run on RCCL
        for i in range(2):
            ts.step(*_ddp_batch(i))
        rows = torch.arange(5 * 7, dtype=torch.float32, device=DEV).view(5, 7)
        assert torch.equal(dp.gather_rows(rows, 5), rows)
        assert torch.equal(dp.all_gather(rows), rows) and dp.all_gather(rows, cat=False).shape == (1, 5, 7)
        assert torch.equal(dp.rank0_to_all(rows), rows[0])
        dp.synchronize()
        torch.cuda.synchronize()
        torch.save({k: p.detach().cpu() for k, p in m.named_parameters()}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_one_rank_rccl_rehearsal_matches_single_process(tmp_path):
    """The RCCL calls of the data-parallel step (never reachable with one rank otherwise) against the single-process step."""
    import socket
    import torch.multiprocessing as mp
    from gpt_amd.trainstep import TrainStep
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = str(tmp_path / "rccl.pt")
    p = mp.get_context("spawn").Process(target=_rccl_rank, args=(port, out_path))
    p.start()
    p.join(timeout=150)
    if p.exitcode is None:
        p.kill()
    assert p.exitcode == 0
    got = torch.load(out_path, weights_only=True)
    m, plan, sd = build("mnist_small", "fp32", seed=3)
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75)
    for i in range(2):
        ts.step(*_ddp_batch(i))
    for k, prm in m.named_parameters():
        if k.endswith("attn.key.bias"):
            continue
        assert float((prm.detach().cpu() - got[k]).abs().max()) < 5e-5 * max(1.0, float(got[k].abs().max())), k


@pytest.mark.parametrize("I,J,R", [(256, 256, 64), (512, 768, 832), (1536, 256, 3328), (256, 512, 8256),     # 8256: 16 row spans, ragged
                                   (256, 256, 40), (512, 256, 130), (768, 512, 1768), (256, 512, 8243)])     # R % 64 != 0: zero-filled last K-tile
@pytest.mark.parametrize("accumulate", [0, 1])
def test_gemm_tn_matches_torch(I, J, R, accumulate):
    """C (+)= P^T Q with both bf16 operands read as stored (transposing LDS reads) against an fp64 product."""
    g = torch.Generator().manual_seed(I + J + R)
    P = torch.randn(R, I, generator=g).bfloat16().to(DEV)
    Q = torch.randn(R, J, generator=g).bfloat16().to(DEV)
    C0 = torch.randn(I, J, generator=g).to(DEV)
    C = C0.clone()
    L.check(L.lib().gpt_gemm_tn(L.ptr(P), I, L.ptr(Q), J, L.ptr(C), J, I, J, R, accumulate, L.stream()), "gpt_gemm_tn")
    ref = P.double().t() @ Q.double() + (C0.double() if accumulate else 0)
    assert float((C.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    # a shorter contraction over the same buffers (any R is accepted: the last K-tile is zero-filled past it)
    if R > 32:
        L.check(L.lib().gpt_gemm_tn(L.ptr(P), I, L.ptr(Q), J, L.ptr(C), J, I, J, R - 29, 0, L.stream()), "gpt_gemm_tn")
        ref = P[:R - 29].double().t() @ Q[:R - 29].double()
        assert float((C.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    # rejected shapes: no launch
    assert L.lib().gpt_gemm_tn(L.ptr(P), I, L.ptr(Q), J, L.ptr(C), J, I - 8, J, R, 0, L.stream()) != 0
    assert L.lib().gpt_gemm_tn(L.ptr(P), I, L.ptr(Q), J, L.ptr(C), J, I, J, 0, 0, L.stream()) != 0


@pytest.mark.parametrize("R", [832, 3328])
def test_gemm_tn_grouped_matches_single_launches(R):
    """The four weight gradients of a Block in one launch (gpt_gemm_tn_grouped) against fp64 products: different shapes per
    problem, accumulation into existing values, one shared contraction length."""
    import ctypes as C
    d = 256
    shapes = [(d, 4 * d), (4 * d, d), (3 * d, d), (d, d)]            # (I, J) of mlp.2, mlp.0, qkv, proj
    g = torch.Generator().manual_seed(R)
    Ps = [torch.randn(R, i, generator=g).bfloat16().to(DEV) for i, _ in shapes]
    Qs = [torch.randn(R, j, generator=g).bfloat16().to(DEV) for _, j in shapes]
    C0 = [torch.randn(i, j, generator=g).to(DEV) for i, j in shapes]
    Cs = [c.clone() for c in C0]
    n = len(shapes)
    vp, i64, i32 = C.c_void_p * n, C.c_int64 * n, C.c_int * n
    rc = L.lib().gpt_gemm_tn_grouped(n, vp(*[p.data_ptr() for p in Ps]), i64(*[i for i, _ in shapes]),
                                     vp(*[q.data_ptr() for q in Qs]), i64(*[j for _, j in shapes]),
                                     vp(*[c.data_ptr() for c in Cs]), i64(*[j for _, j in shapes]),
                                     i32(*[i for i, _ in shapes]), i32(*[j for _, j in shapes]), R, 1, L.stream())
    L.check(rc, "gpt_gemm_tn_grouped")
    torch.cuda.synchronize()
    for P, Q, c0, c in zip(Ps, Qs, C0, Cs):
        ref = P.double().t() @ Q.double() + c0.double()
        assert float((c.double() - ref).abs().max()) < 2e-3 * float(ref.abs().max())
    # more than 4 problems, or a ragged shape in any of them: rejected before launch
    assert L.lib().gpt_gemm_tn_grouped(5, vp(), i64(), vp(), i64(), vp(), i64(), i32(), i32(), R, 1, L.stream()) != 0
    assert L.lib().gpt_gemm_tn_grouped(n, vp(*[p.data_ptr() for p in Ps]), i64(*[i for i, _ in shapes]),
                                       vp(*[q.data_ptr() for q in Qs]), i64(*[j for _, j in shapes]),
                                       vp(*[c.data_ptr() for c in Cs]), i64(*[j for _, j in shapes]),
                                       i32(d, 4 * d, 3 * d, d + 8), i32(*[j for _, j in shapes]), R, 1, L.stream()) != 0


@pytest.mark.parametrize("B", [32, 5, 37])
def test_wgrad_tn_path_equals_transpose_path(B):
    """The weight gradients through the TN kernel (cifar_small: d = 256; B = 5 / 37: B * T = 130 / 962 rows, ragged last K-tile of
    the contraction, and so for the compact rows of the pruned last Block) equal the transpose + NT route on all rows."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from tests.test_hip_parity import build, DEV
from tests.cases import make_inputs
import gpt_amd
m, plan, sd = build("cifar_small", "bf16", seed=6)
m.train()
diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
B = int(sys.argv[2])
_, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=B, seed=7))
g = torch.Generator().manual_seed(8)
x1 = xp + 0.1 * torch.randn(xp.shape, generator=g).to(DEV)
t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
noise = torch.randn(xp.shape, generator=g).to(DEV)
diff.training_losses(m, x1, t, model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), noise=noise)["loss"].mean().backward()
torch.save({k: p.grad.cpu() for k, p in m.named_parameters()}, sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            path = os.path.join(td, f"g{flag}.pt")
            env = dict(os.environ, GPT_WGRAD_TN=flag)
            subprocess.run([sys.executable, "-c", code, path, str(B)], check=True, env=env, timeout=200)
            outs.append(torch.load(path, weights_only=True))
    # (the TN route also prunes the last Block to the decoded rows, whose GEMMs take other kernel instances: equal to bf16 rounding
    # of the activations there, to fp32 summation order elsewhere)
    for k in outs[0]:
        a, b = outs[0][k].double(), outs[1][k].double()
        if k.endswith("attn.key.bias"):
            continue
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert err < 1.5e-2 and cos > 0.9999, (k, err, cos)


def test_pruned_last_block_training_equals_all_rows():
    """prune_last_train (csrc/model.cpp): the last Block behind its attention, ln_f and the decoder on the decoded tokens' rows alone
    (forward with kept activations AND backward: compact dgrad / LayerNorm backward, scatter before the attention backward, the
    grouped weight-gradient launch with one contraction length per problem) against the same step on all rows
    (GPT_PRUNE_LAST_TRAIN=0, a process of its own: the switch is read once): loss and every gradient, store mode and accumulate mode."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from tests.test_hip_parity import build, DEV
from tests.cases import make_inputs
import gpt_amd
from gpt_amd.trainstep import TrainStep
out = {}
for tag, B in (("store", 64), ("accum", 128), ("store_ragged", 37)):      # 37 * 26 = 962 rows, 37 * 11 = 407 compact: no multiple of 64
    m, plan, sd = build("cifar_small", "bf16", seed=6)
    m.train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("cifar_small", B=B, seed=7))
    g = torch.Generator().manual_seed(8)
    x1 = xp + 0.1 * torch.randn(xp.shape, generator=g).to(DEV)
    t = torch.randint(0, 1000, (B,), generator=g).to(DEV)
    noise = torch.randn(xp.shape, generator=g).to(DEV)
    if tag.startswith("store"):
        ts = TrainStep(m, diff, lr=1e-3, wd=0.1, grad_clip=0.75, use_ema=False)
        out[tag + ".mse"] = ts.loss_and_grads(xp, x1, lp, lt, t, noise=noise).cpu()
        assert ts._overwrote
    else:                                  # the autograd bridge accumulates (float atomics over row spans)
        terms = diff.training_losses(m, x1, t, model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), noise=noise)
        terms["loss"].mean().backward()
        out[tag + ".mse"] = terms["mse"].detach().cpu()
    for k, p in m.named_parameters():
        out[tag + "." + k] = p.grad.detach().cpu().clone()
torch.save(out, sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            path = os.path.join(td, f"g{flag}.pt")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, GPT_PRUNE_LAST_TRAIN=flag), timeout=250)
            outs.append(torch.load(path, weights_only=True))
    for k in outs[0]:
        a, b = outs[0][k].double(), outs[1][k].double()
        if k.endswith(".mse"):
            assert float(((a - b).abs() / b.abs()).max()) < 5e-3, k
            continue
        if k.endswith("attn.key.bias"):
            continue
        err = float((a - b).abs().max() / (b.abs().max() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        assert err < 1.5e-2 and cos > 0.9999, (k, err, cos)


def _native_dp_rank(out_path):
    """gpt_dp_* (include/gpt_hip.h) with the one rank a one-GPU box allows: every collective really runs on RCCL; with one rank
    each is the identity, so values are checked exactly."""
    import ctypes as C
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)
    torch.cuda.set_device(0)
    lib, st = L.lib(), L.stream()
    assert lib.gpt_dp_allreduce_grads(None, 4, st) < 0 and b"gpt_dp_init first" in lib.gpt_last_error()
    uid = (C.c_char * L.DP_ID_BYTES)()
    L.check(lib.gpt_dp_unique_id(uid), "gpt_dp_unique_id")
    assert lib.gpt_dp_init(uid, 1, 1) < 0                          # rank out of range: rejected before RCCL sees it
    L.check(lib.gpt_dp_init(uid, 0, 1), "gpt_dp_init")
    assert lib.gpt_dp_rank() == 0 and lib.gpt_dp_world() == 1
    assert lib.gpt_dp_init(uid, 0, 1) < 0                          # one communicator per process
    g = torch.arange(1000, dtype=torch.float32, device=DEV)
    L.check(lib.gpt_dp_allreduce_grads(L.ptr(g), g.numel(), st), "allreduce")
    rs = torch.zeros(1000, device=DEV)
    L.check(lib.gpt_dp_reduce_scatter(L.ptr(g), L.ptr(rs), 1000, L.F32, st), "reduce_scatter")
    h = torch.arange(64, device=DEV).bfloat16()
    ag = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    L.check(lib.gpt_dp_allgather(L.ptr(h), L.ptr(ag), 64, L.BF16, st), "allgather")
    L.check(lib.gpt_dp_allgather(L.ptr(h), L.ptr(h), 64, L.BF16, st), "allgather in place")
    t = torch.arange(7, device=DEV)
    L.check(lib.gpt_dp_broadcast(L.ptr(t), 7, L.DP_I64, 0, st), "broadcast")
    assert lib.gpt_dp_broadcast(L.ptr(t), 7, 5, 0, st) < 0          # unknown dtype
    L.check(lib.gpt_dp_barrier(st), "barrier")                      # synchronises the stream
    ok = (torch.equal(g, torch.arange(1000, dtype=torch.float32, device=DEV)) and torch.equal(rs, g) and torch.equal(ag, h)
          and torch.equal(t, torch.arange(7, device=DEV)))
    L.check(lib.gpt_dp_destroy(), "destroy")
    assert lib.gpt_dp_world() == 1 and lib.gpt_dp_destroy() == 0
    torch.save({"ok": ok}, out_path)


def test_native_rccl_entry_points_one_rank(tmp_path):
    """The C-ABI collectives (dp_init / dp_allreduce_grads / dp_allgather / dp_broadcast / dp_barrier /
    dp_destroy) on RCCL itself, loaded with dlopen by the library."""
    import torch.multiprocessing as mp
    out_path = str(tmp_path / "dp.pt")
    p = mp.get_context("spawn").Process(target=_native_dp_rank, args=(out_path,))
    p.start()
    p.join(timeout=150)
    if p.exitcode is None:
        p.kill()
    assert p.exitcode == 0
    assert torch.load(out_path, weights_only=True)["ok"]
