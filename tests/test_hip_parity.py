"""GPU parity tests: the HIP path, called through the C ABI, against the oracle and the committed golden
fixtures (which were produced by the reference itself).  Run on the MI355X box: pytest -m gpu.

Tolerances.  fp32 mode: BASELINE.json asks for 1e-3 relative error on the denoiser output; the tests
hold the fp32 path to 2e-4 relative-to-scale.  bf16 mode is a throughput mode (bf16 operands, fp32
accumulate): held to 3e-2 of the output scale."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import denoiser as od          # noqa: E402
from oracle import diffusion as odf        # noqa: E402
from oracle import permute as op           # noqa: E402
from tests.cases import CASES, block_case, gpt_kwargs, make_inputs, step_noise  # noqa: E402

import gpt_amd                              # noqa: E402
from gpt_amd import _lib as L               # noqa: E402

DEV = "cuda:0"


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def plan_of(case, predict_xstart=True):
    c = CASES[case]
    return od.Plan(c["parameter_sizes"], 128, 14, True, c["n_head"], c["n_embd"], c["n_layer"],
                   c["chunk_size"], c["split_policy"], predict_xstart)


def build(case, dtype="fp32", seed=0):
    plan = plan_of(case)
    sd = od.random_state_dict(plan, seed=seed)
    m = gpt_amd.Gpt(**gpt_kwargs(case))
    m.load_state_dict(sd)
    return m.to(DEV).eval().set_compute_dtype(dtype), plan, sd


# ------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 1e-2)])
@pytest.mark.parametrize("M,N,K", [(104, 192, 192), (104, 576, 192), (300, 768, 192), (128, 128, 64), (1664, 1536, 1536),
                                   (77, 136, 256),
                                   # the 256 x 256 kernel: 224-row tiles with a ragged last row tile (8320 = 37.1 x 224),
                                   # 256-row tiles filling whole rounds, and a row tail handed to the ring kernels
                                   (8320, 1536, 256), (4096, 4096, 128), (8320, 6144, 128),
                                   # 128 x 256 tiles (gemm128.hip: few 256-row tiles, one 128-row tile per CU), also ragged in M and N
                                   (1664, 4608, 192), (1650, 4600, 192),
                                   # one round of 256-row tiles + a row tail that is one round of the 128 x 256 kernel
                                   (3328, 6144, 128), (3300, 6136, 128),
                                   # one round of 192-row tiles (216 of them; B = 64 sampling's fc1), also ragged in M and N
                                   (1664, 6144, 128), (1650, 6136, 192)])
def test_gemm_epilogues(dtype, tol, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    code = L.dtype_code(dtype)
    td = L.torch_dtype(code)
    Ad, Wd = A.to(DEV).to(td), W.to(DEV).to(td)
    Af, Wf = Ad.float().cpu(), Wd.float().cpu()              # what the kernel actually sees
    ref = Af.double() @ Wf.double().t() + bias.double()
    lib = L.lib()

    bias_d = bias.to(DEV)

    def run(epi, Cbuf, aux=None, ldaux=0, use_bias=True):
        L.check(lib.gpt_gemm_nt(code, L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(bias_d) if use_bias else None, L.ptr(Cbuf), N,
                                L.ptr(aux), ldaux, M, N, K, epi, L.stream()), "gemm")
        torch.cuda.synchronize()

    out = torch.empty(M, N, dtype=td, device=DEV)
    run(L.EPI_STORE, out)
    assert rel_err(out.float().cpu(), ref) < tol
    run(L.EPI_GELU, out)
    assert rel_err(out.float().cpu(), torch.nn.functional.gelu(ref)) < tol
    acc = R.to(DEV).clone()
    run(L.EPI_RESADD, acc)
    assert rel_err(acc.cpu(), R.double() + ref) < tol
    acc2, Rd = torch.empty(M, N, device=DEV), R.to(DEV)
    run(L.EPI_RESADD, acc2, aux=Rd, ldaux=N)
    assert rel_err(acc2.cpu(), R.double() + ref) < tol
    f32o = torch.empty(M, N, device=DEV)
    run(L.EPI_STORE_F32, f32o, use_bias=False)
    assert rel_err(f32o.cpu(), ref - bias.double()) < tol
    pre = torch.empty(M, N, dtype=td, device=DEV)
    run(L.EPI_GELU_AUX, out, aux=pre, ldaux=N)
    assert rel_err(pre.float().cpu(), ref) < tol
    assert rel_err(out.float().cpu(), torch.nn.functional.gelu(ref)) < tol


@pytest.mark.parametrize("M,N,K", [(1664, 1536, 6144), (1700, 1528, 6464), (3328, 1536, 4608), (520, 264, 2048),
                                   # one round of 256-row tiles + a 128-row tail that the ring kernel contracts in 2 K spans
                                   (2176, 8192, 768)])
def test_gemm_split_k_workspace(M, N, K):
    """bf16 problems of few 256 x 256 tiles and a long K go through K spans + a reduce pass once a workspace is registered
    (gpt_gemm_set_workspace): every epilogue against fp64, ragged M / N / K-span counts included, and the output after
    un-registering (single-pass kernels) must agree to bf16 rounding."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Ad, Wd = A.to(DEV).bfloat16(), W.to(DEV).bfloat16()
    ref = Ad.float().cpu().double() @ Wd.float().cpu().double().t() + bias.double()
    lib, bias_d, tol = L.lib(), bias.to(DEV), 1e-2
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)

    def run(epi, Cbuf, aux=None, ldaux=0, use_bias=True):
        L.check(lib.gpt_gemm_nt(L.BF16, L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(bias_d) if use_bias else None, L.ptr(Cbuf), N,
                                L.ptr(aux), ldaux, M, N, K, epi, L.stream()), "gemm")
        torch.cuda.synchronize()

    gelu = torch.nn.functional.gelu
    try:
        L.check(lib.gpt_gemm_set_workspace(L.ptr(ws), ws.numel()), "ws")
        out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        run(L.EPI_STORE, out)
        split = out.clone()
        assert rel_err(out.float().cpu(), ref) < tol
        run(L.EPI_GELU, out)
        assert rel_err(out.float().cpu(), gelu(ref)) < tol
        acc = R.to(DEV).clone()
        run(L.EPI_RESADD, acc)
        assert rel_err(acc.cpu(), R.double() + ref) < 1e-4          # fp32 partials, fp32 output
        acc2, Rd = torch.empty(M, N, device=DEV), R.to(DEV)
        run(L.EPI_RESADD, acc2, aux=Rd, ldaux=N)
        assert torch.equal(acc2, acc)
        f32o = torch.empty(M, N, device=DEV)
        run(L.EPI_STORE_F32, f32o, use_bias=False)
        assert rel_err(f32o.cpu(), ref - bias.double()) < 1e-4
        accum = R.to(DEV).clone()
        run(L.EPI_ACCUM_F32, accum, use_bias=False)
        assert rel_err(accum.cpu(), R.double() + ref - bias.double()) < 1e-4
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        run(L.EPI_GELU_AUX, out, aux=pre, ldaux=N)
        assert torch.equal(pre, split)
        assert rel_err(out.float().cpu(), gelu(ref)) < tol
        z = (torch.randn(M, N, generator=g) * 1.5).to(DEV).bfloat16()
        run(L.EPI_DGELU, out, aux=z, ldaux=N)
        zd = z.float().cpu().double().requires_grad_(True)
        gelu(zd).sum().backward()
        assert rel_err(out.float().cpu(), ref * zd.grad) < tol
    finally:
        L.check(lib.gpt_gemm_set_workspace(None, 0), "ws")
    single = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    run(L.EPI_STORE, single)
    d = (single.float() - split.float()).abs()
    assert bool((d <= split.float().abs() * 2.0 ** -7 + 1e-5).all()), float(d.max())
    L._gemm_ws.pop(torch.device(DEV).index, None)     # the engines' registration was replaced above: let them renew it


@pytest.mark.parametrize("M,N,K,inplace", [(1664, 1536, 1536, True), (1664, 1536, 6144, False), (3328, 1536, 1536, False),
                                            (1700, 2048, 2048, True), (8320, 1536, 1536, True), (520, 264, 2048, False),
                                            # configs[4] (n_embd 3072) at its per-GPU batch 68: M = 1768, proj and mlp.2
                                            (1768, 3072, 3072, True), (1768, 3072, 12288, False)])
def test_gemm_resadd_layernorm(M, N, K, inplace):
    """Residual Linear + the LayerNorm behind it in one call: against fp64, and against the two separate entry points (the
    fused K-span pass at small batches, the two-kernel route at M = 8320 and for widths the fused pass does not cover)."""
    g = torch.Generator().manual_seed(M + N + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05
    bias, R = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    gam, bet = 1 + 0.1 * torch.randn(N, generator=g), 0.1 * torch.randn(N, generator=g)
    Ad, Wd = A.to(DEV).bfloat16(), W.to(DEV).bfloat16()
    xref = R.double() + Ad.float().cpu().double() @ Wd.float().cpu().double().t() + bias.double()
    yref = torch.nn.functional.layer_norm(xref, (N,), gam.double(), bet.double(), 1e-5)
    lib, bias_d, gd, bd = L.lib(), bias.to(DEV), gam.to(DEV), bet.to(DEV)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    try:
        L.check(lib.gpt_gemm_set_workspace(L.ptr(ws), ws.numel()), "ws")
        res = R.to(DEV).clone()
        x = res if inplace else torch.empty(M, N, device=DEV)
        y = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        L.check(lib.gpt_gemm_resadd_layernorm(L.BF16, L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(bias_d), L.ptr(res), N, L.ptr(x), N,
                                              L.ptr(gd), L.ptr(bd), L.ptr(y), L.ptr(mean), L.ptr(rstd), M, N, K, 1e-5,
                                              L.stream()), "fused")
        torch.cuda.synchronize()
        assert rel_err(x.cpu(), xref) < 1e-4                       # fp32 partials, fp32 residual stream
        assert rel_err(y.float().cpu(), yref) < 1e-2
        np.testing.assert_allclose(mean.cpu().numpy(), xref.mean(1).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(rstd.cpu().numpy(), (1 / torch.sqrt(xref.var(1, unbiased=False) + 1e-5)).numpy(), rtol=1e-4)
        # the two entry points it stands for
        x2 = R.to(DEV).clone()
        y2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        L.check(lib.gpt_gemm_nt(L.BF16, L.ptr(Ad), K, L.ptr(Wd), K, L.ptr(bias_d), L.ptr(x2), N, None, 0, M, N, K, L.EPI_RESADD,
                                L.stream()), "gemm")
        L.check(lib.gpt_layernorm(L.ptr(x2), L.ptr(gd), L.ptr(bd), None, 0, L.ptr(y2), L.BF16, None, None, M, N, 1e-5,
                                  L.stream()), "ln")
        torch.cuda.synchronize()
        assert rel_err(x.cpu(), x2.cpu().double()) < 1e-5
        dy = (y.float() - y2.float()).abs()
        assert bool((dy <= y2.float().abs() * 2.0 ** -7 + 1e-3).all()), float(dy.max())
    finally:
        L.check(lib.gpt_gemm_set_workspace(None, 0), "ws")
        L._gemm_ws.pop(torch.device(DEV).index, None)


def test_gemm_gelu_bf16_range():
    """bf16 GELU epilogue (x * sigmoid(fitted odd polynomial)) against the exact erf GELU over [-30, 30], outliers included:
    error <= one bf16 rounding of the result + 2.6e-5 (the fit's bound, gemm_common.h)."""
    M, N, K = 4096, 64, 64
    x = torch.cat([torch.linspace(-30, 30, M - 6), torch.tensor([-1e4, 1e4, -11.5, 11.5, 0.0, -0.0])]).bfloat16()
    A = torch.zeros(M, K, dtype=torch.bfloat16)
    A[:, 0] = x
    W = torch.zeros(N, K, dtype=torch.bfloat16)
    W[:, 0] = 1.0
    Ad, Wd = A.to(DEV), W.to(DEV)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    L.check(L.lib().gpt_gemm_nt(L.BF16, L.ptr(Ad), K, L.ptr(Wd), K, None, L.ptr(out), N, None, 0, M, N, K, L.EPI_GELU,
                                L.stream()), "gemm")
    ref = torch.nn.functional.gelu(x.double())[:, None].expand(M, N)
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= ref.abs() * 2.0 ** -8 + 2.6e-5).all()), float(err.max())


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(8, 40, device=DEV)
    st = L.lib().gpt_gemm_nt(L.F32, L.ptr(a), 40, L.ptr(a), 40, None, L.ptr(a), 8, None, 0, 8, 8, 40, 0, L.stream())
    assert st < 0 and b"multiple" in L.lib().gpt_last_error()


@pytest.mark.parametrize("d", [192, 256, 1536, 2048, 520])
@pytest.mark.parametrize("out_dtype", ["fp32", "bf16"])
def test_layernorm(d, out_dtype):
    g = torch.Generator().manual_seed(d)
    rows = 53
    x = torch.randn(rows, d, generator=g) * 3 + 0.5
    w, b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    pos = torch.randn(26, d, generator=g)
    ref = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5)
    code = L.dtype_code(out_dtype)
    y = torch.empty(rows, d, dtype=L.torch_dtype(code), device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    xd, wd, bd, pd = x.to(DEV), w.to(DEV), b.to(DEV), pos.to(DEV)       # keep the device tensors alive
    L.check(L.lib().gpt_layernorm(L.ptr(xd), L.ptr(wd), L.ptr(bd), None, 0, L.ptr(y), code,
                                  L.ptr(mean), L.ptr(rstd), rows, d, 1e-5, L.stream()), "ln")
    tol = 1e-5 if out_dtype == "fp32" else 1e-2
    assert rel_err(y.float().cpu(), ref) < tol
    np.testing.assert_allclose(mean.cpu().numpy(), x.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rstd.cpu().numpy(), (1 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)).numpy(), rtol=1e-5)
    L.check(L.lib().gpt_layernorm(L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(pd), 26, L.ptr(y),
                                  code, None, None, rows, d, 1e-5, L.stream()), "ln+add")
    ref2 = ref + pos[torch.arange(rows) % 26]
    assert rel_err(y.float().cpu(), ref2) < tol


@pytest.mark.parametrize("d", [1536, 2048])
@pytest.mark.parametrize("out_dtype", ["fp32", "bf16"])
def test_layernorm_many_rows(d, out_dtype):
    """From 4096 rows up the widths of the two transformers take the two-rows-per-wave kernel (norm.hip layernorm2_kernel); an odd
    row count leaves its last wave with one row.  Same arithmetic per row as the one-row kernel: the small-count call on the
    first rows must give the same bits."""
    g = torch.Generator().manual_seed(d + 1)
    rows = 4099
    x = torch.randn(rows, d, generator=g) * 2 - 0.25
    w, b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    ref = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5)
    code = L.dtype_code(out_dtype)
    td = L.torch_dtype(code)
    y = torch.full((rows + 1, d), 7.0, dtype=td, device=DEV)                 # one guard row behind the output
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    L.check(L.lib().gpt_layernorm(L.ptr(xd), L.ptr(wd), L.ptr(bd), None, 0, L.ptr(y), code,
                                  L.ptr(mean), L.ptr(rstd), rows, d, 1e-5, L.stream()), "ln")
    assert rel_err(y[:rows].float().cpu(), ref) < (1e-5 if out_dtype == "fp32" else 1e-2)
    assert bool((y[rows].float() == 7.0).all())
    np.testing.assert_allclose(mean.cpu().numpy(), x.mean(1).numpy(), rtol=1e-5, atol=1e-6)
    y1 = torch.empty(100, d, dtype=td, device=DEV)
    L.check(L.lib().gpt_layernorm(L.ptr(xd), L.ptr(wd), L.ptr(bd), None, 0, L.ptr(y1), code, None, None, 100, d, 1e-5,
                                  L.stream()), "ln")
    assert torch.equal(y1, y[:100])
    y2 = torch.empty(3, d, dtype=td, device=DEV)                             # the last rows, one-row kernel
    tail = xd[rows - 3:].contiguous()
    L.check(L.lib().gpt_layernorm(L.ptr(tail), L.ptr(wd), L.ptr(bd), None, 0, L.ptr(y2), code, None, None, 3, d, 1e-5,
                                  L.stream()), "ln")
    assert torch.equal(y2, y[rows - 3:rows])


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("H,hd,T", [(2, 96, 26), (2, 128, 26), (16, 96, 26), (3, 32, 18), (2, 64, 32), (1, 96, 5),
                                    # T = 32 fills the per-wave LDS tile the output is staged through
                                    (3, 96, 32), (2, 128, 32), (5, 32, 32), (1, 64, 1)])
def test_attention(dtype, tol, H, hd, T):
    g = torch.Generator().manual_seed(H * hd + T)
    B, d = 5, H * hd
    code = L.dtype_code(dtype)
    td = L.torch_dtype(code)
    qkv = (torch.randn(B * T, 3 * d, generator=g)).to(DEV).to(td)
    f = qkv.float().cpu().view(B, T, 3, H, hd)
    q, k, v = (f[:, :, i].transpose(1, 2).double() for i in range(3))
    att = torch.softmax(q @ k.transpose(-2, -1) / np.sqrt(hd), dim=-1)
    ref = (att @ v).transpose(1, 2).reshape(B * T, d)
    y = torch.empty(B * T, d, dtype=td, device=DEV)
    L.check(L.lib().gpt_attention(code, L.ptr(qkv), L.ptr(y), B, T, H, hd, L.stream()), "attention")
    assert rel_err(y.float().cpu(), ref) < tol


def test_scalar_tokens_golden(golden):
    g = golden("g2_freq")
    emb = gpt_amd.FrequencyEmbedder(128, 14).to(DEV)
    assert np.array_equal(emb.frequencies.cpu().numpy(), g["frequencies"])
    # sin/cos of arguments up to 1.6e7 rad: device libm vs host libm may differ in the last ulp of the result
    np.testing.assert_allclose(emb(torch.from_numpy(g["t"]).to(DEV)).cpu().numpy(), g["t_emb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(emb(torch.from_numpy(g["l32"]).to(DEV)).cpu().numpy(), g["l32_emb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(emb(torch.from_numpy(g["l64"]).to(DEV)).cpu().numpy(), g["l64_emb"], rtol=0, atol=2e-6)


# ------------------------------------------------------------------------------------------ denoiser
@pytest.mark.parametrize("case", ["mnist_small", "cifar_small", "cartpole_small"])
def test_forward_golden_fp32(golden, case):
    g = golden(f"g3_forward_{case}")
    m, plan, sd = build(case, "fp32")
    x, t, lt, lp, xp = (v.to(DEV) for v in make_inputs(case, B=4, seed=11))
    with torch.no_grad():
        out = m(x, t, lt, loss_prev=lp, x_prev=xp)
        out_col = m(x, t, lt.view(-1, 1).double(), loss_prev=lp.view(-1, 1).double(), x_prev=xp)
    # error relative to the scale of the network's own contribution (out - x_prev), the strict reading
    delta_scale = np.abs(g["out"] - xp.cpu().numpy()).max()
    assert np.abs(out.cpu().numpy() - g["out"]).max() / delta_scale < 2e-4
    assert np.abs(out_col.cpu().numpy() - g["out_col64"]).max() / delta_scale < 2e-4
    # intermediates from the workspace: pre-decode features of the tokens the decoder reads (the inference forward runs the last
    # Block behind its attention on those rows alone and leaves them compacted: [B, n_out_tok, d])
    To = m.decoder.num_output_heads
    hf = m.engine().buffers()["hf"].float().cpu().view(-1)[:4 * To * plan.n_embd].view(4, To, plan.n_embd)
    assert rel_err(hf, g["ln_f"][:, :To]) < 2e-4


@pytest.mark.parametrize("case", ["mnist", "cifar"])
def test_forward_full_size_vs_oracle(case):
    """north_star check 1 at the pretrained shapes (d = 1536 / 2048, 12 layers): fp32 denoiser output within 1e-3 of the
    oracle, relative to the network's own contribution; B = 64 rows so that the 256 x 256 GEMM kernel, its edge tiles and
    the ring kernels all take part.  bf16 on the same inputs within bf16 rounding of fp32."""
    m, plan, sd = build(case, "fp32", seed=2)
    cpu = make_inputs(case, B=64, seed=13)
    dev = tuple(v.to(DEV) for v in cpu)
    with torch.no_grad():
        out = m(dev[0], dev[1], dev[2], loss_prev=dev[3], x_prev=dev[4]).cpu()
        ref = od.forward(sd, plan, *cpu)
    scale = float((ref - cpu[4]).abs().max())
    assert float((out - ref).abs().max()) / scale < 1e-3
    m.set_compute_dtype("bf16")
    with torch.no_grad():
        out16 = m(dev[0], dev[1], dev[2], loss_prev=dev[3], x_prev=dev[4]).cpu()
    assert float((out16 - ref).abs().max()) / scale < 3e-2


@pytest.mark.parametrize("B", [1, 3, 7, 37])
def test_forward_full_size_ragged_batches_vs_oracle(B):
    """The pretrained mnist shapes at batches far from the ones the kernels were tuned on (26 ... 962 rows: edge tiles of every GEMM
    plan, the 64-column encoder / decoder workgroups, the paired input LayerNorms, the pruned last Block on 11 ... 407 compact
    rows): fp32 within 1e-3 of the oracle, bf16 within its rounding."""
    m, plan, sd = build("mnist", "fp32", seed=5)
    cpu = make_inputs("mnist", B=B, seed=40 + B)
    dev = tuple(v.to(DEV) for v in cpu)
    with torch.no_grad():
        out = m(dev[0], dev[1], dev[2], loss_prev=dev[3], x_prev=dev[4]).cpu()
        ref = od.forward(sd, plan, *cpu)
    scale = float((ref - cpu[4]).abs().max())
    assert float((out - ref).abs().max()) / scale < 1e-3
    m.set_compute_dtype("bf16")
    with torch.no_grad():
        out16 = m(dev[0], dev[1], dev[2], loss_prev=dev[3], x_prev=dev[4]).cpu()
    assert float((out16 - ref).abs().max()) / scale < 3e-2


def test_paired_input_layernorms_are_bit_identical_to_two_launches():
    """The inference forward runs ln_in + position embedding and the first Block's ln1 as one launch at the full-size widths
    (gpt_layernorm_pair, csrc/norm.hip; transformer.py:226, 308-309 and Block.ln_1): same arithmetic in the same order, so
    the denoiser output must equal, bit for bit, the one from two launches (GPT_LN_FUSE0=0, a process of its own: the switch is
    read once).  B = 5 takes the one-row-per-wave second LayerNorm, B = 320 (8320 rows) the two-rows-per-wave one."""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from tests.test_hip_parity import build, DEV
from tests.cases import make_inputs
out = {}
for case in ("mnist", "cifar"):
    for dtype in ("fp32", "bf16"):
        m, plan, sd = build(case, dtype, seed=3)
        for B in (5, 320):
            x, t, lt, lp, xp = (v.to(DEV) for v in make_inputs(case, B=B, seed=B))
            with torch.no_grad():
                out["%%s.%%s.%%d" %% (case, dtype, B)] = m(x, t, lt, lp, xp).float().cpu()
torch.save(out, sys.argv[1])
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("1", "0"):
            path = os.path.join(td, f"f{flag}.pt")
            subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, GPT_LN_FUSE0=flag), timeout=280)
            outs.append(torch.load(path, weights_only=True))
    assert len(outs[0]) == 8
    for k in outs[0]:
        assert torch.isfinite(outs[0][k]).all(), k
        assert torch.equal(outs[0][k], outs[1][k]), (k, float((outs[0][k] - outs[1][k]).abs().max()))


@pytest.mark.parametrize("sizes,policy,chunk,absolute,n_embd,n_head", [
    ([7840, 10, 100, 10], "layer_by_layer", None, True, 128, 4),     # 12 tokens, one of them 7840 wide (the cartpole policy)
    ([432, 4608, 320, 10], "chunk_within_input", 1000, False, 192, 2),  # delta-loss conditioning only: 2 scalar tokens
    ([7840, 10, 100, 10], "chunk", 1990, True, 128, 2),                 # 'chunk': 1990 divides D = 7960, nothing straddles x | x_prev
    ([64, 8, 24], "layer_by_layer", None, True, 64, 2)])                # a tiny net: every token narrower than a K chunk
def test_forward_other_split_policies_vs_oracle(sizes, policy, chunk, absolute, n_embd, n_head):
    """Gpt.forward under the reference's other tokenisation policies (GPT.build_splits, transformer.py:251-300) and with
    delta-only loss conditioning: HIP fp32 path against the oracle (pinned for the token plans by fixture g1)."""
    names = [f"p{i}" for i in range(len(sizes))]
    plan = od.Plan(sizes, 128, 14, absolute, n_head, n_embd, 2, chunk, policy, True)
    sd = od.random_state_dict(plan, seed=4)
    m = gpt_amd.Gpt(parameter_sizes=list(sizes), parameter_names=names, num_frequencies=128, max_freq_log2=14, predict_xstart=True,
                    absolute_loss_conditioning=absolute, n_embd=n_embd, n_head=n_head, n_layer=2, chunk_size=chunk,
                    split_policy=policy, encoder_depth=1, decoder_depth=1, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    m.load_state_dict(sd)
    m = m.to(DEV).eval().set_compute_dtype("fp32")
    D, B = sum(sizes), 5
    g = torch.Generator().manual_seed(D)
    xp, x = torch.randn(B, D, generator=g) * 0.5, torch.randn(B, D, generator=g)
    t = torch.tensor([999, 0, 1, 500, 37])
    lp, lt = 0.3 + 2 * torch.rand(B, generator=g), torch.zeros(B)
    with torch.no_grad():
        ref = od.forward(sd, plan, x, t, lt, lp, xp)
        got = m(x.to(DEV), t.to(DEV), lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV)).cpu()
    assert got.shape == (B, D)
    assert float((got - ref).abs().max()) / float((ref - xp).abs().max()) < 1e-3


@pytest.mark.parametrize("B", [1, 3, 33, 65])
def test_forward_edge_batches_and_argument_forms(B):
    """Batches of one row and of row counts no tile divides; the argument forms the reference accepts (transformer.py:440-464,
    SURVEY.md 8a-5): losses as (B,) or (B, 1), fp32 or fp64, t as int64 or int32; an empty batch returns an empty tensor."""
    m, plan, sd = build("mnist_small", "fp32", seed=3)
    cpu = make_inputs("mnist_small", B=B, seed=60 + B)
    x, t, lt, lp, xp = cpu
    with torch.no_grad():
        ref = od.forward(sd, plan, *cpu)
        scale = float((ref - xp).abs().max())
        got = m(x.to(DEV), t.to(DEV), lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
        assert got.shape == (B, plan.D) and float((got.cpu() - ref).abs().max()) / scale < 1e-3
        alt = m(x.to(DEV), t.int().to(DEV), lt.view(B, 1).to(DEV), loss_prev=lp.view(B, 1).to(DEV), x_prev=xp.to(DEV))
        assert torch.equal(alt, got)
        ref64 = od.forward(sd, plan, x, t, lt.double(), lp.double(), xp)
        got64 = m(x.to(DEV), t.to(DEV), lt.double().to(DEV), loss_prev=lp.double().to(DEV), x_prev=xp.to(DEV))
        assert float((got64.cpu() - ref64).abs().max()) / scale < 1e-3
        empty = m(x[:0].to(DEV), t[:0].to(DEV), lt[:0].to(DEV), loss_prev=lp[:0].to(DEV), x_prev=xp[:0].to(DEV))
        assert empty.shape == (0, plan.D)
    with pytest.raises(AssertionError):           # shape contracts fail before any launch
        m(x.to(DEV), t[:-1].to(DEV) if B > 1 else torch.zeros(2, dtype=torch.long, device=DEV), lt.to(DEV), loss_prev=lp.to(DEV),
          x_prev=xp.to(DEV))
    with pytest.raises(AssertionError):
        m(x[:, :-1].to(DEV), t.to(DEV), lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp[:, :-1].to(DEV))


@pytest.mark.parametrize("case", ["mnist_small", "cifar_small", "cartpole_small"])
def test_forward_oracle_bf16(case):
    m, plan, sd = build(case, "bf16", seed=3)
    cpu = make_inputs(case, B=7, seed=5)
    x, t, lt, lp, xp = (v.to(DEV) for v in cpu)
    with torch.no_grad():
        out = m(x, t, lt, loss_prev=lp, x_prev=xp)
        ref = od.forward(sd, plan, *cpu)
    delta = (ref - cpu[4]).numpy()
    assert np.abs(out.cpu().numpy() - ref.numpy()).max() / np.abs(delta).max() < 3e-2


@pytest.mark.parametrize("name,d,H", [("mnist", 1536, 16), ("cifar", 2048, 16)])
def test_full_width_block_golden(golden, name, d, H):
    """One full-width Block (transformer.py:54-72) assembled from the per-kernel C ABI, fp32."""
    g = golden("g4_block")
    sd, xin = block_case(d)
    w = {k: v.to(DEV) for k, v in sd.items()}
    lib, s = L.lib(), L.stream()
    M = xin.shape[0] * xin.shape[1]
    x = xin.reshape(M, d).to(DEV).clone()
    h = torch.empty(M, d, device=DEV)
    wqkv = torch.cat([w["attn.query.weight"], w["attn.key.weight"], w["attn.value.weight"]]).contiguous()
    bqkv = torch.cat([w["attn.query.bias"], w["attn.key.bias"], w["attn.value.bias"]]).contiguous()
    qkv, att, act = torch.empty(M, 3 * d, device=DEV), torch.empty(M, d, device=DEV), torch.empty(M, 4 * d, device=DEV)
    L.check(lib.gpt_layernorm(L.ptr(x), L.ptr(w["ln1.weight"]), L.ptr(w["ln1.bias"]), None, 0, L.ptr(h), 0, None, None, M, d, 1e-5, s), "ln1")
    L.check(lib.gpt_gemm_nt(0, L.ptr(h), d, L.ptr(wqkv), d, L.ptr(bqkv), L.ptr(qkv), 3 * d, None, 0, M, 3 * d, d, L.EPI_STORE, s), "qkv")
    L.check(lib.gpt_attention(0, L.ptr(qkv), L.ptr(att), 2, 26, H, d // H, s), "att")
    L.check(lib.gpt_gemm_nt(0, L.ptr(att), d, L.ptr(w["attn.proj.weight"]), d, L.ptr(w["attn.proj.bias"]), L.ptr(x), d, None, 0, M, d, d, L.EPI_RESADD, s), "proj")
    L.check(lib.gpt_layernorm(L.ptr(x), L.ptr(w["ln2.weight"]), L.ptr(w["ln2.bias"]), None, 0, L.ptr(h), 0, None, None, M, d, 1e-5, s), "ln2")
    L.check(lib.gpt_gemm_nt(0, L.ptr(h), d, L.ptr(w["mlp.0.weight"]), d, L.ptr(w["mlp.0.bias"]), L.ptr(act), 4 * d, None, 0, M, 4 * d, d, L.EPI_GELU, s), "fc1")
    L.check(lib.gpt_gemm_nt(0, L.ptr(act), 4 * d, L.ptr(w["mlp.2.weight"]), 4 * d, L.ptr(w["mlp.2.bias"]), L.ptr(x), d, None, 0, M, d, 4 * d, L.EPI_RESADD, s), "fc2")
    y = x.cpu().view(2, 26, d).numpy()
    assert np.abs(y - g[f"{name}_y"]).max() / np.abs(g[f"{name}_y"] - xin.numpy()).max() < 2e-4


# ------------------------------------------------------------------------------------------ diffusion
@pytest.mark.parametrize("D", [7960, 5370, 33])
def test_q_sample_and_posterior_bit_exact(D):
    """The elementwise DDPM kernels reproduce the oracle's fp32 arithmetic bit for bit (same roundings)."""
    B = 6
    g = torch.Generator().manual_seed(D)
    x0, nz, xt, mo = (torch.randn(B, D, generator=g) for _ in range(4))
    t = torch.tensor([0, 1, 17, 500, 998, 999])
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    tb = odf.Tables(1000)
    got = diff.q_sample(x0.to(DEV), t.to(DEV), noise=nz.to(DEV)).cpu()
    assert torch.equal(got, odf.q_sample(tb, x0, t, nz))
    for px, clamp in ((True, None), (False, None), (True, (-0.5, 0.7)), (False, (-1.0, 1.0))):
        diff.model_mean_type = gpt_amd.ModelMeanType.START_X if px else gpt_amd.ModelMeanType.EPSILON
        out, x0p = diff._posterior(mo.to(DEV), xt.to(DEV), nz.to(DEV), t.to(DEV), clamp, want_xstart=True)
        ref, ref0 = odf.posterior_step(tb, mo, xt, t, nz, px, False, *(clamp or (None, None)))
        np.testing.assert_allclose(x0p.cpu().numpy(), ref0.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)
    tg, o = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    mse, tgd, od_ = torch.empty(B, device=DEV), tg.to(DEV), o.to(DEV)
    L.check(L.lib().gpt_mse_rows(L.ptr(tgd), L.ptr(od_), L.ptr(mse), B, D, L.stream()), "mse")
    np.testing.assert_allclose(mse.cpu().numpy(), ((tg - o) ** 2).mean(1).numpy(), rtol=1e-5)


@pytest.mark.parametrize("case,resp,tag,px,clip", [
    ("mnist_small", "50", "r50", True, False), ("cifar_small", "50", "r50", True, False),
    ("mnist_small", "50", "r50_eps_clip", False, True), ("mnist_small", "", "full", True, False)])
def test_p_sample_loop_golden(golden, case, resp, tag, px, clip):
    """Whole reverse loop in one host call, injected noise, against the reference's trajectory end point."""
    g = golden(f"g7_sampling_{case}_{tag}")
    m, plan, sd = build(case, "fp32")
    m.predict_xstart = px
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000,
                                    timestep_respacing=resp)
    n = diff.num_timesteps
    assert n == int(g["num_steps"])
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    noise = torch.stack([step_noise(1234, i, B, plan.D) for i in range(n)])
    kw = dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
    got = diff.p_sample_loop(m, (B, plan.D), noise=step_noise(1234, -1, B, plan.D).to(DEV), clip_denoised=clip,
                             model_kwargs=kw, device="cuda", step_noise=noise)
    ref = g["sample_at_0"]
    assert np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-3
    if tag == "r50":   # the per-step (progressive) path must agree with the fused loop's structure
        torch.manual_seed(0)
        outs = list(diff.p_sample_loop_progressive(m, (B, plan.D), noise=step_noise(1234, -1, B, plan.D).to(DEV),
                                                   clip_denoised=clip, model_kwargs=kw, device="cuda"))
        assert len(outs) == n and outs[-1]["sample"].shape == (B, plan.D)
        assert torch.isfinite(outs[-1]["sample"]).all()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sampled_mlp_downstream_loss_within_1pct(golden, dtype):
    """north_star check 2, on the surrogate SURVEY.md 8c allows (no pretrained weights / MNIST offline): MNIST-MLP parameter
    vectors sampled by the HIP path (same weights, prompts and injected noise as the reference's 50-step trajectory of
    fixture g7) must score a downstream test loss within 1 % of the loss of the REFERENCE's samples, in the fp32 and in the
    bf16 denoiser.  Scored by the batched MLP evaluator (pinned by fixture g10) on g10's 64 synthetic images, after the
    reference's un-normalisation (/ 4.185, Gpt/data/normalization.py:27-29)."""
    g = golden("g7_sampling_mnist_small_r50")
    ev = golden("g10_mlp_eval")
    m, plan, sd = build("mnist_small", dtype)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="50")
    n, B = diff.num_timesteps, 3
    _, _, lt, lp, xp = make_inputs("mnist_small", B=B, seed=31)
    noise = torch.stack([step_noise(1234, i, B, plan.D) for i in range(n)])
    kw = dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
    got = diff.p_sample_loop(m, (B, plan.D), noise=step_noise(1234, -1, B, plan.D).to(DEV), clip_denoised=False,
                             model_kwargs=kw, device="cuda", step_noise=noise)
    imgs, labels = torch.from_numpy(ev["imgs"]).to(DEV), torch.from_numpy(ev["labels"]).to(DEV)
    ref = torch.from_numpy(g["sample_at_0"]).to(DEV)
    loss_ref, _ = gpt_amd.mlp_test_losses(ref / 4.185, imgs, labels)
    loss_got, _ = gpt_amd.mlp_test_losses(got / 4.185, imgs, labels)
    rel = ((loss_got - loss_ref).abs() / loss_ref).max().item()
    assert rel < 1e-2, (rel, loss_got.tolist(), loss_ref.tolist())


@pytest.mark.parametrize("case,px,clip", [("mnist_small", True, False), ("cifar_small", False, True), ("cartpole_small", True, False)])
def test_seeded_loop_equals_injected_philox_noise(case, px, clip):
    """The DDPM loop with in-kernel noise (Philox drawn inside the posterior kernel, timestep as a scalar) is bit-identical to the
    same loop fed the buffers gpt_randn writes for stream (seed, i): D % 4 == 0 (16-byte groups = whole Philox blocks), D % 4 == 2
    (cifar: 8-byte groups, half blocks) and an odd row width (cartpole, D = 1250: rows start inside a block)."""
    m, plan, sd = build(case, "fp32")
    m.predict_xstart = px
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000, timestep_respacing="20")
    n, B, seed = diff.num_timesteps, 5, 987654321
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    noise = torch.empty(n, B, plan.D, device=DEV)
    for i in range(n):
        L.check(L.lib().gpt_randn(L.ptr(noise[i]), B * plan.D, seed, i, L.stream()), "gpt_randn")
    kw = dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
    start = step_noise(77, -1, B, plan.D).to(DEV)
    a = diff.p_sample_loop(m, (B, plan.D), noise=start.clone(), clip_denoised=clip, model_kwargs=kw, device="cuda", seed=seed)
    b = diff.p_sample_loop(m, (B, plan.D), noise=start.clone(), clip_denoised=clip, model_kwargs=kw, device="cuda", step_noise=noise)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("case,resp,tag,eta,px,clip", [
    ("mnist_small", "ddim25", "ddim25_eta0", 0.0, True, False), ("mnist_small", "50", "r50_eta05", 0.5, True, False),
    ("cifar_small", "ddim25", "ddim25_eta1_eps_clip", 1.0, False, True)])
def test_ddim_sample_loop_golden(golden, case, resp, tag, eta, px, clip):
    """ddim_sample_loop: whole loop in one host call against the reference's end point, and the
    per-step path (fused DDIM kernel per step) against the fused loop."""
    g = golden(f"g11_ddim_{case}_{tag}")
    m, plan, sd = build(case, "fp32")
    m.predict_xstart = px
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000,
                                    timestep_respacing=resp)
    n = diff.num_timesteps
    assert n == int(g["num_steps"])
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    noise = torch.stack([step_noise(1234, i, B, plan.D) for i in range(n)])
    kw = dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
    start = step_noise(1234, -1, B, plan.D).to(DEV)
    got = diff.ddim_sample_loop(m, (B, plan.D), noise=start, clip_denoised=clip, model_kwargs=kw, device="cuda", eta=eta,
                                step_noise=noise)
    ref = g["sample_at_0"]
    assert np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-3
    x = start.clone()
    t = torch.empty(B, dtype=torch.long, device=DEV)
    with torch.no_grad():
        for i in reversed(range(n)):
            t.fill_(i)
            out = diff.ddim_sample(m, x, t, clip_denoised=clip, model_kwargs=kw, eta=eta, noise=noise[i].to(DEV))
            if i == n - 1:
                r0 = g[f"xstart_at_{i}"]
                assert np.abs(out["pred_xstart"].cpu().numpy() - r0).max() / np.abs(r0).max() < 1e-3
            x = out["sample"]
    assert torch.equal(x, got)                              # same kernels, same order: bit-identical


def test_ddim_step_bit_exact():
    """The fused DDIM update against the oracle's fp32 op-by-op restatement on the same inputs: bit-exact."""
    from oracle import diffusion as odf
    B, D = 5, 7960
    g = torch.Generator().manual_seed(77)
    mo, xt, nz = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    t = torch.tensor([0, 1, 17, 500, 999])
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    tb = odf.Tables(1000)
    mod, xtd, nzd, td = mo.to(DEV), xt.to(DEV), nz.to(DEV), t.to(DEV)
    for eta, px, clamp in ((0.0, True, None), (0.5, True, (-0.7, 0.7)), (1.0, False, None)):
        out, x0 = torch.empty(B, D, device=DEV), torch.empty(B, D, device=DEV)
        lo, hi = clamp if clamp else (0.0, 0.0)
        L.check(L.lib().gpt_ddim_step(L.ptr(mod), L.ptr(xtd), L.ptr(nzd), L.ptr(td),
                                      L.ptr(diff.coef_table(DEV)), L.ptr(diff._ac_table(DEV)), int(px), int(clamp is not None),
                                      lo, hi, eta, L.ptr(out), L.ptr(x0), B, D, L.stream()), "ddim")
        ref, ref0 = odf.ddim_step(tb, mo, xt, t, nz, eta, px, False, *(clamp if clamp else (None, None)))
        assert torch.equal(x0.cpu(), ref0) and torch.equal(out.cpu(), ref), (eta, px, clamp)


def test_p_sample_loop_philox_statistics():
    """Throughput mode (in-kernel Philox noise): repeatable for a seed, and N(0,1) to sampling error."""
    n = 1 << 20
    a, b = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    L.check(L.lib().gpt_randn(L.ptr(a), n, 7, 3, L.stream()), "randn")
    L.check(L.lib().gpt_randn(L.ptr(b), n, 7, 3, L.stream()), "randn")
    assert torch.equal(a, b)
    L.check(L.lib().gpt_randn(L.ptr(b), n, 7, 4, L.stream()), "randn")
    assert not torch.equal(a, b)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3) < 0.1 and abs(float((a * b).mean())) < 5e-3


def test_sampling_properties_full_size():
    """Full-size MNIST model, bf16: rows are independent (sharding by rows changes nothing) and the fused
    loop equals the per-step path fed the same noise."""
    m, plan, sd = build("mnist", "bf16", seed=1)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="4")
    B, n = 6, diff.num_timesteps
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("mnist", B=B, seed=9))
    noise = torch.stack([step_noise(99, i, B, plan.D) for i in range(n)]).to(DEV)
    start = step_noise(99, -1, B, plan.D).to(DEV)
    kw = dict(loss_target=lt, loss_prev=lp, x_prev=xp)
    full = diff.p_sample_loop(m, (B, plan.D), noise=start, clip_denoised=False, model_kwargs=kw, step_noise=noise)
    half = diff.p_sample_loop(m, (3, plan.D), noise=start[:3], clip_denoised=False, step_noise=noise[:, :3].contiguous(),
                              model_kwargs=dict(loss_target=lt[:3], loss_prev=lp[:3], x_prev=xp[:3]))
    assert torch.equal(full[:3], half)
    x = start.clone()
    tmap = torch.tensor(diff.timestep_map, device=DEV)
    for i in reversed(range(n)):
        t = torch.full((B,), i, device=DEV, dtype=torch.long)
        with torch.no_grad():
            out = m(x, tmap[t], lt, loss_prev=lp, x_prev=xp)
        x, _ = diff._posterior(out, x, noise[i].contiguous(), t, None)
    assert torch.equal(full, x)


@pytest.mark.parametrize("dtype,bar", [("fp32", 1e-3), ("bf16", 1e-2)])
def test_full_size_sampling_vs_oracle_and_downstream_loss(golden, dtype, bar):
    """The pretrained mnist_loss shapes (d = 1536, 12 layers, D = 7960): a 10-step respaced p_sample_loop with injected noise
    on the HIP path against the oracle's loop on the host.  fp32: sampled vectors within 1e-3 (north_star check 1 carried
    through the whole loop); both dtypes: downstream MLP test loss of the samples within 1 % of the oracle's (check 2)."""
    m, plan, sd = build("mnist", dtype, seed=2)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="10")
    tb = odf.Tables(1000, "10")
    B, n = 4, diff.num_timesteps
    _, _, lt, lp, xp = make_inputs("mnist", B=B, seed=12)
    start = step_noise(77, -1, B, plan.D)
    noise = torch.stack([step_noise(77, i, B, plan.D) for i in range(n)])
    with torch.no_grad():
        ref = odf.p_sample_loop(tb, lambda x, t: od.forward(sd, plan, x, t, lt, lp, xp), start, lambda i: noise[i])
    kw = dict(loss_target=lt.to(DEV), loss_prev=lp.to(DEV), x_prev=xp.to(DEV))
    got = diff.p_sample_loop(m, (B, plan.D), noise=start.to(DEV), clip_denoised=False, model_kwargs=kw, device="cuda",
                             step_noise=noise)
    if dtype == "fp32":
        assert rel_err(got.cpu(), ref.double()) < bar
    ev = golden("g10_mlp_eval")
    imgs, labels = torch.from_numpy(ev["imgs"]).to(DEV), torch.from_numpy(ev["labels"]).to(DEV)
    loss_ref, _ = gpt_amd.mlp_test_losses(ref.to(DEV) / 4.185, imgs, labels)
    loss_got, _ = gpt_amd.mlp_test_losses(got / 4.185, imgs, labels)
    rel = ((loss_got - loss_ref).abs() / loss_ref).max().item()
    assert rel < 1e-2, (rel, loss_got.tolist(), loss_ref.tolist())


# ------------------------------------------------------------------------------------------ permutation
def test_permutation_maps_bit_exact(golden):
    g = golden("g8_permute")
    P = gpt_amd.permute
    for arch, key, draw, idx in ((P.MLP_ARCH, "mlp", op.draw_mlp, op.mlp_index), (P.CNN_ARCH, "cnn", op.draw_cnn, op.cnn_index)):
        perms = P.draw_permutations(arch, range(16))
        index = P.build_index(arch, perms, DEV).cpu().numpy()
        assert index.dtype == np.int64
        assert np.array_equal(index, g[key].astype(np.int64))
        assert np.array_equal(index[5], idx(*draw(5)))
        D = index.shape[1]
        gen = torch.Generator().manual_seed(1)
        a, b = torch.randn(16, D, generator=gen), torch.randn(16, D, generator=gen)
        pa, pb = P.random_permute_batch([a.to(DEV), b.to(DEV)], arch, range(16))
        ia = torch.from_numpy(index)
        assert torch.equal(pa.cpu(), torch.gather(a, 1, ia)) and torch.equal(pb.cpu(), torch.gather(b, 1, ia))


def test_actor_permutation_maps_bit_exact(golden):
    """The third shipped task (Cartpole policy, data_gen/train_rl.py:51-74): int64 maps of gpt_perm_index_chain against the
    reference's (fixture g15, 16 seeds) and the oracle's, and the gather of a pair through them."""
    g = golden("g15_actor_permute")
    P = gpt_amd.permute
    arch = P.ACTOR_ARCH
    assert P.arch_dim(arch) == 1250
    perms = P.draw_permutations(arch, range(16))
    index = P.build_index(arch, perms, DEV).cpu().numpy()
    assert index.dtype == np.int64 and np.array_equal(index, g["actor"].astype(np.int64))
    assert np.array_equal(index[7], op.actor_index(op.draw_actor(7)))
    gen = torch.Generator().manual_seed(2)
    a, b = torch.randn(16, 1250, generator=gen), torch.randn(16, 1250, generator=gen)
    pa, pb = P.random_permute_batch([a.to(DEV), b.to(DEV)], arch, range(16))
    ia = torch.from_numpy(index)
    assert torch.equal(pa.cpu(), torch.gather(a, 1, ia)) and torch.equal(pb.cpu(), torch.gather(b, 1, ia))
    # a deeper chain with other widths (three hidden layers, two leading scalars) against the oracle's construction
    arch2 = dict(kind="chain", prefix=2, dims=[5, 7, 16, 3, 2])
    perms2 = P.draw_permutations(arch2, [11, 12])
    idx2 = P.build_index(arch2, perms2, DEV).cpu().numpy()
    for row, seed in zip(idx2, (11, 12)):
        want = op.actor_index(op.draw_actor(seed, (7, 16, 3)), obs=5, act=2)
        assert np.array_equal(row, want) and sorted(row.tolist()) == list(range(P.arch_dim(arch2)))
    # rejected before launch: a chain with one Linear, too many widths
    import ctypes as C
    lib = L.lib()
    assert lib.gpt_perm_index_chain(L.ptr(perms.to(DEV)), L.ptr(torch.empty(4, dtype=torch.long, device=DEV)), 1, 0, 2, (C.c_int * 2)(4, 1), L.stream()) != 0
    assert lib.gpt_perm_index_chain(L.ptr(perms.to(DEV)), L.ptr(torch.empty(4, dtype=torch.long, device=DEV)), 1, 0, 10, (C.c_int * 10)(*[2] * 10), L.stream()) != 0


@pytest.mark.parametrize("dtype,bar", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_cartpole_full_size_forward_vs_oracle(dtype, bar):
    """configs/train/cartpole.yaml:38-46 at its real shapes: layer_by_layer tokens (18 of them, widths 1 ... 1024), n_embd 1536,
    12 heads of 128, 12 layers; prompts are returns (hundreds), B = 8 (the reference's vis batch, cartpole.yaml:22-24)."""
    case = "cartpole"
    m, plan, sd = build(case, dtype, seed=15)
    assert plan.T == 18
    cpu = make_inputs(case, B=8, seed=41)
    x, t, lt, lp, xp = (v.to(DEV) for v in cpu)
    with torch.no_grad():
        out = m(x, t, lt, loss_prev=lp, x_prev=xp).cpu()
        ref = od.forward(sd, plan, *cpu)
    scale = float((ref - cpu[4]).abs().max())
    assert float((out - ref).abs().max()) / scale < bar


def test_batched_mlp_eval_golden_and_oracle(golden):
    """All sampled nets scored in one pass (fc1 of every net as one fp32 MFMA GEMM + fused head kernel)."""
    from oracle import evaluate as oe
    g = golden("g10_mlp_eval")
    nets, imgs, labels = (torch.from_numpy(g[k]) for k in ("nets", "imgs", "labels"))
    loss, err = gpt_amd.mlp_test_losses(nets.to(DEV), imgs.to(DEV), labels.to(DEV))
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=2e-5)
    np.testing.assert_allclose(err.cpu().numpy(), g["err"], rtol=0, atol=1e-4)
    # ragged sizes (image count not a multiple of 8) and a larger batch of nets against the oracle
    gen = torch.Generator().manual_seed(77)
    nets = torch.randn(37, 7960, generator=gen) * 0.1
    imgs = torch.randn(203, 1, 28, 28, generator=gen)
    labels = torch.randint(0, 10, (203,), generator=gen)
    want_loss, want_err = oe.mlp_eval(nets, imgs, labels)
    loss, err = gpt_amd.mlp_test_losses(nets.to(DEV), imgs.to(DEV), labels.to(DEV))
    np.testing.assert_allclose(loss.cpu().numpy(), want_loss.numpy(), rtol=2e-5)
    np.testing.assert_allclose(err.cpu().numpy(), want_err.numpy(), rtol=0, atol=1e-4)
    with pytest.raises(L.GptHipError):
        gpt_amd.mlp_test_losses(nets, imgs, labels)                 # CPU tensors: no fallback


@pytest.mark.parametrize("tag,respacing,px", [("full_x0", "", True), ("r50_eps", "50", False)])
def test_diffusion_misc_golden(golden, tag, respacing, px):
    """ddim_reverse_sample (gpt_ddim_reverse_step), q_posterior_mean_variance (fused posterior kernel), q_mean_variance
    (gpt_q_sample) BIT-EXACT against the reference's fixture; _prior_bpd (gpt_prior_bpd) to 1e-6 (row-mean order)."""
    g = golden("g14_diffusion_misc")
    d = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000,
                                 timestep_respacing=respacing)
    x, x0, mo, t = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "x0", "mo", f"{tag}_t"))
    r = d.ddim_reverse_sample(lambda xx, tt, **kw: mo, x, t, clip_denoised=(tag == "r50_eps"))
    np.testing.assert_array_equal(r["sample"].cpu().numpy(), g[f"{tag}_rev"])
    np.testing.assert_array_equal(r["pred_xstart"].cpu().numpy(), g[f"{tag}_rev_x0"])
    m, v, lv = d.q_mean_variance(x0, t)
    assert v.shape == x0.shape and lv.shape == x0.shape
    np.testing.assert_array_equal(m.cpu().numpy(), g[f"{tag}_qm"])
    np.testing.assert_array_equal(v[:, 0].cpu().numpy(), g[f"{tag}_qv"])
    np.testing.assert_array_equal(lv[:, 0].cpu().numpy(), g[f"{tag}_qlv"])
    m, v, lv = d.q_posterior_mean_variance(x0, x, t)
    np.testing.assert_array_equal(m.cpu().numpy(), g[f"{tag}_pm"])
    np.testing.assert_array_equal(v[:, 0].cpu().numpy(), g[f"{tag}_pv"])
    np.testing.assert_array_equal(lv[:, 0].cpu().numpy(), g[f"{tag}_plv"])
    np.testing.assert_allclose(d._prior_bpd(x0).cpu().numpy(), g[f"{tag}_bpd"], rtol=1e-6)
    with pytest.raises(AssertionError):
        d.ddim_reverse_sample(lambda xx, tt, **kw: mo, x, t, eta=0.5)


def test_batched_cnn_eval_golden_and_oracle(golden):
    """All sampled CIFAR-10 ConvNets scored by one fused kernel (conv1 VALU, conv2 fp32 MFMA, pool, fc, loss): fp32
    throughout, summation order differs from ATen's convolutions -> rtol 2e-5 on the mean loss; the error count is exact
    unless two logits tie to within rounding (none do on these seeds)."""
    from oracle import evaluate as oe
    g = golden("g12_cnn_eval")
    nets, imgs, labels = (torch.from_numpy(g[k]) for k in ("nets", "imgs", "labels"))
    loss, err = gpt_amd.cnn_test_losses(nets.to(DEV), imgs.to(DEV), labels.to(DEV))
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss"], rtol=2e-5)
    np.testing.assert_allclose(err.cpu().numpy(), g["err"], rtol=0, atol=1e-4)
    # more nets than one workgroup round, a ragged last group (B % 4 = 1), an image count that the chunks do not divide,
    # weights at trained-net scale so that the ReLUs and the logits are far from degenerate
    gen = torch.Generator().manual_seed(78)
    nets = torch.randn(1029, 5370, generator=gen) * 0.3
    imgs = torch.randn(203, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 10, (203,), generator=gen)
    loss, err = gpt_amd.cnn_test_losses(nets.to(DEV), imgs.to(DEV), labels.to(DEV))
    sel = torch.tensor([0, 1, 2, 3, 4, 515, 1024, 1027, 1028])
    want_loss, want_err = oe.cnn_eval(nets[sel], imgs, labels)
    np.testing.assert_allclose(loss.cpu()[sel].numpy(), want_loss.numpy(), rtol=2e-5)
    np.testing.assert_allclose(err.cpu()[sel].numpy(), want_err.numpy(), rtol=0, atol=1e-4)
    # one image, one net
    l1, e1 = gpt_amd.cnn_test_losses(nets[:1].to(DEV), imgs[:1].to(DEV), labels[:1].to(DEV))
    w1, we1 = oe.cnn_eval(nets[:1], imgs[:1], labels[:1])
    np.testing.assert_allclose(l1.cpu().numpy(), w1.numpy(), rtol=2e-5)
    assert float(e1) == float(we1)
    with pytest.raises(L.GptHipError):
        gpt_amd.cnn_test_losses(nets, imgs, labels)                 # CPU tensors: no fallback


@pytest.mark.parametrize("case", ["mnist", "cifar"])
def test_checkpoint_store_pairs_golden_and_oracle(golden, tmp_path, case):
    """CheckpointStore (split resident in HBM, one gather launch per batch) against the reference's ParameterDataset
    fixture: BIT-EXACT parameters (row gather, fp32 multiply by the coefficient, integer permutation map), identical keys,
    spacings and metrics for the same RNG state; then a larger batch and the un-augmented path against the oracle."""
    from gpt_amd import data as gdata
    from oracle import data as odata
    from tests.cases import PAIR_CASES, make_pair_dataset
    g, c = golden("g13_pairs"), PAIR_CASES[case]
    dbs = make_pair_dataset(str(tmp_path), case)
    for path, db in dbs.items():
        gdata.write_flat_db(path, db.items())          # beside runs.json, as the converter leaves it
    m = c["train_metric"]
    for split in ("train", "test"):
        ds = gdata.CheckpointStore(str(tmp_path), c["dataset_name"], split=split, num_test_runs=c["num_test_runs"],
                                   train_metric=m, openai_coeff=c["openai_coeff"], target_epoch_size=1000,
                                   permute_augment=True, device=DEV)
        tag = f"{case}_{split}"
        assert list(g[f"{tag}_runs"]) == ds.runs and list(g[f"{tag}_sizes"]) == ds.parameter_sizes
        np.testing.assert_array_equal(g[f"{tag}_scalars"], np.array([ds.min_val, ds.max_val, ds.optimal_test_loss,
                                                                     float(len(ds)), *ds.get_range(normalize=True)]))
        torch.manual_seed(c["seed"] + 1)
        b = ds.get_batch(c["items"])
        assert b["checkpoint_key_0"] == list(g[f"{tag}_key0"]) and b["checkpoint_key_1"] == list(g[f"{tag}_key1"])
        assert b["run_name"] == list(g[f"{tag}_run"])
        np.testing.assert_array_equal(b["step_spacing"].numpy(), g[f"{tag}_spacing"])
        assert b[f"{m}_0"].dtype == torch.float64
        np.testing.assert_array_equal(b[f"{m}_0"].numpy(), g[f"{tag}_m0"])
        np.testing.assert_array_equal(b[f"{m}_1"].numpy(), g[f"{tag}_m1"])
        np.testing.assert_array_equal(b["parameters_0"].cpu().numpy()[:, ::7], g[f"{tag}_p0"])
        np.testing.assert_array_equal(b["parameters_1"].cpu().numpy()[:, ::7], g[f"{tag}_p1"])
        net = ds.get_run_network(len(ds.runs) - 1, iter=2, normalize=True, augment=False)
        np.testing.assert_array_equal(net.cpu().numpy()[::7], g[f"{tag}_net"])
        # the same items one at a time consume the RNG identically
        torch.manual_seed(c["seed"] + 1)
        one = [ds[i] for i in c["items"][:3]]
        for k, it in enumerate(one):
            assert torch.equal(it["parameters_1"], b["parameters_1"][k]) and it["checkpoint_key_0"] == b["checkpoint_key_0"][k]
    # a batch of 300 indices, no augmentation, explicit generator, identity normaliser: against the oracle
    ds = gdata.CheckpointStore(str(tmp_path), c["dataset_name"], split="train", num_test_runs=c["num_test_runs"], train_metric=m,
                               normalizer_name="none", target_epoch_size=1000, device=DEV, min_step_spacing=2,
                               max_step_spacing=3, generator=torch.Generator().manual_seed(5))
    ref = odata.RefPairDataset(str(tmp_path), lambda p: dbs[p], split="train", num_test_runs=c["num_test_runs"],
                               train_metric=m, normalizer_name="none", target_epoch_size=1000, min_step_spacing=2,
                               max_step_spacing=3)
    ref.generator = torch.Generator().manual_seed(5)
    idx = torch.randint(0, len(ds), (300,), generator=torch.Generator().manual_seed(6)).tolist()
    b = ds.get_batch(idx)
    want = [ref[i] for i in idx]
    assert torch.equal(b["parameters_0"].cpu(), torch.stack([w["parameters_0"] for w in want]))
    assert torch.equal(b["parameters_1"].cpu(), torch.stack([w["parameters_1"] for w in want]))
    assert b["checkpoint_key_1"] == [w["checkpoint_key_1"] for w in want]
    assert ds.get_range(normalize=True) == ref.get_range()
    # one epoch of batches for two ranks: disjoint shares of one permutation, whole batches only
    seen = []
    for rank in (0, 1):
        n = 0
        for bb in ds.batches(64, seed=3, rank=rank, world=2):
            assert bb["parameters_0"].shape == (64, ds.D)
            n += 1
        seen.append(n)
    assert seen[0] == seen[1] == (len(ds) // 2) // 64


def test_synth_matches_p_sample_loop_with_static_thresholding(golden):
    """synth (Gpt/vis.py:210-252) = p_sample_loop with the harness' model_kwargs and the 'static' denoised_fn."""
    m, plan, sd = build("mnist_small", "fp32")
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                                    timestep_respacing="10")
    _, _, lt, lp, xp = (v.to(DEV) for v in make_inputs("mnist_small", B=4, seed=9))
    noise = torch.randn(4, m.D, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        a = gpt_amd.synth(diff, m, lt, lp, xp, param_range=(-0.5, 0.5), thresholding="static", noise=noise, seed=11)
        b = diff.p_sample_loop(m, xp.shape, noise=noise, clip_denoised=False, denoised_fn=gpt_amd.StaticThreshold(-0.5, 0.5),
                               model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), device=DEV, seed=11)
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    with pytest.raises(NotImplementedError):
        gpt_amd.create_thresholding_fn("dynamic", (-1, 1))


@pytest.mark.parametrize("dtype,K,tol", [("fp32", 48, 1e-5), ("bf16", 96, 2e-2)])
def test_gemm_many_tiles_k_not_multiple_of_the_big_k_tile(dtype, K, tol):
    """>= 128 output tiles but K is not a multiple of the 256 x 256 kernel's 128-byte K-tile: must take the ring
    kernels (a silent K truncation would drop the last 64 bytes of every row)."""
    code, td = L.dtype_code(dtype), L.torch_dtype(L.dtype_code(dtype))
    g = torch.Generator().manual_seed(K)
    M, N = 2560, 3328
    A = torch.randn(M, K, generator=g).to(td).to(DEV)
    W = torch.randn(N, K, generator=g).to(td).to(DEV)
    Cm = torch.empty(M, N, device=DEV, dtype=torch.float32)
    L.check(L.lib().gpt_gemm_nt(code, L.ptr(A), K, L.ptr(W), K, None, L.ptr(Cm), N, None, 0, M, N, K, L.EPI_STORE_F32, L.stream()),
            "gpt_gemm_nt")
    ref = A.double() @ W.double().t()
    assert float((Cm.double() - ref).abs().max()) < tol * float(ref.abs().max())
