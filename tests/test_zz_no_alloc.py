"""gpt_gemm_nt_static - the entry point that allocates nothing (its K-span scratch is a static device array inside the
library) - against gpt_gemm_nt with a registered workspace of the same size: bit-identical, on two calls in a row, the
second with a smaller input.  Collected last (file name) so that a failure here cannot hide another test under `pytest -x`."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpt_amd import _lib as L               # noqa: E402

DEV = "cuda:0"
SCRATCH = 64 << 20                          # GPT_GEMM_STATIC_SCRATCH_BYTES (include/gpt_hip.h)


def _problem(M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    W = (torch.randn(N, K, generator=g) * 0.05).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV)
    return A, W, bias, R


def _run(fn, A, W, bias, R, epi):
    M, K = A.shape
    N = W.shape[0]
    if epi == L.EPI_RESADD:
        out = R.clone()
    else:
        out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
    L.check(fn(L.BF16, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(out), N, None, 0, M, N, K, epi, L.stream()), "gemm")
    torch.cuda.synchronize()
    return out


def test_static_scratch_entry_point_matches_workspace_entry_point():
    try:
        _static_against_workspace()
    finally:
        # leave the process as the other tests expect it: the engines' K-span workspace registered again (the body below
        # unregisters it; an engine built earlier would silently take the single-pass kernels otherwise - ADVICE r02)
        L._gemm_ws.pop(torch.device(DEV).index, None)
        L.ensure_gemm_workspace(torch.device(DEV))


def _static_against_workspace():
    lib = L.lib()
    ws = torch.empty(SCRATCH, dtype=torch.uint8, device=DEV)
    # two calls in a row, the second smaller (it finds the first call's partial tiles in the scratch and must not read them);
    # both shapes take the K-span path: few 256 x 256 tiles, K = 6144 / 4608
    shapes = [(1664, 1536, 6144), (832, 1536, 4608)]
    problems = [_problem(M, N, K, seed=M + K) for M, N, K in shapes]
    try:
        L.check(lib.gpt_gemm_set_workspace(L.ptr(ws), ws.numel()), "ws")
        want = [[_run(lib.gpt_gemm_nt, *p, epi) for epi in (L.EPI_STORE, L.EPI_GELU, L.EPI_RESADD)] for p in problems]
    finally:
        L.check(lib.gpt_gemm_set_workspace(None, 0), "ws")        # nothing registered from here on: the static scratch or nothing
        L._gemm_ws.pop(torch.device(DEV).index, None)
    single = _run(lib.gpt_gemm_nt, *problems[0], L.EPI_STORE)       # without a workspace: the single-pass kernels
    got = [[_run(lib.gpt_gemm_nt_static, *p, epi) for epi in (L.EPI_STORE, L.EPI_GELU, L.EPI_RESADD)] for p in problems]
    for w_row, g_row in zip(want, got):
        for w, g in zip(w_row, g_row):
            assert torch.equal(w, g)
    # the K-span path really ran (its fp32 partial sums round differently from the single-pass kernel somewhere)
    assert not torch.equal(single, got[0][0])
    assert float((single.float() - got[0][0].float()).abs().max()) <= 2.0 ** -6 * float(single.float().abs().max())
    # and again, first shape after the smaller one
    again = _run(lib.gpt_gemm_nt_static, *problems[0], L.EPI_STORE)
    assert torch.equal(again, want[0][0])
    # beyond the maximum: rejected before any launch, with a message
    M, N, K = 8320, 1536, 64
    A, W, bias, R = _problem(M, N, K, 5)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    rc = lib.gpt_gemm_nt_static(L.BF16, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(out), N, None, 0, M, N, K, L.EPI_STORE, L.stream())
    assert rc < 0 and b"exceeds the static scratch" in lib.gpt_last_error()
    assert float(out.float().abs().max()) == 0.0
