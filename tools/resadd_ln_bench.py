#!/usr/bin/env python3
"""gpt_gemm_resadd_layernorm (residual Linear + the LayerNorm behind it) at the small-batch shapes; run under GPT_GEMM_FUSELN=0/1 to
compare the fused span-sum + LayerNorm route with GEMM (+ span sum) + LayerNorm.  usage: resadd_ln_bench.py M [M ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt_amd import _lib as L

dev = "cuda:0"
lib = L.lib()
L.ensure_gemm_workspace(dev)
d = 1536
for M in [int(a) for a in sys.argv[1:]] or [1664, 3328, 3520]:
    for K in (1536, 6144):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(d, K, device=dev) * 0.05).bfloat16()
        bias, x = torch.randn(d, device=dev), torch.randn(M, d, device=dev)
        gam, bet = torch.ones(d, device=dev), torch.zeros(d, device=dev)
        y = torch.empty(M, d, device=dev, dtype=torch.bfloat16)

        def fn():
            L.check(lib.gpt_gemm_resadd_layernorm(L.BF16, L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(x), d, L.ptr(x), d, L.ptr(gam), L.ptr(bet),
                                                  L.ptr(y), None, None, M, d, K, 1e-5, L.stream()), "rl")
        for _ in range(5):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record()
        torch.cuda.synchronize()
        print(f"M={M} K={K}: {a.elapsed_time(b) * 20:.1f} us", flush=True)
