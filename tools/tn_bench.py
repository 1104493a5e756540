#!/usr/bin/env python3
"""TN (wgrad) kernel vs the NT kernel on the same contraction.  usage: tn_bench.py [R]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt_amd import _lib as L
lib, dev = L.lib(), "cuda"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 26624


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for I, J in ((6144, 1536), (1536, 6144), (4608, 1536), (1536, 1536), (8192, 8192)):
    P = torch.randn(R, I, device=dev).bfloat16()
    Q = torch.randn(R, J, device=dev).bfloat16()
    Pt, Qt = P.t().contiguous(), Q.t().contiguous()
    C = torch.zeros(I, J, device=dev)
    fl = 2.0 * I * J * R
    t_tn = timeit(lambda: L.check(lib.gpt_gemm_tn(L.ptr(P), I, L.ptr(Q), J, L.ptr(C), J, I, J, R, 1, L.stream()), "tn"))
    t_nt = timeit(lambda: L.check(lib.gpt_gemm_nt(L.BF16, L.ptr(Pt), R, L.ptr(Qt), R, None, L.ptr(C), J, None, 0, I, J, R,
                                                  L.EPI_ACCUM_F32, L.stream()), "nt"))
    print(f"I={I:5d} J={J:5d} R={R}:  TN {t_tn:8.1f} us {fl / t_tn / 1e6:7.1f} TF   NT {t_nt:8.1f} us {fl / t_nt / 1e6:7.1f} TF", flush=True)
