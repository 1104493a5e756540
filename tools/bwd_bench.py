#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound backward kernels at the training shapes (M = 128*26 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gpt_amd import _lib as L

dev = torch.device("cuda")
lib = L.lib()
M, d = int(os.environ.get("M", 3328)), 1536


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


dy = torch.randn(M, d, device=dev).bfloat16()
x = torch.randn(M, d, device=dev)
mean, rstd = x.mean(1).contiguous(), (1 / x.std(1)).contiguous()
gamma = torch.randn(d, device=dev)
resid = torch.randn(M, d, device=dev)
gout = torch.empty(M, d, device=dev)
gt = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
us = timeit(lambda: L.check(lib.gpt_layernorm_bwd(L.ptr(dy), L.BF16, L.ptr(x), L.ptr(mean), L.ptr(rstd), L.ptr(gamma), L.ptr(resid),
                                                   L.ptr(gout), L.ptr(gt), L.BF16, L.ptr(dg), L.ptr(db), M, d, L.stream()), "ln"))
byts = M * d * (2 + 4 + 4 + 4 + 2)
print(f"layernorm_bwd  {us:7.1f} us  {byts / us / 1e6:6.2f} TB/s")
for cols in (d, 3 * d, 4 * d):
    X = torch.randn(M, cols, device=dev).bfloat16()
    out = torch.zeros(cols, device=dev)
    us = timeit(lambda: L.check(lib.gpt_colsum(L.BF16, L.ptr(X), cols, L.ptr(out), M, cols, L.stream()), "cs"))
    print(f"colsum {cols:5d}   {us:7.1f} us  {M * cols * 2 / us / 1e6:6.2f} TB/s")
    Mp = (M + 63) // 64 * 64
    Xt = torch.empty(cols, Mp, device=dev, dtype=torch.bfloat16)
    us = timeit(lambda: L.check(lib.gpt_transpose(L.BF16, L.ptr(X), L.ptr(Xt), M, cols, Mp, L.stream()), "tr"))
    print(f"transpose {cols:5d}{us:7.1f} us  {M * cols * 4 / us / 1e6:6.2f} TB/s")
