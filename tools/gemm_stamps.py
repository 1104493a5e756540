#!/usr/bin/env python3
"""Where a 256 x 256 GEMM workgroup spends its time: reads the s_memtime stamps of the DIAGNOSTIC library build
(make -C g.pt_amd/csrc OUT=../lib/libgpt_hip_diag.so OBJDIR=../lib/obj_diag EXTRA=-DGPT_STAMPS) after one launch per shape.
Usage: GPT_HIP_LIB=g.pt_amd/lib/libgpt_hip_diag.so GPT_GEMM_TILE=3 python tools/gemm_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_amd import _lib as L  # noqa: E402


def main():
    lib = L.lib()
    lib.gpt_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    dev = "cuda:0"
    M = 320 * 26
    d = 1536
    shapes = [("qkv", M, 3 * d, d, L.EPI_STORE), ("proj", M, d, d, L.EPI_RESADD), ("fc1", M, 4 * d, d, L.EPI_GELU),
              ("fc2", M, d, 4 * d, L.EPI_RESADD), ("sq8192", 8192, 8192, 8192, L.EPI_STORE)]
    for name, m, n, k, epi in shapes:
        A = torch.randn(m, k, device=dev).to(torch.bfloat16)
        W = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        bias = torch.randn(n, device=dev)
        Cb = torch.zeros(m, n, device=dev, dtype=torch.float32 if epi == L.EPI_RESADD else torch.bfloat16)
        for _ in range(30):     # warm clocks
            L.check(lib.gpt_gemm_nt(L.BF16, L.ptr(A), k, L.ptr(W), k, L.ptr(bias), L.ptr(Cb), n, None, 0, m, n, k, epi, L.stream()), "gemm")
        torch.cuda.synchronize()
        ntile = ((m + 255) // 256) * ((n + 255) // 256)
        nb = min(ntile, 2048)
        buf = (C.c_ulonglong * (nb * 8))()
        assert lib.gpt_debug_stamps(buf, nb * 8) == 0
        s = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 8).astype(np.int64)
        pro, main_, epi_, tot = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2], s[:, 3] - s[:, 0]
        rt = (s[:, 5] - s[:, 4]).astype(np.float64) / 100.0          # us (100 MHz)
        clk = np.median(tot / np.maximum(rt, 1e-3)) / 1e3            # GHz
        start = (s[:, 4] - s[:, 4].min()) / 100.0
        end = (s[:, 5] - s[:, 4].min()) / 100.0
        nk = k // 64
        med = lambda v: float(np.median(v))
        print(f"{name:7s} tiles {ntile:5d} nk {nk:3d} clk {clk:.2f} GHz | cycles: prologue {med(pro):7.0f} main {med(main_):8.0f} "
              f"({med(main_) / nk:6.0f}/ktile) epilogue {med(epi_):7.0f} (p90 {np.percentile(epi_, 90):7.0f}) | us: total {med(rt):6.1f} "
              f"start p50/max {med(start):6.1f}/{start.max():6.1f} end p50/max {med(end):6.1f}/{end.max():6.1f}", flush=True)


if __name__ == "__main__":
    main()
