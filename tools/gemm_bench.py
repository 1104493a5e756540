#!/usr/bin/env python3
"""Micro-benchmark of gpt_gemm_nt on the denoiser's GEMM shapes (random operands, interleaved rounds in one
process, HIP-event timing per launch).  Usage: python tools/gemm_bench.py [--batch 320] [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=320)
    ap.add_argument("--d", type=int, default=1536)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=20)
    ap.add_argument("--shapes", nargs="*", default=None)
    args = ap.parse_args()
    code = L.dtype_code(args.dtype)
    td = L.torch_dtype(code)
    dev = "cuda:0"
    M, d = args.batch * 26, args.d
    shapes = [("qkv", M, 3 * d, d, L.EPI_STORE), ("proj", M, d, d, L.EPI_RESADD), ("fc1", M, 4 * d, d, L.EPI_GELU),
              ("fc2", M, d, 4 * d, L.EPI_RESADD), ("sq4096", 4096, 4096, 4096, L.EPI_STORE),
              ("sq8192", 8192, 8192, 8192, L.EPI_STORE)]
    if args.shapes:                      # "name:M:N:K[:epi]" ...
        shapes = []
        for spec in args.shapes:
            f = spec.split(":")
            shapes.append((f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]) if len(f) > 4 else L.EPI_STORE))
    lib = L.lib()
    L.ensure_gemm_workspace(dev)          # GPT_GEMM_SPLITK=0 in the environment keeps the single-pass kernels
    bufs = []
    for name, m, n, k, epi in shapes:
        A = torch.randn(m, k, device=dev).to(td)
        W = (torch.randn(n, k, device=dev) * 0.05).to(td)
        bias = torch.randn(n, device=dev)
        out_dtype = torch.float32 if epi in (L.EPI_RESADD, L.EPI_STORE_F32) else td
        Cb = torch.zeros(m, n, device=dev, dtype=out_dtype)
        bufs.append((A, W, bias, Cb))
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in shapes]
          for _ in range(args.rounds)]
    s = L.stream()
    for r in range(-2, args.rounds):
        for i, (name, m, n, k, epi) in enumerate(shapes):
            A, W, bias, Cb = bufs[i]
            if r >= 0:
                ev[r][i][0].record()
            L.check(lib.gpt_gemm_nt(code, L.ptr(A), k, L.ptr(W), k, L.ptr(bias), L.ptr(Cb), n, None, 0, m, n, k, epi, s),
                    "gemm")
            if r >= 0:
                ev[r][i][1].record()
    torch.cuda.synchronize()
    tot_t = tot_f = 0.0
    for i, (name, m, n, k, epi) in enumerate(shapes):
        ts = sorted(ev[r][i][0].elapsed_time(ev[r][i][1]) for r in range(args.rounds))
        med, mn = ts[len(ts) // 2], ts[0]
        fl = 2.0 * m * n * k
        if i < 4 and not args.shapes:
            tot_t += med
            tot_f += fl
        print(f"{name:8s} M={m:6d} N={n:5d} K={k:5d}  median {med * 1e3:8.1f} us  {fl / med / 1e9:7.1f} TF   min {mn * 1e3:8.1f} us {fl / mn / 1e9:7.1f} TF")
    if not args.shapes:
        print(f"layer total (4 GEMMs): {tot_t * 1e3:.1f} us  {tot_f / tot_t / 1e9:.1f} TF")


if __name__ == "__main__":
    main()
