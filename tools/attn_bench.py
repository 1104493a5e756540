#!/usr/bin/env python3
"""Micro-benchmark of gpt_attention (bf16) over (B, heads, head dim) cases: HIP events around each launch, median of 30, four rotating
operand sets.  Usage: python tools/attn_bench.py [B:H:hd ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_amd import _lib as L  # noqa: E402

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(64, 16, 96), (320, 16, 96), (68, 24, 128), (68, 16, 128),
                                                                       (64, 16, 128), (1280, 16, 128), (128, 12, 128)]
T, dev, lib = 26, "cuda:0", L.lib()
for B, H, hd in cases:
    d = H * hd
    sets = [(torch.randn(B * T, 3 * d, device=dev).bfloat16(), torch.empty(B * T, d, device=dev, dtype=torch.bfloat16)) for _ in range(4)]
    ts = []
    for i in range(-5, 30):
        q, y = sets[i % 4]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.gpt_attention(L.BF16, L.ptr(q), L.ptr(y), B, T, H, hd, L.stream()), "gpt_attention")
        e1.record()
        torch.cuda.synchronize()
        if i >= 0:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    dy = torch.randn(B * T, d, device=dev).bfloat16()
    dq = torch.empty(B * T, 3 * d, device=dev, dtype=torch.bfloat16)
    tb = []
    for i in range(-5, 30):
        q, y = sets[i % 4]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.check(lib.gpt_attention_bwd(L.BF16, L.ptr(q), L.ptr(dy), L.ptr(dq), B, T, H, hd, L.stream()), "gpt_attention_bwd")
        e1.record()
        torch.cuda.synchronize()
        if i >= 0:
            tb.append(e0.elapsed_time(e1) * 1e3)
    tb.sort()
    mbb = B * T * 7 * d * 2 / 1e6
    print(f"   backward: median {tb[len(tb) // 2]:7.1f} us  min {tb[0]:7.1f} us  {mbb:7.1f} MB  {mbb / tb[len(tb) // 2]:6.2f} TB/s")
    mb = B * T * 4 * d * 2 / 1e6
    print(f"B={B:5d} H={H:3d} hd={hd:4d}: median {ts[len(ts) // 2]:7.1f} us  min {ts[0]:7.1f} us  {mb:7.1f} MB  {mb / ts[len(ts) // 2] * 1e-3 * 1e3:6.2f} TB/s", flush=True)
