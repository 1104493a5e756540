#!/usr/bin/env python3
"""Micro-benchmark of gpt_adamw_ema_step on an arena of the mnist model's size (HIP events, interleaved rounds)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpt_amd
from gpt_amd import _lib as L
from gpt_amd.configs import gpt_kwargs
from gpt_amd.flat import FlatState

dev = "cuda:0"
m = gpt_amd.Gpt(**gpt_kwargs("mnist")).to(dev).set_compute_dtype("bf16")
flat = FlatState(m, torch.device(dev))
n = flat.numel
g, ea, es, ema = (torch.randn(n, device=dev) * 1e-3 for _ in range(4))
es.abs_()
seg, n_seg = flat.segments(L.BF16)
packed = flat.packed(L.BF16)
norm = torch.ones(1, device=dev)
lib = L.lib()
ts = []
for r in range(-2, 12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    L.check(lib.gpt_adamw_ema_step(L.ptr(flat.params), L.ptr(g), L.ptr(ea), L.ptr(es), L.ptr(ema), L.ptr(packed), L.BF16, L.ptr(seg), n_seg, n,
                                   L.ptr(norm), 0.75, 1.0, 4e-4, 0.9, 0.999, 1e-8, 0.1, 3, 0.9999, int(os.environ.get("ZG", "1")), L.stream()), "adamw")
    b.record()
    torch.cuda.synchronize()
    if r >= 0:
        ts.append(a.elapsed_time(b))
ts.sort()
bytes_ = n * (42 if os.environ.get("ZG", "1") == "1" else 38)
print(f"zero_grad={os.environ.get('ZG','1')}: median {ts[len(ts)//2]*1e3:.0f} us, min {ts[0]*1e3:.0f} us, {bytes_/ts[len(ts)//2]/1e9:.2f} TB/s")
