#!/usr/bin/env python3
"""Probe (round 3): can weight-gradient TN launches hide in the gaps of the backward's dgrad chain?  Times the B-row backward
alone, 12 extra grouped TN launches (one Block's four weight gradients each) alone, and both at once on two streams.
usage: overlap_probe.py [B]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpt_amd
from gpt_amd import _lib as L
from gpt_amd.backward import backward_into, forward_keep, _ensure_training_state
from bench import make_model
from gpt_amd.trainstep import synthetic_pairs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
lib = L.lib()
m = make_model("mnist", "bf16", dev).train()
w_t, w_t1, l_t, l_t1 = synthetic_pairs("mnist", B, dev, 1)
t = torch.randint(0, 1000, (B,), device=dev)
x, tm, lt, lp, xp = m._prepare(w_t1, t, l_t1, l_t, w_t)
eng = m.engine(dev)
_ensure_training_state(eng)
eng, ws, out = forward_keep(m, x, tm, lt, lp, xp)
d_out = torch.randn_like(out) * 1e-2
buf = torch.zeros(eng.arena.numel, device=dev)
R, d = B * 26, 1536
shapes = [(d, 4 * d), (4 * d, d), (3 * d, d), (d, d)]
Ps = [torch.randn(R, i, device=dev).bfloat16() for i, _ in shapes]
Qs = [torch.randn(R, j, device=dev).bfloat16() for _, j in shapes]
Cs = [torch.zeros(i, j, device=dev) for i, j in shapes]
n = 4
vp, i64, i32 = C.c_void_p * n, C.c_int64 * n, C.c_int * n
args = (n, vp(*[p.data_ptr() for p in Ps]), i64(*[i for i, _ in shapes]), vp(*[q.data_ptr() for q in Qs]), i64(*[j for _, j in shapes]),
        vp(*[c.data_ptr() for c in Cs]), i64(*[j for _, j in shapes]), i32(*[i for i, _ in shapes]), i32(*[j for _, j in shapes]), R, 0)
side = torch.cuda.Stream(device=dev, priority=0)
main = torch.cuda.Stream(device=dev, priority=-1)


def bwd():
    backward_into(eng, ws, x, xp, d_out, buf, overwrite=True)


def tn(k=12):
    for _ in range(k):
        L.check(lib.gpt_gemm_tn_grouped(*args, L.stream()), "tn")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def both():
    s0 = torch.cuda.current_stream()
    main.wait_stream(s0)
    side.wait_stream(s0)
    with torch.cuda.stream(main):
        bwd()
    with torch.cuda.stream(side):
        tn()
    s0.wait_stream(main)
    s0.wait_stream(side)


t1, t2, t3 = timed(bwd), timed(tn), timed(both)
print(f"B={B}: backward alone {t1:.3f} ms, 12 grouped TN launches alone {t2:.3f} ms, both on two streams {t3:.3f} ms "
      f"(serial {t1 + t2:.3f}; hidden {t1 + t2 - t3:.3f} ms)")
