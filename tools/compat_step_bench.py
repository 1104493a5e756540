#!/usr/bin/env python3
"""ms per step of the per-step sampling path (p_sample_loop_progressive: one Gpt.forward + one fused posterior kernel per Python
iteration, the operand pack guarded by the device-side checksum) against the one-call loop.  usage: compat_step_bench.py [B]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpt_amd
from bench import make_model, synth_inputs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
m = make_model("mnist", "bf16", dev)
diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000, timestep_respacing="100")
start, lt, lp, xp = synth_inputs("mnist", B, dev, 1)
kw = dict(loss_target=lt, loss_prev=lp, x_prev=xp)
for name in ("progressive", "fused"):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "progressive":
            for out in diff.p_sample_loop_progressive(m, (B, m.D), noise=start.clone(), clip_denoised=False, model_kwargs=kw, device=dev):
                pass
        else:
            diff.p_sample_loop(m, (B, m.D), noise=start.clone(), clip_denoised=False, model_kwargs=kw, device=dev, seed=1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / diff.num_timesteps
    print(f"B={B} {name}: {dt:.3f} ms per step")
