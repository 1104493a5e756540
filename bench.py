#!/usr/bin/env python3
"""bench.py - throughput of the G.pt denoiser hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--mode sample|train] [--dtype bf16|fp32] [--batch B]

A "step" (sample mode, the default) is ONE reverse-diffusion step of the DDPM p_sample_loop over one batch
of B parameter vectors per GPU: denoiser forward + in-kernel Philox noise + fused posterior update, inputs
resident in HBM.  Workload (BASELINE.json configs[1]): the pretrained `mnist_loss` G.pt shapes
(D = 7960, n_embd 1536, 16 heads, 12 layers, 26 tokens), random-initialised weights, synthetic inputs
of SURVEY.md section 8(d), B = 320 per GPU (the reference's evaluation batch, 16 nets x 20 prompts).
`value` = sampled parameter vectors per second under the full 1000-step sampler = B_total / (1000 x step time).
Rows are independent, so for N > 1 every rank samples its own B rows with no traffic (weak scaling) and the
ranks exchange only the final gather of the sampled vectors.

One JSON line is printed by rank 0.  Extra objects: `roofline` (the dominant kernel = the MFMA GEMM,
timed with HIP events recorded on its own stream in a second, instrumented pass of the same steps) and
`cpu_baseline` (the oracle - a CPU restatement of the reference's PyTorch path - timed on the host cores
on a bounded sample, rank 0 at N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import gpt_amd                                   # noqa: E402
from gpt_amd import _lib as L                    # noqa: E402
from gpt_amd import dp                           # noqa: E402
from gpt_amd.configs import CASES, gpt_kwargs    # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}          # dense MFMA TFLOP/s, /opt/skills/guides/MI355X_MICROARCH.md


def flops_per_forward(case):
    """Algorithmic FLOPs per sample per denoiser forward."""
    c = CASES[case]
    d, Lr, T, D = c["n_embd"], c["n_layer"], n_tokens(case), sum(c["parameter_sizes"])
    return 2 * T * Lr * 12 * d * d + 4 * Lr * T * T * d + 2 * d * (2 * D + 4 * 257) + 2 * d * D


def n_tokens(case):
    """26 tokens for the MNIST / CIFAR-10 plans (chunk_within_layer), 18 for Cartpole's (layer_by_layer: 7 + 7 tensors + 4 scalars)."""
    return 18 if case.startswith("cartpole") else 26


def make_model(case, dtype, device, seed=0):
    torch.manual_seed(seed)
    if case == "cifar_big":          # 2.7 B parameters: drawn on the device (the host's generator needs half a minute for them)
        with torch.device(device):
            m = gpt_amd.Gpt(**gpt_kwargs(case))
    else:
        m = gpt_amd.Gpt(**gpt_kwargs(case))
    with torch.no_grad():                                    # the reference zero-inits these (identity model)
        m.decoder.pos_emb.normal_(0, 0.02)
        for p in m.decoder.output_parameter_projections:
            p[0].weight.normal_(0, 0.02)
    return m.to(device).eval().set_compute_dtype(dtype)


def synth_inputs(case, B, device, seed):
    c = CASES[case]
    D = sum(c["parameter_sizes"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    x_prev = (torch.randn(B, D, generator=g) * c["w_std"]).to(device)
    loss_prev = (c["loss_lo"] + (c["loss_hi"] - c["loss_lo"]) * torch.rand(B, generator=g)).to(device)
    loss_target = torch.full((B,), float(c["loss_target"]), device=device)
    start = torch.randn(B, D, generator=g).to(device)
    return start, loss_target, loss_prev, x_prev


class SampleRunner:
    """Drives gpt_sample_loop over arbitrary step counts (wrapping around the 1000-step schedule)."""

    def __init__(self, model, diff, B, inputs, device, seed=1234):
        self.m, self.diff, self.B, self.dev, self.seed = model, diff, B, device, seed
        self.x, self.lt, self.lp, self.xp = inputs
        self.eng = model.engine(device)
        self.ws = self.eng.workspace(B)
        n = diff.num_timesteps
        self.tmap = (C.c_int64 * n)(*diff.timestep_map)
        self.t_model = torch.empty(B, dtype=torch.long, device=device)
        self.t_index = torch.empty(B, dtype=torch.long, device=device)
        self.model_out = torch.empty_like(self.x)
        self.noise = torch.empty_like(self.x)
        self.coef = diff.coef_table(device)
        self.cursor = n - 1

    def run(self, steps):
        n = self.diff.num_timesteps
        while steps > 0:
            k = min(steps, self.cursor + 1)
            first, last = self.cursor, self.cursor - k + 1
            L.check(L.lib().gpt_sample_loop(
                C.byref(self.eng.cmodel), C.byref(self.ws), L.ptr(self.x), L.ptr(self.lt), L.ptr(self.lp), 0,
                L.ptr(self.xp), L.ptr(self.coef), self.tmap, n, first, last, 1, 0, 0.0, 0.0, None, self.seed,
                L.ptr(self.t_model), L.ptr(self.t_index), L.ptr(self.model_out), L.ptr(self.noise), self.B,
                L.stream()), "gpt_sample_loop")
            steps -= k
            self.cursor = last - 1 if last > 0 else n - 1


def cpu_baseline(case, seconds_target=20.0):
    """The oracle's denoising step (forward + posterior update) on the host cores, bounded sample."""
    from oracle import denoiser as od, diffusion as odf
    c = CASES[case]
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "64"))))
    plan = od.Plan(c["parameter_sizes"], 128, 14, True, c["n_head"], c["n_embd"], c["n_layer"], c["chunk_size"],
                   c["split_policy"], True)
    sd = od.random_state_dict(plan, seed=0)
    tb = odf.Tables(1000)
    B = 64                                   # the CPU reference is timed at B = 64
    g = torch.Generator().manual_seed(1)
    x, xp = torch.randn(B, plan.D, generator=g), torch.randn(B, plan.D, generator=g) * 0.538
    lt, lp = torch.zeros(B), 0.3 + 2.1 * torch.rand(B, generator=g)
    nz = torch.randn(B, plan.D, generator=g)

    def step(i):
        t = torch.full((B,), i, dtype=torch.long)
        with torch.no_grad():
            out = od.forward(sd, plan, x, t, lt, lp, xp)
            return odf.posterior_step(tb, out, x, t, nz)[0]

    step(999)
    t0 = time.perf_counter()
    step(998)
    one = time.perf_counter() - t0
    reps = max(2, min(100, int(seconds_target / max(one, 1e-3))))     # a bounded sample: about 20 s of CPU work
    t0 = time.perf_counter()
    for i in range(reps):
        x = step(997 - i)
    dt = (time.perf_counter() - t0) / reps
    return {"value": B / dt / 1000.0, "unit": "vectors/s (1000-step DDPM, extrapolated from timed steps)",
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} denoising steps of the CPU oracle (PyTorch fp32) at B={B}, {case} shapes; "
                      f"{B / dt:.1f} sample-steps/s"}


def cpu_baseline_train(case, iters=3):
    """the reference's training iteration (main.py:32-81) restated on the oracle, on the host cores:
    q_sample -> oracle forward -> mean_flat MSE -> autograd backward -> clip_grad_norm_ -> torch AdamW (two groups) -> EMA, at
    B = 64, `iters` timed iterations after one untimed."""
    from oracle import denoiser as od, diffusion as odf
    c = CASES[case]
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "64"))))
    plan = od.Plan(c["parameter_sizes"], 128, 14, True, c["n_head"], c["n_embd"], c["n_layer"], c["chunk_size"],
                   c["split_policy"], True)
    sd = od.random_state_dict(plan, seed=0)
    names = [k for k in sd if k != "scalar_embedder.frequencies"]
    for k in names:
        sd[k] = sd[k].clone().requires_grad_(True)
    decay = [k for k in names if k.endswith("weight") and ".ln" not in k and "ln_" not in k]
    no_decay = [k for k in names if k not in set(decay)]
    opt = torch.optim.AdamW([{"params": [sd[k] for k in decay], "weight_decay": 0.1},
                             {"params": [sd[k] for k in no_decay], "weight_decay": 0.0}], lr=4e-4, betas=(0.9, 0.999))
    ema = {k: sd[k].detach().clone() for k in names}
    tb = odf.Tables(1000)
    B = 64
    g = torch.Generator().manual_seed(2)
    w_t = torch.randn(B, plan.D, generator=g) * c["w_std"]
    w_t1 = w_t + 0.1 * torch.randn(B, plan.D, generator=g)
    l_t, l_t1 = (c["loss_lo"] + (c["loss_hi"] - c["loss_lo"]) * torch.rand(B, generator=g) for _ in range(2))

    def iteration(i):
        t = torch.randint(0, 1000, (B,), generator=g)
        noise = torch.randn(B, plan.D, generator=g)
        mse, _, _ = odf.training_mse(tb, lambda x, tm: od.forward(sd, plan, x, tm, l_t1, l_t, w_t), w_t1, t, noise)
        loss = mse.mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([sd[k] for k in names], 0.75)
        opt.step()
        with torch.no_grad():
            torch._foreach_mul_(list(ema.values()), 0.9999)
            torch._foreach_add_(list(ema.values()), [sd[k].detach() for k in names], alpha=1 - 0.9999)
        return float(loss.detach())

    iteration(0)
    t0 = time.perf_counter()
    for i in range(iters):
        loss = iteration(1 + i)
    dt = (time.perf_counter() - t0) / iters
    return {"value": 1.0 / dt, "unit": "iters/s at B=64", "samples_per_s": B / dt, "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} training iterations of the CPU oracle (PyTorch fp32 autograd + torch AdamW + EMA) at B={B}, {case} shapes, "
                      f"{dt:.2f} s each; the GPU `train_weak` object runs B=128 per step", "final_loss": loss}


def hbm_kernels(case, B, device):
    """Achieved HBM GB/s of the elementwise kernels of one sampling step at the headline shapes (north_star: 'achieved HBM GB/s for
    the elementwise kernels'), each timed alone with events around 48 back-to-back launches on the current stream that rotate over four operand sets; algorithmic
    bytes per launch as DESIGN.md section 4 states them; peak = 8 TB/s (MI355X_MICROARCH.md)."""
    c = CASES[case]
    d, H, T, D = c["n_embd"], c["n_head"], 26, sum(c["parameter_sizes"])
    M = B * T
    lib, st = L.lib(), L.stream()
    NS = 4          # operand sets used in rotation (0.3 - 0.4 GB together): a launch must not find its operands in the 256 MB Infinity Cache
    gam, bet = torch.ones(d, device=device), torch.zeros(d, device=device)
    xs = [torch.randn(M, d, device=device) for _ in range(NS)]
    ys = [torch.empty(M, d, dtype=torch.bfloat16, device=device) for _ in range(NS)]
    qkvs = [torch.randn(M, 3 * d, device=device).bfloat16() for _ in range(NS)]
    atts = [torch.empty(M, d, dtype=torch.bfloat16, device=device) for _ in range(NS)]
    posts = [[torch.randn(B, D, device=device) for _ in range(3)] for _ in range(NS)]
    t = torch.full((B,), 500, dtype=torch.long, device=device)
    coef = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000).coef_table(device)
    runs = {
        "layernorm (fp32 row -> bf16 row)": (M * d * 6, lambda i: lib.gpt_layernorm(
            L.ptr(xs[i]), L.ptr(gam), L.ptr(bet), None, 0, L.ptr(ys[i]), L.BF16, None, None, M, d, 1e-5, st)),
        "attention (qkv -> heads' outputs)": (M * 3 * d * 2 + M * d * 2, lambda i: lib.gpt_attention(
            L.BF16, L.ptr(qkvs[i]), L.ptr(atts[i]), B, T, H, d // H, st)),
        "posterior update (model_out, x_t, noise -> x_{t-1})": (4 * B * D * 4, lambda i: lib.gpt_posterior_step(
            L.ptr(posts[i][0]), L.ptr(posts[i][1]), L.ptr(posts[i][2]), L.ptr(t), L.ptr(coef), 1, 0, 0.0, 0.0, L.ptr(posts[i][1]), None,
            B, D, st)),
    }
    out = {}
    for name, (nbytes, fn) in runs.items():
        res = {"bytes": nbytes}
        # "cold": the four operand sets in rotation (every launch reads from HBM); "warm": one set again and again (the Infinity
        # Cache holds it: in the sampling step these kernels read what the kernel before them just wrote, which lies in between)
        for tag, ns in (("cold", NS), ("warm", 1)):
            for i in range(NS):
                L.check(fn(i % ns), name)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(48):
                L.check(fn(i % ns), name)
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 48
            gbs = nbytes / (us * 1e-6) / 1e9
            res[tag] = {"us": us, "achieved_GBps": gbs, "frac_of_8TBps": gbs / 8000.0}
        out[name] = res
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher in front: one fresh child process per GPU, as the reference starts its own
    ranks (main.py:344-354, start_processes).  This parent has made no GPU call (nothing above touches the device before
    main() parses its arguments): it starts `python -m torch.distributed.run` as a child with the same arguments, passes
    the ranks' stdout (rank 0's one JSON line) and stderr through, and returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:                       # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="sample", choices=["sample", "train"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--case", default="mnist", choices=["mnist", "cifar", "cifar_big", "cartpole"])
    ap.add_argument("--data", default="synthetic", choices=["synthetic", "store"],
                    help="--mode train: 'store' draws every batch from an HBM-resident checkpoint store (g.pt_amd/data.py)")
    ap.add_argument("--augment", action="store_true",
                    help="train mode: on-device permutation augmentation of every checkpoint pair (configs[4])")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default 320 sample / 128 train)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the short training-step measurement ('train' object)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short fp32 / cifar sampling measurements ('fp32', 'cifar' objects)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed region of --steps steps is run this many times: `value` is the FIRST (the contract's region), "
                         "`spread` reports median / min / max over all of them")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    # seconds after which a run that is still going dumps every thread's stack and exits non-zero (a hung collective must leave a
    # trace, not a silent process); 0 disables.  The default run takes two to three minutes.
    wd = float(os.environ.get("GPT_BENCH_WATCHDOG", "900" if args.steps <= 2000 else "0"))
    if wd > 0:
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # GPT_BENCH_BACKEND=gloo GPT_BENCH_ONE_GPU=1: rehearsal of the N > 1 control flow of this file on a one-GPU machine (every
    # rank on GPU 0, exchanges host-staged over gloo: g.pt_amd/dp.py); the numbers of such a run mean nothing
    backend = os.environ.get("GPT_BENCH_BACKEND", "nccl")
    if os.environ.get("GPT_BENCH_ONE_GPU") == "1":
        local = 0
    install_term_guard(rank)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1 or os.environ.get("GPT_DP_REHEARSE") == "1":   # one-rank rehearsal of the RCCL calls: g.pt_amd/dp.py exchanging()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from gpt_amd.trainstep import bench_train
    if args.mode == "train":
        line = bench_train(args, device, world, rank)
    else:
        line = bench_sample(args, device, world, rank)
        if not args.no_extra and (args.case, args.dtype) == ("mnist", "bf16"):
            if rank == 0:
                line["hbm_kernels"] = guarded(lambda: hbm_kernels(args.case, args.batch or 320, device), rank, line, "hbm_kernels", 60.0)
            line["fp32"] = secondary_sample(args, device, world, rank, line, "fp32", "mnist", "fp32", args.batch or 320, 20, 3)
            # configs[0] / [1] name 64 vectors: the small-batch leg of the same workload
            line["b64"] = secondary_sample(args, device, world, rank, line, "b64", "mnist", "bf16", 64, 100, 10)
            line["cifar"] = secondary_sample(args, device, world, rank, line, "cifar", "cifar", "bf16", 1280, 20, 3)
        if not args.no_train:
            line["train"] = guarded_train(args, device, world, rank, line, "train", max(1, 1024 // world), "strong")
            # efficiency also at a fixed per-GPU batch: 128 pairs per GPU (the 8-GPU share of 1024) at every N
            line["train_weak"] = (dict(line["train"], scaling="weak") if 1024 // world == 128 else
                                  guarded_train(args, device, world, rank, line, "train_weak", 128, "weak"))
            if not args.no_extra:
                # configs[4]: CIFAR-10 shapes, scaled-up transformer (n_embd 3072, 24 heads, 24 layers: 2.7 B parameters), 550 / 8 = 68
                # pairs per GPU, every pair permuted on the device inside the timed step
                line["cifar_big"] = guarded_train(args, device, world, rank, line, "cifar_big", 68, "weak", limit_s=240.0,
                                                  case="cifar_big", augment=True, steps=6, warmup=2)
                if rank == 0 and world == 1 and not args.no_cpu_baseline:
                    line["cpu_baseline_train"] = guarded(lambda: cpu_baseline_train("mnist"), rank, line, "cpu_baseline_train", 400.0)
    if rank == 0:
        _PENDING["printed"] = True
        print(json.dumps(line), flush=True)
    if dp.exchanging():
        dist.barrier()
        dist.destroy_process_group()


# What rank 0 would still owe the caller if the process were ended now: the line measured so far and the leg that was running.
_PENDING = {"line": None, "key": None, "printed": False}


def install_term_guard(rank):
    """If another rank dies inside a secondary leg, the launcher ends the survivors with SIGTERM (SIGKILL half a minute later) -
    usually while rank 0's main thread sits inside a GPU wait or a collective, where a Python-level handler would never run.  The
    C-level handler only writes the signal number to a wake-up socket; a helper thread reads it, prints the headline line measured
    so far with an error object under the leg that was running, and leaves with a non-zero code.  Nothing is printed when no
    headline exists yet or the line is already out."""
    import signal
    import socket
    import threading
    if rank != 0 or threading.current_thread() is not threading.main_thread():
        return
    rd, wr = socket.socketpair()
    wr.setblocking(False)
    signal.signal(signal.SIGTERM, lambda *_: None)        # a Python handler must exist for the wake-up byte to be written
    signal.set_wakeup_fd(wr.fileno(), warn_on_full_buffer=False)
    install_term_guard.keep = (rd, wr)

    def watch():
        while True:
            got = rd.recv(16)
            if not got or signal.SIGTERM in got:
                break
        line, key = _PENDING["line"], _PENDING["key"] or "run"
        if line is not None and not _PENDING["printed"]:
            line[key] = {"error": "terminated (SIGTERM) during this measurement: another rank failed or the launcher gave up"}
            print(json.dumps(line, default=str), flush=True)
        sys.stderr.write(f"bench.py: SIGTERM during '{key}'\n")
        sys.stderr.flush()
        os._exit(143)

    threading.Thread(target=watch, daemon=True).start()


def guarded(fn, rank, line, key, limit_s):
    """Runs a secondary measurement under a watchdog.  If it does not finish within limit_s the headline line is printed
    with an error object under `key` (the line must never be lost), every thread's stack goes to stderr so that the hang
    can be root-caused, and the process exits NON-ZERO: a hung GPU step must not look like a successful run."""
    import faulthandler
    import threading

    def give_up():
        if rank == 0 and not _PENDING["printed"]:
            _PENDING["printed"] = True
            line[key] = {"error": f"did not finish within {limit_s:.0f} s (hung step or collective): process aborted, exit code 3"}
            print(json.dumps(line), flush=True)
        sys.stderr.write(f"bench.py: '{key}' measurement exceeded {limit_s:.0f} s; stacks follow\n")
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        sys.stderr.flush()
        os._exit(3)

    timer = threading.Timer(limit_s, give_up)
    timer.daemon = True
    timer.start()
    _PENDING.update(line=line, key=key)
    try:
        return fn()
    except Exception as e:  # noqa: BLE001 - the headline line must survive a failure of a secondary measurement
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        timer.cancel()


def guarded_train(args, device, world, rank, line, key, batch, scaling, limit_s=150.0, case=None, augment=None, steps=10, warmup=3):
    """BASELINE.json's second metric (train iters/sec, configs[3]) measured briefly after the headline, 10 timed steps:
    "train" at global batch 1024 = int(1024 / N) pairs per GPU as configs/train/mnist_loss.yaml runs it (strong scaling),
    "train_weak" at 128 pairs per GPU for every N."""
    from gpt_amd.trainstep import bench_train

    def run():
        targs = argparse.Namespace(**vars(args))
        targs.steps, targs.warmup, targs.batch = steps, warmup, batch
        if case is not None:
            targs.case = case
        if augment is not None:
            targs.augment = augment
        res = bench_train(targs, device, world, rank)
        keep = ("value", "unit", "ms_per_step", "steps", "warmup", "samples_per_s", "model_tflops_per_gpu", "final_loss", "first_step_check",
                "roofline")
        out = {k: res[k] for k in keep}
        out["model_flops_frac_of_peak"] = res["model_tflops_per_gpu"] / PEAK[args.dtype]
        out.update(metric=res["metric"], scaling=scaling, global_batch=res["config"]["global_batch"],
                   batch_per_gpu=res["config"]["batch_per_gpu"], workload=res["config"]["workload"])
        return out

    out = guarded(run, rank, line, key, limit_s)
    torch.cuda.empty_cache()
    return out


def secondary_sample(args, device, world, rank, line, key, case, dtype, batch, steps, warmup, limit_s=150.0):
    """The other stated sampling configurations as short objects of the same line (each with its own roofline):
    `fp32` = configs[1]'s fp32 leg (mnist shapes, B = 320), `cifar` = configs[2] at N = 1 (CIFAR-10 shapes, B = 1280, bf16)."""
    def run():
        sargs = argparse.Namespace(**vars(args))
        sargs.case, sargs.dtype, sargs.batch, sargs.steps, sargs.warmup, sargs.repeats = case, dtype, batch, steps, warmup, 1
        res = bench_sample(sargs, device, world, rank, cpu=False)
        keep = ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "denoiser_tflops_per_gpu", "roofline")
        out = {k: res[k] for k in keep if k in res}
        out.update(metric=res["metric"], workload=res["config"]["workload"], batch_per_gpu=batch)
        return out

    out = guarded(run, rank, line, key, limit_s)
    torch.cuda.empty_cache()
    return out


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else None


def pmc_traffic(case, B, dtype):
    """Bytes per GEMM launch behind L2 from the PMC passes committed under profiles/ (FETCH_SIZE x 2 on gfx950 +
    WRITE_SIZE, collected with rocprofv3 --pmc on this same command): counters cannot be read from inside the run, so
    the figure is only reported for the workload it was collected on, else null."""
    path = _latest_profile("gemm_traffic.json")
    if (case, B, dtype) != ("mnist", 320, "bf16") or path is None:
        return None
    with open(path) as f:
        return json.load(f)["traffic_bytes_per_launch"]


def pmc_mfma_util(case, B, dtype):
    """Fraction of the chip's matrix-pipe cycles that were busy while the GEMM kernels ran, from the PMC pass committed under
    profiles/ (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs), tools/pmc_mfma.py, same command): like
    `traffic` it cannot be read from inside the run, so it is reported for the workload it was collected on only."""
    path = _latest_profile("mfma_util.json")
    if (case, B, dtype) != ("mnist", 320, "bf16") or path is None:
        return None
    with open(path) as f:
        rows = [v for k, v in json.load(f).items() if "gemm256_kernel" in k or "gemm_nt_kernel" in k]
    busy = sum(v["mfma_busy_cycles"] for v in rows)
    cyc = sum(v["gui_active_cycles"] / 8.0 * 256 * 4 for v in rows)
    return busy / cyc if cyc else None


def sync_all(world):
    if dp.exchanging():
        dist.barrier()
    torch.cuda.synchronize()


def bench_sample(args, device, world, rank, cpu=True):
    case, B = args.case, args.batch or 320
    model = make_model(case, args.dtype, device)
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    runner = SampleRunner(model, diff, B, synth_inputs(case, B, device, seed=100 + rank), device,
                          seed=1234 * world + rank)      # one Philox stream per rank (Gpt/utils.py:73-75: seed * G + rank)
    D = model.D

    runner.run(args.warmup)
    sync_all(world)
    t0 = time.perf_counter()
    runner.run(args.steps)
    local_rows = runner.x
    if dp.exchanging():                                       # the only traffic: final gather of the samples
        local_rows = dp.gather_rows(runner.x, B * world)
    sync_all(world)
    elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(elapsed, device)
    assert local_rows.shape == (B * world, D) and bool(torch.isfinite(runner.x).all())
    ms_per_step = elapsed * 1e3 / args.steps
    # the same region again (no gather): one number says nothing about its own spread
    regions = [ms_per_step]
    for _ in range(max(0, getattr(args, "repeats", 1) - 1)):
        sync_all(world)
        t1 = time.perf_counter()
        runner.run(args.steps)
        sync_all(world)
        regions.append((time.perf_counter() - t1) * 1e3 / args.steps)
    vectors_per_s = B * world / (ms_per_step * 1e-3 * diff.num_timesteps)
    fwd_flops = flops_per_forward(case) * B
    line = {
        "metric": "sampled param-vectors/sec (DDPM)", "value": vectors_per_s, "unit": "vectors/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{'mnist_loss' if case == 'mnist' else 'cifar10_error' if case == 'cifar' else case} G.pt shapes "
                               f"(D={D}, n_embd={CASES[case]['n_embd']}, {CASES[case]['n_layer']} layers, {n_tokens(case)} tokens), "
                               f"1000-step DDPM p_sample_loop, B={B} vectors per GPU, random-init weights, "
                               f"in-kernel Philox noise; step = 1 reverse-diffusion step",
                   "batch_per_gpu": B, "diffusion_steps": diff.num_timesteps},
        "denoiser_tflops_per_gpu": fwd_flops / (ms_per_step * 1e-3) / 1e12,
    }
    if len(regions) > 1:
        srt = sorted(regions)
        line["spread"] = {"regions": len(regions), "steps_each": args.steps, "ms_per_step_median": srt[len(srt) // 2],
                          "ms_per_step_min": srt[0], "ms_per_step_max": srt[-1],
                          "note": "rank-local times of the same timed region repeated; `value` is the first region"}
    if not args.no_roofline:
        lib = L.lib()
        sync_all(world)
        L.check(lib.gpt_gemm_prof_enable(1), "prof")
        runner.run(min(args.steps, 40))
        torch.cuda.synchronize()
        ms, fl, nl = C.c_double(), C.c_double(), C.c_int()
        L.check(lib.gpt_gemm_prof_collect(C.byref(ms), C.byref(fl), C.byref(nl)), "prof")
        L.check(lib.gpt_gemm_prof_enable(0), "prof")
        achieved = fl.value / (ms.value * 1e-3) / 1e12
        line["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": PEAK[args.dtype], "unit": "TFLOP/s",
                            "frac": achieved / PEAK[args.dtype], "traffic": pmc_traffic(case, B, args.dtype),
                            "mfma_util_pmc": pmc_mfma_util(case, B, args.dtype),
                            "kernel": "gemm256_kernel (+ gemm_nt_kernel for row tails): the qkv / proj / fc1 / fc2 contractions of every layer",
                            "launches": nl.value,
                            "avg_launch_us": ms.value * 1e3 / max(nl.value, 1),
                            "flops_per_launch": fl.value / max(nl.value, 1)}
    if cpu and rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(case)
    return line


if __name__ == "__main__":
    main()
