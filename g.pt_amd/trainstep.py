"""One G.pt optimisation step, fused: the sequence of main.py:32-81 (run_diffusion_vlb + train_epoch body) without
the autograd graph.

    lr schedule -> t ~ U{0..999} -> x_t = q_sample(w_t1, t) -> denoiser forward (activations kept) -> per-row MSE
    -> hand-written backward into ONE flat fp32 gradient arena -> bucketed RCCL all-reduce of the arena (overlapped
    with the rest of the backward) -> global-norm clip -> AdamW (2 groups) -> EMA -> packed low-precision weights.

`TrainStep` is what `bench.py --mode train` times; `diffusion.training_losses(model, ...)["loss"].backward()` with a
stock torch optimizer (the reference's own loop) runs through g.pt_amd/backward.py and gives the same numbers
(tests/test_hip_training.py pins both against the reference's fixtures).
"""
import ctypes as C
import math
import os
import time
from copy import deepcopy

import numpy as np
import torch
import torch.distributed as dist

from . import _lib as L
from . import dp
from .backward import _ensure_training_state, backward_into, can_overwrite_wgrads, forward_keep
from .ddpm import ModelMeanType
from .flat import FlatState


def lr_at(base_lr, num_ep, warmup_epochs, warmup_factor, cur_epoch, schedule="cos"):
    """Gpt/utils.py:103-124: cosine (or fixed) schedule with a linear warm-up factor."""
    lr = 0.5 * base_lr * (1.0 + math.cos(math.pi * cur_epoch / num_ep)) if schedule == "cos" else base_lr
    if cur_epoch < warmup_epochs:
        alpha = cur_epoch / warmup_epochs
        lr *= warmup_factor * (1.0 - alpha) + alpha
    return lr


@torch.no_grad()
def accumulate(ema, model, decay=0.9999):
    """Gpt/utils.py:146-152 as two multi-tensor launches."""
    pe = [p.data for _, p in sorted(ema.named_parameters())]
    pm = [p.data for _, p in sorted(model.named_parameters())]
    torch._foreach_mul_(pe, decay)
    torch._foreach_add_(pe, pm, alpha=1 - decay)


class TrainStep:
    """One model replica's training state: flat arenas for parameters, gradients, AdamW moments and the EMA copy.

    fused=True (default): clip + AdamW + EMA + operand re-packing run as gpt_grad_sumsq + ONE gpt_adamw_ema_step pass.
    fused=False: the reference's own sequence on torch ops (clip_grad_norm_, torch.optim.AdamW from
    configure_optimizers, accumulate) over the same gradient arena - the cross-check of the fused pass."""

    def __init__(self, model, diffusion, lr=4e-4, wd=0.1, betas=(0.9, 0.999), eps=1e-8, grad_clip=0.75, ema_decay=0.9999,
                 use_ema=True, bucket_layers=None, fused=True, overlap=True, shard=None, grad_payload=None,
                 overlap_update=None):
        self.model, self.diffusion = model, diffusion
        self.device = model.decoder.pos_emb.device
        self.lr, self.wd, self.betas, self.eps = lr, wd, betas, eps
        self.grad_clip, self.ema_decay, self.fused, self.overlap = grad_clip, ema_decay, fused, overlap
        self._handles = []
        self._side, self._t_ready = None, None
        # the Blocks' share of the optimizer pass on a side stream, layer by layer, beside the NEXT step's forward (see update());
        # opt-in (GPT_OPT_OVERLAP=1): whoever reads parameters, EMA or operands other than through this engine's forward or
        # state_dict() has to call join() first
        if overlap_update is None:
            overlap_update = os.environ.get("GPT_OPT_OVERLAP", "0") != "0"
        self.overlap_update = bool(overlap_update and fused)
        self._opt_done = None
        self.world = dp.get_world_size()
        self.flat = FlatState(model, self.device)
        if dp.exchanging():                             # rank 0's weights everywhere (DDP construction, main.py:231-236)
            dist.broadcast(self.flat.params, src=0)
        self.ema = None
        if use_ema:
            self.ema = deepcopy(model)
            for p in self.ema.parameters():
                p.requires_grad = False
        self.flat_ema = FlatState(self.ema, self.device) if use_ema else None
        self.eng = model.engine(self.device)
        self.eng.managed = fused            # the fused pass rewrites the packed operands with the parameters
        _ensure_training_state(self.eng)
        self.code = self.eng.code
        self.arena = self.flat.arena
        self.grads = self.flat.like()
        for p, g in zip(model.parameters(), self.flat.views(self.grads, model)):
            p.grad = g
        # data-parallel optimizer sharding (fused step only; GPT_DP_SHARD=0 keeps every rank's full update)
        if shard is None:
            shard = os.environ.get("GPT_DP_SHARD", "1") != "0"
        explicit = shard is True
        self.shard = bool(shard and fused and dp.exchanging())
        self.grad_payload = grad_payload or os.environ.get("GPT_DP_GRAD_PAYLOAD", "fp32")
        assert self.grad_payload in ("fp32", "bf16")
        self.rank = dp.get_rank()
        if fused and not self.shard:
            self.exp_avg, self.exp_avg_sq = self.flat.like(), self.flat.like()
        if fused:
            self.norm_scratch = torch.zeros(1024 + 8, dtype=torch.float32, device=self.device)
            self.optimizer = None
        else:
            self.optimizer = model.configure_optimizers(lr=lr, wd=wd, betas=betas)
        if bucket_layers is None:
            # 2 layers = 226 MB (MNIST shapes) per all-reduce: large enough for RCCL's ring to run at link rate, and the first one
            # goes in flight after 1/6 of the backward instead of 1/3 (A/B on a multi-GPU node: GPT_DP_BUCKET_LAYERS)
            bucket_layers = int(os.environ.get("GPT_DP_BUCKET_LAYERS", "2"))
        assert bucket_layers >= 1
        self.buckets = self._make_buckets(bucket_layers)
        if self.shard and any((hi - lo) % (4 * self.world) for lo, hi, _, region in self.buckets if region == "S"):
            # S buckets hold whole Linear weights (multiples of 64 elements): any world <= 16 that is a power of two divides them;
            # 3, 6, 12 or more than 16 ranks (or an n_embd that is no multiple of 64) may not
            assert not explicit, f"shard=True needs every S bucket to split evenly over {self.world} ranks (16-byte chunks)"
            import warnings
            warnings.warn(f"TrainStep: the weight buckets do not split evenly over {self.world} ranks; "
                          "falling back to the replicated all-reduce step")
            self.shard = False
            self.exp_avg, self.exp_avg_sq = self.flat.like(), self.flat.like()
        if self.shard:
            self._init_shards()
        elif self.overlap_update:
            self.model.register_state_dict_pre_hook(lambda *a, **k: self.join())
            if self.ema is not None:
                self.ema.register_state_dict_pre_hook(lambda *a, **k: self.join())
                self.ema.register_forward_pre_hook(lambda *a, **k: self.join())
        if fused:
            # anything that makes the managed engine re-derive its operands from the fp32 parameters (a version bump behind the
            # step's back) must first see whole, settled parameters: the other ranks' shards and the side-stream pass
            self.eng.owner_sync = self.sync_params
        self.tmap = torch.tensor(diffusion._timestep_map(), device=self.device, dtype=torch.long)
        self.iteration = 0

    def join(self):
        """Makes the current stream wait for an optimizer pass still running on the side stream (overlap_update)."""
        if self._opt_done is not None:
            torch.cuda.current_stream().wait_event(self._opt_done)
            self._opt_done = None

    def _make_buckets(self, bucket_layers):
        """Contiguous arena ranges (lo, hi, stage_end, region), in the order the backward finishes them (complete once the
        backward stages < stage_end ran): the decoder Linears, layer groups of weights from the last block to the first
        (region S of the arena), then region R (biases, LayerNorm, pos_emb, encoder Linears)."""
        ar, Lr = self.arena, self.model.decoder.n_layer
        starts = list(ar.s_layer_start) + [ar.s_dec_start]
        out = [(ar.s_dec_start, ar.s_end, 1, "S")]
        hi = Lr
        while hi > 0:
            # the blocks the backward reaches last go out one by one: what is still on the wire when the backward ends is
            # the exposed tail of the exchange (113 MB instead of 226 MB at the mnist shapes)
            lo = max(0, hi - (bucket_layers if hi > bucket_layers + 1 else 1))
            out.append((starts[lo], starts[hi], Lr - lo + 1, "S"))
            hi = lo
        out.append((ar.r_start, ar.numel, Lr + 2, "R"))
        return out

    # ---- data-parallel sharding of the optimizer (ZeRO-1 style) ------------------------------------------------------------
    # Replicated step (what DistributedDataParallel + a per-rank optimizer do in the reference, main.py:229-240, 66-74): every
    # rank all-reduces all gradients (ring: reduce-scatter + all-gather of 4 P bytes) and then runs the SAME optimizer pass
    # over all P parameters (42 bytes of HBM traffic per parameter: 2.85 ms of a 14 ms step at B = 128, on every rank).
    # Sharded step: for every S bucket the fp32 gradient sum is reduce-scattered (rank r receives chunk r), the fused
    # clip + AdamW + EMA pass updates that 1/world of the parameters (moments and EMA exist for the own shard only) and writes
    # their bf16 operand copies, and the OPERANDS are all-gathered (bf16: half the bytes of the all-reduce's second half;
    # fp32 compute mode gathers the fp32 parameters).  Region R (1 % of the elements) stays all-reduced and replicated.
    # fp32 parameters (and the EMA) outside the own shard are refreshed lazily: sync_params() / sync_ema(), which the
    # state_dict() hooks call, so checkpoints are always whole.
    def _init_shards(self):
        w, r = self.world, self.rank
        self.s_chunks = []                   # per S bucket: (lo, hi, c, own_lo, g_off)
        g_off = 0
        for lo, hi, _, region in self.buckets:
            if region != "S":
                continue
            assert (hi - lo) % (4 * w) == 0          # checked (with a fallback to the replicated step) in __init__
            c = (hi - lo) // w
            self.s_chunks.append((lo, hi, c, lo + r * c, g_off))
            g_off += c
        self.n_own, self.n_rep = g_off, self.arena.numel - self.arena.r_start
        dev = self.device
        self.gshard = torch.zeros(self.n_own, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.n_own + self.n_rep, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_own + self.n_rep, dtype=torch.float32, device=dev)
        self.seg_tables = [self.flat.segment_table(self.code, own, own + c) for _, _, c, own, _ in self.s_chunks]
        self.seg_rep = self.flat.segment_table(self.code, self.arena.r_start, self.arena.numel)
        if self.grad_payload == "bf16":      # bf16 gradient payload: staging copies of the S region and of the own shard
            self.g16 = torch.zeros(self.arena.s_end, dtype=torch.bfloat16, device=dev)
            self.gshard16 = torch.zeros(self.n_own, dtype=torch.bfloat16, device=dev)
        self._params_stale = self._ema_stale = False
        self._gather_events = self._operands_done = None
        self.model.register_state_dict_pre_hook(lambda *a, **k: self.sync_params())
        if self.ema is not None:
            self.ema.register_state_dict_pre_hook(lambda *a, **k: self.sync_ema())
            # a forward through the EMA copy mid-training (the reference's periodic EMA evaluation, main.py:303-308) goes through
            # an unmanaged engine that repacks from flat_ema.params: the other ranks' shards must be there first
            self.ema.register_forward_pre_hook(lambda *a, **k: self.sync_ema())

    def _launch_bucket(self, lo, hi, region):
        """Puts the gradient exchange of one finished bucket in flight."""
        if not self.shard or region == "R":
            self._handles.extend(dp.all_reduce_ranges(self.grads, [(lo, hi)]))
            return
        _, _, c, _, g_off = next(ch for ch in self.s_chunks if ch[0] == lo)
        if self.grad_payload == "bf16":
            L.check(L.lib().gpt_f32_to_bf16(L.ptr(self.grads[lo:hi]), L.ptr(self.g16[lo:hi]), hi - lo, L.stream()), "gpt_f32_to_bf16")
            self._handles.append(dp.reduce_scatter_into(self.gshard16[g_off:g_off + c], self.g16[lo:hi]))
        else:
            self._handles.append(dp.reduce_scatter_into(self.gshard[g_off:g_off + c], self.grads[lo:hi]))

    def _gather(self, buf, stale_attr):
        """All-gathers every S bucket of an arena-shaped buffer in place from the ranks' own chunks."""
        self._wait_operands()
        hs = [dp.all_gather_into(buf[lo:hi], buf[own:own + c]) for lo, hi, c, own, _ in self.s_chunks]
        dp.wait_all(hs)
        setattr(self, stale_attr, False)

    def _gather_operands(self, buf):
        """The all-gather of the updated operands, OFF the compute stream: the buckets are gathered in forward order (block 0
        first, the decoder Linears last) on the side stream, each followed by an event, and the engine's forward waits for a
        block's event right before that block (gpt_layer_t.ready) - the next step's q_sample, encoder and first blocks run
        while the later buckets are still on the wire.  GPT_DP_GATHER_OVERLAP=0: the compute stream waits for all of them here."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        side, ar, Lr = self._side, self.arena, self.model.decoder.n_layer
        if self._gather_events is None:
            self._gather_events = [torch.cuda.Event() for _ in self.s_chunks]
        side.wait_stream(main)                       # the optimizer pass that wrote the own chunks
        layer_ev, dec_ev = [None] * Lr, None
        with torch.cuda.stream(side):
            for (lo, hi, c, own, _), ev in sorted(zip(self.s_chunks, self._gather_events), key=lambda z: z[0][0]):
                h = dp.all_gather_into(buf[lo:hi], buf[own:own + c])
                if h is not None:
                    h.wait()                         # the side stream, not the compute stream, waits for the collective
                ev.record(side)
                for l in range(Lr):
                    if lo <= ar.s_layer_start[l] < hi:
                        layer_ev[l] = ev
                if lo <= ar.s_dec_start < hi:
                    dec_ev = ev
        self._operands_done = side.record_event()
        if os.environ.get("GPT_DP_GATHER_OVERLAP", "1") == "0":      # A/B measurements
            self._wait_operands()
            return
        self.eng.set_ready(layer_ev, dec_ev)

    def _wait_operands(self):
        """Makes the compute stream wait for an operand all-gather still in flight (anything but the engine's forward that
        touches the parameters or their operand copies: the lazy fp32 gathers, state_dict())."""
        if getattr(self, "_operands_done", None) is not None:
            torch.cuda.current_stream().wait_event(self._operands_done)
            self._operands_done = None

    @torch.no_grad()
    def sync_params(self):
        """fp32 parameters of the other ranks' shards (bf16 compute mode gathers only the bf16 operands per step).
        A COLLECTIVE under the sharded step when the shards are stale: every rank must call it (and so everything that
        calls it: state_dict() of the model, a re-pack of the engine) at the same point; a rank-0-only checkpoint is preceded
        by sync_params() / sync_ema() on all ranks."""
        self.join()
        if self.shard:
            self._wait_operands()
            if self._params_stale:
                self._gather(self.flat.params, "_params_stale")

    @torch.no_grad()
    def sync_ema(self):
        self.join()
        if self.shard and self._ema_stale and self.flat_ema is not None:
            self._gather(self.flat_ema.params, "_ema_stale")

    def set_lr(self, lr):
        self.lr = lr
        if self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr

    @property
    def grad_norm_sq(self):
        return self.norm_scratch[1024:1025]

    @torch.no_grad()
    def loss_and_grads(self, w_t, w_t1, loss_t, loss_t1, t, weights=None, noise=None):
        """run_diffusion_vlb (main.py:32-49) + backward of loss = mean(mse * weights) into the gradient arena
        (which must be zero on entry).  Returns the per-sample mse."""
        lib, dev = L.lib(), self.device
        model, diff = self.model, self.diffusion
        B, D = w_t1.shape
        if noise is None:
            noise = torch.empty_like(w_t1)
            L.check(lib.gpt_randn(L.ptr(noise), noise.numel(), 0x5EED, self.iteration * self.world + dp.get_rank(),
                                  L.stream()), "gpt_randn")
        x_t = diff.q_sample(w_t1, t, noise=noise)
        x, tm, lt, lp, xp = model._prepare(x_t, self.tmap[t], loss_t1, loss_t, w_t)
        eng, ws, out = forward_keep(model, x, tm, lt, lp, xp)
        target = w_t1 if diff.model_mean_type == ModelMeanType.START_X else noise
        mse = torch.empty(B, dtype=torch.float32, device=dev)
        L.check(lib.gpt_mse_rows(L.ptr(target), L.ptr(out), L.ptr(mse), B, D, L.stream()), "gpt_mse_rows")
        g_rows = torch.full((B,), 1.0 / B, device=dev) if weights is None else (weights.float() / B).contiguous()
        d_out = torch.empty_like(out)
        L.check(lib.gpt_mse_rows_bwd(L.ptr(target), L.ptr(out), L.ptr(g_rows), L.ptr(d_out), B, D, L.stream()),
                "gpt_mse_rows_bwd")
        # DDP's gradient exchange (main.py:229-240): the SUM all-reduce of a bucket goes in flight on RCCL's stream as
        # soon as the backward stages that fill it are enqueued, and overlaps with the remaining stages
        self._handles = []
        schedule = None
        if dp.exchanging() and self.overlap:
            schedule = [(stage_end, (lambda lo=lo, hi=hi, region=region: self._launch_bucket(lo, hi, region)))
                        for lo, hi, stage_end, region in self.buckets]
        if self._t_ready is not None:
            torch.cuda.current_stream().wait_event(self._t_ready)
            self._t_ready = None
        # the Blocks' weight gradients (95 % of the arena) are STORED by the grouped weight-gradient launch where it applies:
        # no read-modify-write of the arena in its epilogue, and update() zeroes only what is still accumulated
        self._overwrote = self.fused and can_overwrite_wgrads(eng, B)
        # one GPU, store-mode weight gradients: their share of |g|^2 comes out of the TN launch's accumulators (under data
        # parallelism the norm is that of the SUMMED gradients, so it has to wait for the exchange)
        self._tile_sumsq = None
        if self._overwrote and schedule is None and self.grad_clip > 0 and os.environ.get("GPT_WGRAD_SUMSQ", "1") != "0":
            self._tile_sumsq = self.norm_scratch[1028:1029]
            self._tile_sumsq.zero_()
        backward_into(eng, ws, x, xp, d_out, self.grads, schedule, overwrite=self._overwrote, sumsq=self._tile_sumsq)
        return mse

    @torch.no_grad()
    def exchange(self):
        """Completes the gradient exchange (launches it here when not overlapped); the 1/world scale is folded into the
        optimizer pass."""
        if dp.exchanging() and not self.overlap:
            for lo, hi, _, region in self.buckets:
                self._launch_bucket(lo, hi, region)
        dp.wait_all(self._handles)
        self._handles = []
        if self.shard and self.grad_payload == "bf16":
            L.check(L.lib().gpt_bf16_to_f32(L.ptr(self.gshard16), L.ptr(self.gshard), self.n_own, L.stream()), "gpt_bf16_to_f32")

    @torch.no_grad()
    def update(self):
        """clip -> AdamW -> EMA -> operand copies (main.py:66-74)."""
        self.iteration += 1
        model = self.model
        if not self.fused:
            if self.world > 1:
                self.grads.mul_(1.0 / self.world)
            if self.grad_clip > 0:
                torch.nn.utils.clip_grad_norm_(model.parameters(), self.grad_clip)
            self.optimizer.step()
            if self.ema is not None:
                accumulate(self.ema, model, self.ema_decay)
            self.grads.zero_()
            return
        lib, st = L.lib(), L.stream()
        n = self.flat.numel
        packed = self.flat.packed(self.code)
        ema = self.flat_ema.params if self.flat_ema is not None else None

        def adamw(lo, count, grads, m_off, seg, n_seg, zero_grad, st=st):
            """The fused pass over arena elements [lo, lo + count); moments at m_off of the moment buffers."""
            L.check(lib.gpt_adamw_ema_step(
                L.ptr(self.flat.params[lo:]), L.ptr(grads), L.ptr(self.exp_avg[m_off:]), L.ptr(self.exp_avg_sq[m_off:]),
                L.ptr(ema[lo:]) if ema is not None else None, L.ptr(packed), self.code, L.ptr(seg), n_seg, count,
                L.ptr(self.grad_norm_sq), float(self.grad_clip), 1.0 / self.world, self.lr, self.betas[0], self.betas[1],
                self.eps, self.wd, self.iteration, self.ema_decay, zero_grad, st), "gpt_adamw_ema_step")

        if not self.shard:
            if self.grad_clip > 0 and getattr(self, "_tile_sumsq", None) is not None:
                cut = self.arena.s_dec_start         # [0, cut): the Blocks' weights, summed by the TN launches already
                part = self.norm_scratch[1026:1027]
                L.check(lib.gpt_grad_sumsq(L.ptr(self.grads[cut:]), n - cut, L.ptr(self.norm_scratch), L.ptr(part), st), "gpt_grad_sumsq")
                torch.add(part, self._tile_sumsq, out=self.grad_norm_sq)
            elif self.grad_clip > 0:
                L.check(lib.gpt_grad_sumsq(L.ptr(self.grads), n, L.ptr(self.norm_scratch), L.ptr(self.grad_norm_sq), st),
                        "gpt_grad_sumsq")
            overwrote = getattr(self, "_overwrote", False)
            cut = self.arena.s_dec_start
            if overwrote or self.overlap_update:
                if not hasattr(self, "_seg_split"):
                    self._seg_split = (self.flat.segment_table(self.code, 0, cut), self.flat.segment_table(self.code, cut, n))
                (sa, na), (sb, nb) = self._seg_split
            # store mode: nothing to zero in the Blocks' weights (the next backward overwrites them); the rest still accumulates
            zero_blocks = 0 if overwrote else 1
            if self.overlap_update:
                # The optimizer pass is HBM-bound (38 - 42 bytes per parameter, 2.4 ms for the Blocks' weights at the mnist
                # shapes) and the forward MFMA-bound: the Blocks' share runs on the side stream one layer at a time, each
                # followed by an event the NEXT forward waits for right before that block (gpt_layer_t.ready, as for the
                # sharded step's operand gather); everything the encoder and the blocks' biases / LayerNorms need (region R
                # and the decoder Linears, 5 % of the arena) is updated on the compute stream first.  The next backward
                # waits for the whole side stream (_t_ready below) before it writes the gradients these launches read / zero.
                adamw(cut, n - cut, self.grads[cut:], cut, sb, nb, 1)
                starts = list(self.arena.s_layer_start) + [cut]
                if not hasattr(self, "_seg_layers"):
                    self._seg_layers = [self.flat.segment_table(self.code, starts[l], starts[l + 1]) for l in range(len(starts) - 1)]
                    self._layer_events = [torch.cuda.Event() for _ in self._seg_layers]
                main = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(main)
                with torch.cuda.stream(self._side):
                    sst = L.stream()
                    for l, ((sl, nl), ev) in enumerate(zip(self._seg_layers, self._layer_events)):
                        adamw(starts[l], starts[l + 1] - starts[l], self.grads[starts[l]:], starts[l], sl, nl, zero_blocks, st=sst)
                        ev.record(self._side)
                self._opt_done = self._side.record_event()
                self.eng.set_ready(list(self._layer_events), None)
            elif overwrote:
                adamw(0, cut, self.grads, 0, sa, na, 0)
                adamw(cut, n - cut, self.grads[cut:], cut, sb, nb, 1)
            else:
                seg, n_seg = self.flat.segments(self.code)
                adamw(0, n, self.grads, 0, seg, n_seg, 1)
        else:
            r0 = self.arena.r_start
            if self.grad_clip > 0:
                # |g|^2 = the ranks' own S chunks summed over the ranks + region R (identical on every rank after its all-reduce)
                part = self.norm_scratch[1026:1028]
                L.check(lib.gpt_grad_sumsq(L.ptr(self.gshard), self.n_own, L.ptr(self.norm_scratch), L.ptr(part[0:1]), st), "gpt_grad_sumsq")
                if dp._host_staged(part):
                    host = part[0:1].cpu()
                    dist.all_reduce(host)
                    part[0:1].copy_(host)
                else:
                    dist.all_reduce(part[0:1])
                L.check(lib.gpt_grad_sumsq(L.ptr(self.grads[r0:]), self.n_rep, L.ptr(self.norm_scratch), L.ptr(part[1:2]), st), "gpt_grad_sumsq")
                torch.add(part[0:1], part[1:2], out=self.grad_norm_sq)
            for (lo, hi, c, own, g_off), (seg, n_seg) in zip(self.s_chunks, self.seg_tables):
                adamw(own, c, self.gshard[g_off:], g_off, seg, n_seg, 0)
            adamw(r0, self.n_rep, self.grads[r0:], self.n_own, self.seg_rep[0], self.seg_rep[1], 0)
            if getattr(self, "_overwrote", False):
                self.grads[self.arena.s_dec_start:].zero_()      # decoder Linears + region R: all that is still accumulated
            else:
                self.grads.zero_()
            # the updated operands of every rank's shard: bf16 copies in bf16 compute mode (half the bytes), else the fp32 parameters
            if self.code == L.BF16:
                self._gather_operands(packed)
                self._params_stale = True      # the fp32 parameters of the other shards follow on demand (sync_params)
            else:
                self._gather_operands(self.flat.params)
                self._params_stale = False
            self._ema_stale = ema is not None
        # the step wrote the parameters and this engine's packed operands together, behind torch's back (the EMA copy's own
        # engines re-derive their operands on every use: denoiser.py _Engine.refresh)
        self.eng.touched()
        # transposed dgrad operands of the new weights: needed by the NEXT backward only, so the 48 transposes run on a side
        # stream beside the next forward (memory-bound beside MFMA-bound; 0.9 ms of an 18 ms step at B = 128 otherwise)
        if os.environ.get("GPT_TRAIN_SIDE_STREAM", "1") == "0":      # A/B measurements
            self.join()
            if self.shard:
                self._wait_operands()
            _ensure_training_state(self.eng)
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            _ensure_training_state(self.eng)
        self._t_ready = self._side.record_event()

    def step(self, w_t, w_t1, loss_t, loss_t1, t, weights=None, noise=None):
        """One optimisation step on one micro-batch; returns the per-sample mse (device tensor, no host sync)."""
        mse = self.loss_and_grads(w_t, w_t1, loss_t, loss_t1, t, weights, noise)
        self.exchange()
        self.update()
        return mse


def synthetic_pairs(case, B, device, seed):
    """checkpoint pairs of the MNIST shapes."""
    from .configs import CASES
    c = CASES[case]
    D = sum(c["parameter_sizes"])
    g = torch.Generator(device="cpu").manual_seed(seed)
    w_t = torch.randn(B, D, generator=g) * c["w_std"]
    w_t1 = w_t + 0.1 * torch.randn(B, D, generator=g)
    lo, hi = c["loss_lo"], c["loss_hi"]
    loss_t = (lo + (hi - lo) * torch.rand(B, generator=g, dtype=torch.float64))
    loss_t1 = (lo + (hi - lo) * torch.rand(B, generator=g, dtype=torch.float64))
    return [v.to(device) for v in (w_t, w_t1, loss_t, loss_t1)]


def c_std(case):
    from .configs import CASES
    return CASES[case]["w_std"]


def bench_train(args, device, world, rank):
    """bench.py --mode train: configs/train/mnist_loss.yaml hyper-parameters on synthetic checkpoint pairs."""
    import gpt_amd
    from bench import flops_per_forward, make_model, sync_all, PEAK
    from .configs import CASES
    case = args.case
    B = args.batch or 128                      # int(1024 / 8): the per-GPU batch of the reference's 8-GPU run
    torch.manual_seed(0)                       # same initial weights on every rank (then broadcast anyway)
    model = make_model(case, args.dtype, device).train()
    diff = gpt_amd.create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    sampler = gpt_amd.UniformSampler(diff)
    np.random.seed(world + rank)               # Gpt/utils.py:73-75 with rng_seed = 1
    mnist = case.startswith("mnist")
    # train.grad_clip of configs/train/mnist_loss.yaml / cartpole.yaml (no clipping) / cifar10_error.yaml
    clip = 0.75 if mnist else 0.0 if case.startswith("cartpole") else 0.1
    cfg_name = ("configs/train/mnist_loss.yaml" if mnist else "configs/train/cartpole.yaml" if case.startswith("cartpole")
                else "configs/train/cifar10_error.yaml")
    if case == "cifar_big":
        c = CASES[case]
        cfg_name += f" with a scaled-up transformer (n_embd {c['n_embd']}, {c['n_head']} heads, {c['n_layer']} layers)"
    # the training loop proper: nothing reads parameters between steps, so the Blocks' optimizer pass may run beside the next
    # forward (GPT_OPT_OVERLAP=0 for A/B)
    ts = TrainStep(model, diff, lr=4e-4, wd=0.1, betas=(0.9, 0.999), grad_clip=clip, ema_decay=0.9999,
                   overlap_update=os.environ.get("GPT_OPT_OVERLAP", "1") != "0")
    batches = [synthetic_pairs(case, B, device, seed=1000 * rank + i) for i in range(4)]
    iters_per_epoch = 1000
    augment = getattr(args, "augment", False)
    if augment:
        from . import permute
        arch = (permute.MLP_ARCH if case.startswith("mnist") else permute.ACTOR_ARCH if case.startswith("cartpole")
                else permute.CNN_ARCH)

    store = None
    if getattr(args, "data", "synthetic") == "store":
        # the whole input pipeline in the timed step: pairs drawn from a checkpoint store resident in HBM (g.pt_amd/data.py)
        import tempfile
        import numpy
        from . import data as gdata
        name, metric = ("mnist_loss", "avg_test_loss") if mnist else ("cifar10_error", "avg_test_err")
        with tempfile.TemporaryDirectory() as root:
            gdata.write_synthetic_dataset(root, name, runs=64, ckpts=64, train_metric=metric, seed=rank, std=c_std(case))
            store = gdata.CheckpointStore(root, name, split="train", num_test_runs=0, train_metric=metric, device=device,
                                          normalizer_name="none", permute_augment=augment, target_epoch_size=1 << 20)
        order = torch.randperm(len(store), generator=torch.Generator().manual_seed(rank))
        rng = numpy.random.default_rng([1, rank])

    def get_batch(i):
        if store is not None:
            b = store.get_batch(order[(i * B) % (len(store) - B):][:B].tolist(), rng)
            return (b["parameters_0"], b["parameters_1"], b[metric + "_0"].to(device, non_blocking=True),
                    b[metric + "_1"].to(device, non_blocking=True))
        w_t, w_t1, l_t, l_t1 = batches[i % len(batches)]
        if augment:     # one function-preserving permutation per pair, on the GPU (Gpt/data/augment.py:25-75)
            seeds = range((i * world + rank) * B, (i * world + rank + 1) * B)
            w_t, w_t1 = permute.random_permute_batch([w_t, w_t1], arch, seeds)
        return w_t, w_t1, l_t, l_t1

    def run(n, first):
        last = None
        for i in range(first, first + n):
            ts.set_lr(lr_at(4e-4, 25000, 5, 0.0, i / iters_per_epoch))
            t, wts = sampler.sample(B, device)
            w_t, w_t1, l_t, l_t1 = get_batch(i)
            last = ts.step(w_t, w_t1, l_t, l_t1, t, weights=wts)
        return last

    # The first warm-up step is value-checked, whatever the batch size (the oracle comparisons of tests/test_hip_fullsize.py stop
    # at B = 128): its per-row mse (training forward with kept activations at M = 26 B rows) against
    # mean_flat((x_start - model(x_t, t))^2) (gaussian_diffusion.py:765-773) of its first <= 128 rows through the inference
    # forward (other kernel instances at M <= 3328), same weights, t and noise.
    first_check = None
    n_warm = args.warmup
    if args.warmup > 0:
        ts.set_lr(lr_at(4e-4, 25000, 5, 0.0, 0.0))
        t, wts = sampler.sample(B, device)
        w_t, w_t1, l_t, l_t1 = get_batch(0)
        noise = torch.empty_like(w_t1)
        L.check(L.lib().gpt_randn(L.ptr(noise), noise.numel(), 0x5EED, rank, L.stream()), "gpt_randn")
        n = min(B, 128)
        with torch.no_grad():
            x_t = diff.q_sample(w_t1[:n], t[:n], noise=noise[:n])
            out = model(x_t, ts.tmap[t[:n]], l_t1[:n], loss_prev=l_t[:n], x_prev=w_t[:n])
            target = w_t1[:n] if diff.model_mean_type == ModelMeanType.START_X else noise[:n]
            want = (target - out).square().mean(dim=1)
        got = ts.step(w_t, w_t1, l_t, l_t1, t, weights=wts, noise=noise)[:n]
        rel = float(((got - want).abs() / want.abs().clamp_min(1e-12)).max())
        bar = 2e-2 if args.dtype == "bf16" else 1e-3
        assert rel < bar, f"first training step: per-row mse differs from the inference forward's by {rel:.3e} (bar {bar})"
        first_check = {"rows": n, "mse_max_rel_err_vs_inference_forward": rel, "bar": bar, "loss": float(got.mean())}
        n_warm -= 1
    run(n_warm, 1)
    sync_all(world)
    t0 = time.perf_counter()
    mse = run(args.steps, args.warmup)
    sync_all(world)
    elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(elapsed, device)
    loss = float(mse.mean().item())
    assert math.isfinite(loss), "training diverged"
    ms = elapsed * 1e3 / args.steps
    flops = 3 * flops_per_forward(case) * B
    roofline = None
    if not getattr(args, "no_roofline", False):
        # the MFMA GEMM launches of the step (forward NT, dgrad NT incl. their K-span passes, weight-gradient TN), each bracketed
        # by a pair of HIP events on its own stream in two extra, untimed steps: algorithmic 2 M N K FLOPs / summed kernel time
        lib = L.lib()
        sync_all(world)
        L.check(lib.gpt_gemm_prof_enable(1), "prof")
        run(2, args.warmup + args.steps)
        torch.cuda.synchronize()
        pms, pfl, pnl = C.c_double(), C.c_double(), C.c_int()
        L.check(lib.gpt_gemm_prof_collect(C.byref(pms), C.byref(pfl), C.byref(pnl)), "prof")
        L.check(lib.gpt_gemm_prof_enable(0), "prof")
        if pnl.value and pms.value > 0:
            ach = pfl.value / (pms.value * 1e-3) / 1e12
            roofline = {"bound": "mfma", "achieved": ach, "peak": PEAK[args.dtype], "unit": "TFLOP/s", "frac": ach / PEAK[args.dtype],
                        "traffic": None, "kernel": "all MFMA GEMM launches of the training step (gemm256 / gemm128 / gemm_nt forward and "
                        "dgrad, gemm256_tn weight gradients), HIP events on the launch stream",
                        "launches_per_step": pnl.value / 2, "gemm_ms_per_step": pms.value / 2, "flops_per_step": pfl.value / 2}
    return {
        "metric": "train iters/sec (DDP)", "value": 1e3 / ms, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{cfg_name} step on synthetic checkpoint pairs"
                               + (" fetched per step from an HBM-resident checkpoint store (pair sampling + gather"
                                  " + normalisation inside the timed step)" if store is not None else "")
                               + ": q_sample + denoiser "
                               f"forward/backward + grad all-reduce + clip {clip} + AdamW(4e-4, wd 0.1) + EMA 0.9999; "
                               f"B={B} pairs per GPU (global {B * world})"
                               + ("; on-device permutation augmentation of every pair" if augment else ""),
                   "batch_per_gpu": B, "global_batch": B * world},
        "samples_per_s": B * world * 1e3 / ms, "model_tflops_per_gpu": flops / (ms * 1e-3) / 1e12,
        "final_loss": loss, "first_step_check": first_check, "roofline": roofline,
    }
