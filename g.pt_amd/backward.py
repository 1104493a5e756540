"""Training path of the denoiser: forward with kept activations + hand-written backward (gpt_denoiser_backward),
bridged to torch.autograd so `diffusion.training_losses(model, ...)["loss"].backward()` (main.py:47-67) works
unchanged, also under DistributedDataParallel.

Gradients are produced into ONE flat fp32 arena; each parameter's gradient is a view of it (the query / key /
value weight gradients of a layer are three consecutive slices of the fused [3d, d] gradient the kernels write).
"""
import ctypes as C
import os

import torch

from . import _lib as L


class GradArena:
    """Flat fp32 gradient buffer + the C descriptor that points the kernels at its slices.

    Two regions.  S ("sharded"): the Linear weights of the blocks, layer by layer (query | key | value back to back = the fused
    [3d, d] operand, proj, mlp.0, mlp.2), then the decoder Linears - 99 % of the elements, every tensor a multiple of 64
    elements (n_embd % 64 == 0), so any run of whole tensors splits evenly over 1 / 2 / 4 / 8 / 16 ranks: the region the
    data-parallel step reduce-scatters, updates 1/world of, and all-gathers as bf16 operands (trainstep.py).  R ("replicated"):
    biases, LayerNorm parameters, pos_emb and the ragged encoder Linears (K-padded operand layout) - all-reduced and updated on
    every rank."""

    def __init__(self, model, device):
        dec = model.decoder
        d, Lr = dec.n_embd, dec.n_layer
        self.slices = {}                      # parameter name -> (offset, numel)
        off = 0

        def take(name, numel):
            nonlocal off
            self.slices[name] = (off, numel)
            off += (numel + 7) // 8 * 8       # every slice 32-byte aligned (16 bytes in a bf16 copy of the arena)
            return self.slices[name][0]

        self.layer_off = [dict() for _ in range(Lr)]
        # ---- region S
        self.s_layer_start = []
        for l in range(Lr):
            p = f"decoder.blocks.{l}."
            lo = self.layer_off[l]
            self.s_layer_start.append(off)
            # fused [3d, d] weight gradient: query | key | value back to back (no padding between them)
            lo["g_wqkv"] = off
            for nm in ("query", "key", "value"):
                self.slices[p + f"attn.{nm}.weight"] = (off, d * d)
                off += d * d
            off = (off + 7) // 8 * 8
            lo["g_wproj"] = take(p + "attn.proj.weight", d * d)
            lo["g_w1"] = take(p + "mlp.0.weight", 4 * d * d)
            lo["g_w2"] = take(p + "mlp.2.weight", 4 * d * d)
        self.s_dec_start = off
        self.dec_off = [take(f"decoder.output_parameter_projections.{j}.0.weight", n * d)
                        for j, n in enumerate(dec.output_splits)]
        self.s_end = self.r_start = off
        # ---- region R
        for l in range(Lr):
            p = f"decoder.blocks.{l}."
            lo = self.layer_off[l]
            lo["g_bqkv"] = off                # [3d] bias gradient: query | key | value back to back
            for nm in ("query", "key", "value"):
                self.slices[p + f"attn.{nm}.bias"] = (off, d)
                off += d
            off = (off + 7) // 8 * 8
            lo["g_bproj"] = take(p + "attn.proj.bias", d)
            lo["g_b1"] = take(p + "mlp.0.bias", 4 * d)
            lo["g_b2"] = take(p + "mlp.2.bias", d)
            for ln in ("ln1", "ln2"):
                lo[f"g_{ln}_w"] = take(p + ln + ".weight", d)
                lo[f"g_{ln}_b"] = take(p + ln + ".bias", d)
        self.glob = {"g_pos_emb": take("decoder.pos_emb", dec.block_size * d)}
        for ln, key in (("ln_in", "g_ln_in"), ("ln_f", "g_ln_f")):
            self.glob[key + "_w"] = take(f"decoder.{ln}.weight", d)
            self.glob[key + "_b"] = take(f"decoder.{ln}.bias", d)
        self.enc_off = [take(f"decoder.input_parameter_projections.{i}.0.weight", d * k)
                        for i, k in enumerate(dec.input_splits)]
        self.numel = off
        self.device = device
        self.names = [n for n, _ in model.named_parameters()]
        assert set(self.names) == set(self.slices), set(self.names) ^ set(self.slices)

    def new_buffer(self):
        return torch.zeros(self.numel, dtype=torch.float32, device=self.device)

    def views(self, buf, model):
        out = []
        for name, p in model.named_parameters():
            o, n = self.slices[name]
            out.append(buf[o:o + n].view_as(p))
        return out


def _ensure_training_state(eng):
    """Transposed packed weights (dgrad operands) + backward scratch, refreshed with the packed weights."""
    m, dec = eng.model, eng.model.decoder
    if getattr(eng, "_t_sig", None) == eng.sig and getattr(eng, "arena", None) is not None:
        return
    lib = L.lib()
    if getattr(eng, "arena", None) is None:
        eng.arena = GradArena(m, eng.device)
    have = getattr(eng, "wt", None)
    wt = []
    for l in range(dec.n_layer):
        ly = eng.cmodel.layers[l]
        d = dec.n_embd
        ts = {}
        for key, ptr, rows, cols in (("wqkv_t", ly.wqkv, 3 * d, d), ("wproj_t", ly.wproj, d, d), ("w1_t", ly.w1, 4 * d, d),
                                     ("w2_t", ly.w2, d, 4 * d)):
            t = have[l][key] if have else torch.empty(cols, rows, dtype=eng.tdtype, device=eng.device)
            L.check(lib.gpt_transpose(eng.code, ptr, L.ptr(t), rows, cols, rows, L.stream()), "gpt_transpose")
            ts[key] = t
        wt.append(ts)
    eng.wt = wt
    eng._t_sig = eng.sig


def _scratch(eng, B, slots=1):
    """Scratch of the backward; `slots` > 1: the dY operands of deferred weight-gradient launches (gpt_backward_t.wgrad_defer)."""
    have = getattr(eng, "_bw_scratch", None)
    if have is not None and have[0] >= B and have[2] >= slots:
        return have[1]
    dec = eng.model.decoder
    d, T = dec.n_embd, dec.block_size
    M = B * T
    Mp = (M + 63) // 64 * 64
    dev, td = eng.device, eng.tdtype
    s = dict(g=torch.empty(M, d, dtype=torch.float32, device=dev), gt=torch.empty(slots * M, d, dtype=td, device=dev),
             d_h=torch.empty(M, d, dtype=td, device=dev), d_qkv=torch.empty(slots * M, 3 * d, dtype=td, device=dev),
             d_act=torch.empty(slots * M, 4 * d, dtype=td, device=dev), tA=torch.empty(4 * d, Mp, dtype=td, device=dev),
             tB=torch.empty(4 * d, Mp, dtype=td, device=dev), d_tok=torch.empty(M, d, dtype=torch.float32, device=dev),
             # per-workgroup dgamma / dbeta / bias sums of every LayerNorm backward launch of a call (gpt_backward_t.ln_partials)
             ln_partials=torch.empty((2 * dec.n_layer + 2) * 256 * 3 * d, dtype=torch.float32, device=dev))
    eng._bw_scratch = (B, s, slots)
    return s


def _grouped_wgrads(eng, B):
    """True where gpt_denoiser_backward computes the Blocks' weight gradients by the grouped TN launch."""
    dec = eng.model.decoder
    return eng.code == L.BF16 and dec.n_embd % 256 == 0 and \
        os.environ.get("GPT_WGRAD_TN", "1") != "0"


def wgrad_defer_layers(eng, B):
    """Blocks whose grouped weight-gradient launches are worth merging (gpt_backward_t.wgrad_defer): the smallest G <= 4 that
    minimises whole rounds of CUs per Block, ceil(G * tiles / CUs) / G, with tiles = 12 (n_embd / 256)^2 per Block (n_embd 1536:
    432 tiles -> G = 4: 7 rounds per four Blocks instead of 8; n_embd 2048: 768 tiles = 3 rounds exactly -> 1).
    GPT_WGRAD_DEFER=<G> forces a value (A/B measurements)."""
    if not _grouped_wgrads(eng, B):
        return 1
    forced = os.environ.get("GPT_WGRAD_DEFER")
    if forced is not None:
        return max(1, min(4, int(forced)))
    tiles = 12 * (eng.model.decoder.n_embd // 256) ** 2
    cus = torch.cuda.get_device_properties(eng.device).multi_processor_count
    best, best_cost = 1, None
    for g in range(1, 5):
        cost = -(-g * tiles // cus) / g
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = g, cost
    return best


def can_overwrite_wgrads(eng, B):
    """True where gpt_denoiser_backward computes the Blocks' weight gradients by the grouped TN launch, which can store them
    instead of accumulating (gpt_backward_t.wgrad_overwrite)."""
    return _grouped_wgrads(eng, B) and os.environ.get("GPT_WGRAD_OVERWRITE", "1") != "0"


def _descriptor(eng, buf, scratch, overwrite=False, defer=1, sumsq=None):
    dec = eng.model.decoder
    ar = eng.arena
    base = buf.data_ptr()
    layers = (L.LayerGrad * dec.n_layer)()
    for l in range(dec.n_layer):
        lg = layers[l]
        for k, t in eng.wt[l].items():
            setattr(lg, k, L.ptr(t))
        for k, o in ar.layer_off[l].items():
            setattr(lg, k, base + 4 * o)
    bw = L.Backward()
    bw.layers = C.cast(layers, C.POINTER(L.LayerGrad))
    for k, o in ar.glob.items():
        setattr(bw, k, base + 4 * o)
    for i, o in enumerate(ar.enc_off):
        bw.g_enc_w[i] = base + 4 * o
    for j, o in enumerate(ar.dec_off):
        bw.g_dec_w[j] = base + 4 * o
    for k, t in scratch.items():
        setattr(bw, k, L.ptr(t))
    bw.ln_partials_bytes = scratch["ln_partials"].numel() * 4
    bw.wgrad_overwrite = int(overwrite)
    bw.wgrad_defer = int(defer)
    bw.wgrad_sumsq = L.ptr(sumsq) if sumsq is not None else None
    return bw, layers


def forward_keep(model, x, t, lt, lp, x_prev):
    """Training forward: keeps per-layer activations and LayerNorm statistics in the engine's workspace."""
    eng = model.engine(x.device)
    _ensure_training_state(eng)
    B = x.shape[0]
    ws = eng.workspace(B, keep_activations=True)
    eng.keep_gen += 1
    out = torch.empty_like(x)
    L.check(L.lib().gpt_denoiser_forward(C.byref(eng.cmodel), C.byref(ws), L.ptr(x), L.ptr(t), L.ptr(lt), L.ptr(lp),
                                         int(lt.dtype == torch.float64), L.ptr(x_prev), L.ptr(out), B, L.stream()),
            "gpt_denoiser_forward")
    return eng, ws, out


def backward_into(eng, ws, x, x_prev, d_out, buf, schedule=None, overwrite=False, sumsq=None):
    """Runs the backward of the last forward_keep on this engine, accumulating into the flat buffer `buf`.

    overwrite: the Blocks' weight gradients are stored, not accumulated (their part of `buf` need not be zero; only where
    can_overwrite_wgrads holds).
    schedule: optional [(stage_end, callback)] with increasing stage_end (stages as in gpt_denoiser_backward_stages):
    each callback is called on the host right after the stages before its stage_end have been enqueued - the hook the
    training step uses to put the gradient all-reduce of finished layers in flight behind them."""
    B = x.shape[0]
    # without a schedule (nothing waits for the gradients of single Blocks) the grouped weight-gradient launches of several
    # Blocks are merged; with one, every stage range ends with its own launch (the C side merges inside one call only)
    defer = wgrad_defer_layers(eng, B) if not schedule else 1
    scratch = _scratch(eng, B, defer)
    # sumsq (a zeroed fp32 device scalar, store mode only): += the sum of squares of the Blocks' weight gradients
    bw, keep = _descriptor(eng, buf, scratch, overwrite, defer, sumsq)
    n_stages = eng.model.decoder.n_layer + 2
    done = 0
    for stage_end, callback in list(schedule or []) + [(n_stages, None)]:
        if stage_end > done:
            L.check(L.lib().gpt_denoiser_backward_stages(C.byref(eng.cmodel), C.byref(ws), C.byref(bw), L.ptr(x), L.ptr(x_prev),
                                                         L.ptr(d_out), B, done, stage_end, L.stream()),
                    "gpt_denoiser_backward_stages")
            done = stage_end
        if callback is not None:
            callback()
    del keep


def stage_parameter_names(model):
    """Parameter names by backward stage (gpt_denoiser_backward_stages): 0 = decoder Linears + ln_f; s in 1..L = block L - s;
    L + 1 = pos_emb, ln_in, encoder Linears.  (The bias gradient of a block's mlp.2 is produced one stage EARLIER, by the
    LayerNorm backward that yields its output gradient, so it is complete when its own stage returns it.)"""
    dec = model.decoder
    Lr = dec.n_layer
    stages = [[] for _ in range(Lr + 2)]
    for name, _ in model.named_parameters():
        if name.startswith("decoder.blocks."):
            stages[Lr - int(name.split(".")[2])].append(name)
        elif name.startswith("decoder.output_parameter_projections.") or name.startswith("decoder.ln_f."):
            stages[0].append(name)
        else:
            stages[Lr + 1].append(name)
    return stages


class _PendingBackward:
    """What the stage nodes of ONE training forward share: engine, kept-activation workspace, inputs, the gradient arena."""

    def __init__(self, model, eng, ws, x, x_prev):
        self.model, self.eng, self.ws, self.x, self.x_prev = model, eng, ws, x, x_prev
        # the kept activations live in the engine's ONE training workspace: a second grad-enabled forward of the same model
        # before this backward overwrites them (or re-allocates them for a larger batch).  The generation counter turns that
        # into an error instead of silently wrong gradients; the buffers stay referenced here so that a re-allocation cannot
        # free memory this backward's descriptor still points at.
        self.keep_gen, self.bufs = eng.keep_gen, eng.buffers(keep_activations=True)
        self.versions = tuple(p._version for p in model.parameters())
        self.buf = self.bw = self.keep = self.d_out = None
        self.done = 0

    def begin(self, d_out):
        eng = self.eng
        if eng.keep_gen != self.keep_gen:
            raise RuntimeError("Gpt: another grad-enabled forward of this model ran before this backward; the HIP path keeps the "
                               "activations of ONE pending forward per model (call backward() after each forward, as "
                               "main.py:47-67 does)")
        if self.versions != tuple(p._version for p in self.model.parameters()):
            raise RuntimeError("G.pt parameters changed between forward and backward")
        self.d_out = d_out.detach().float().contiguous()
        self.buf = _gradient_buffer(eng, self.model)
        self.bw, self.keep = _descriptor(eng, self.buf, _scratch(eng, self.x.shape[0]))

    def run_stage(self, stage):
        assert stage == self.done, "backward stages run in order"
        L.check(L.lib().gpt_denoiser_backward_stages(C.byref(self.eng.cmodel), C.byref(self.ws), C.byref(self.bw), L.ptr(self.x),
                                                     L.ptr(self.x_prev), L.ptr(self.d_out), self.x.shape[0], stage, stage + 1,
                                                     L.stream()), "gpt_denoiser_backward_stages")
        self.done = stage + 1


def _gradient_buffer(eng, model):
    """A zeroed gradient arena.  The previous call's arena is reused (one memset instead of a 1.5 GB allocation + fill per
    iteration) unless some parameter's .grad still aliases it (autograd may adopt a returned gradient as .grad: a caller who
    accumulates over several backward calls must keep what it holds)."""
    buf = getattr(eng, "_grad_buf", None)
    if buf is not None:
        lo, hi = buf.data_ptr(), buf.data_ptr() + buf.numel() * 4
        if any(p.grad is not None and lo <= p.grad.data_ptr() < hi for p in model.parameters()):
            buf = None
    if buf is None:
        buf = eng._grad_buf = eng.arena.new_buffer()
    else:
        buf.zero_()
    return buf


class _StageFunction(torch.autograd.Function):
    """One backward stage as its own autograd node.  The nodes of a forward form a chain (the output of stage s + 1 is a
    one-element carrier fed to stage s), so autograd runs them in stage order and hands every stage's parameter gradients to
    their AccumulateGrad nodes - and DistributedDataParallel's bucket hooks on them (main.py:229-240) - as soon as that stage
    has been ENQUEUED: the reducer's all-reduces overlap with the remaining stages, as they do for the reference's per-op graph.
    The forward itself runs once, as a whole, in the deepest node."""

    @staticmethod
    def forward(ctx, pending, stage, n_stages, names, carrier, *params):
        ctx.pending, ctx.stage, ctx.names = pending, stage, names
        ctx.has_carrier = carrier is not None
        if stage == 0:
            return pending.out
        return pending.x.new_zeros(1)

    @staticmethod
    def backward(ctx, g):
        pd = ctx.pending
        if ctx.stage == 0:
            pd.begin(g)
        pd.run_stage(ctx.stage)
        ar = pd.eng.arena
        named = dict(pd.model.named_parameters())
        grads = []
        for n in ctx.names:
            o, cnt = ar.slices[n]
            grads.append(pd.buf[o:o + cnt].view_as(named[n]))
        carrier_grad = g.new_zeros(1) if ctx.has_carrier else None
        return (None, None, None, None, carrier_grad, *grads)


class DenoiseFunction:
    """autograd bridge of one denoiser call: n_layer + 2 chained stage nodes (see _StageFunction).  The parameters are explicit
    inputs of their stage's node so that their AccumulateGrad nodes (and DDP's hooks on them) receive the gradients."""

    @staticmethod
    def run(model, x, t, lt, lp, x_prev):
        eng, ws, out = forward_keep(model, x, t, lt, lp, x_prev)
        pending = _PendingBackward(model, eng, ws, x, x_prev)
        pending.out = out
        stages = getattr(model, "_stage_names", None)
        if stages is None:
            stages = model._stage_names = stage_parameter_names(model)
        named = dict(model.named_parameters())
        if not all(p.requires_grad for p in named.values()):
            raise NotImplementedError("the HIP training path differentiates all G.pt parameters together (freeze none, or run "
                                      "under torch.no_grad())")
        n = len(stages)
        carrier = None
        for stage in range(n - 1, -1, -1):
            carrier = _StageFunction.apply(pending, stage, n, stages[stage], carrier, *[named[k] for k in stages[stage]])
        return carrier
