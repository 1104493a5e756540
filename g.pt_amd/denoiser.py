"""Host side of the G.pt denoiser on MI355X.

`Gpt` keeps the reference's constructor, `forward(x, t, loss_target, loss_prev, x_prev)` signature and
`state_dict` layout (Gpt/models/transformer.py:337-464; checkpoint ABI in SURVEY.md section 8b), so
checkpoints, EMA copies, DDP wrapping and `main.py` / `playground.py` work unchanged.  The torch modules
below are parameter containers only: the arithmetic runs in the hand-written gfx950 kernels of
g.pt_amd/csrc through the C ABI of include/gpt_hip.h.  There is no PyTorch or CPU fallback.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib as L


# ------------------------------------------------------------------------------------------ token plan
def flatten_parameter_sizes(parameter_sizes):
    """Reference: GPT.flatten_parameter_sizes (transformer.py:244-249)."""
    assert len(parameter_sizes) > 0
    if isinstance(parameter_sizes[0], (list, tuple)):
        return [n for group in parameter_sizes for n in group]
    return parameter_sizes


def _chunks(total, chunk):
    return [chunk] * (total // chunk) + ([total % chunk] if total % chunk else [])


def build_splits(parameter_sizes, split_policy="chunk", chunk_size=None):
    """Token widths for the four tokenisation policies (transformer.py:251-300)."""
    if split_policy == "chunk":
        assert chunk_size is not None
        return _chunks(sum(flatten_parameter_sizes(parameter_sizes)), chunk_size)
    if split_policy == "layer_by_layer":
        return flatten_parameter_sizes(parameter_sizes)
    if split_policy == "chunk_within_layer":
        assert chunk_size is not None
        return [w for n in flatten_parameter_sizes(parameter_sizes) for w in _chunks(n, chunk_size)]
    if split_policy == "chunk_within_input":
        out = []
        for group in parameter_sizes:
            assert isinstance(group, (list, tuple))
            out.extend(build_splits(group, "chunk", chunk_size))
        return out
    raise NotImplementedError


# ------------------------------------------------------------------------------------------ parameter containers
class SelfAttention(nn.Module):
    """Parameters of transformer.py:12-51 (separate key/query/value/proj Linears: checkpoint ABI)."""

    def __init__(self, n_embd, n_head, attn_pdrop=0.0, resid_pdrop=0.0):
        super().__init__()
        assert n_embd % n_head == 0
        self.key = nn.Linear(n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(n_embd, n_embd)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        self.proj = nn.Linear(n_embd, n_embd)
        self.n_head = n_head


class Block(nn.Module):
    """Parameters of transformer.py:54-72."""

    def __init__(self, n_embd, n_head, attn_pdrop=0.0, resid_pdrop=0.0):
        super().__init__()
        self.ln1 = nn.LayerNorm(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = SelfAttention(n_embd, n_head, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(nn.Linear(n_embd, 4 * n_embd), nn.GELU(), nn.Linear(4 * n_embd, n_embd),
                                 nn.Dropout(resid_pdrop))


class FrequencyEmbedder(nn.Module):
    """Holds the persistent fp32 `frequencies` buffer (transformer.py:317-322); the embedding itself is
    computed by gpt_scalar_tokens inside the denoiser forward."""

    def __init__(self, num_frequencies, max_freq_log2):
        super().__init__()
        self.register_buffer("frequencies", 2 ** torch.linspace(0, max_freq_log2, steps=num_frequencies))

    def forward(self, x):
        """Stand-alone embedding of (N,) or (N,1) scalars on the GPU: [sin(f s), cos(f s), s]."""
        s = x.reshape(-1)
        if not s.is_cuda:
            raise L.GptHipError("FrequencyEmbedder: the HIP path needs GPU tensors")
        n = s.shape[0]
        nf = self.frequencies.numel()
        zeros = torch.zeros(n, dtype=torch.float32, device=s.device)
        if s.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8):
            t, a = s.long().contiguous(), zeros
        else:
            t = torch.zeros(n, dtype=torch.int64, device=s.device)
            a = s.contiguous() if s.dtype in (torch.float32, torch.float64) else s.float().contiguous()
        is64 = a.dtype == torch.float64
        b = torch.zeros_like(a)
        out = torch.empty(n, 4 * (2 * nf + 1), dtype=torch.float32, device=s.device)
        L.check(L.lib().gpt_scalar_tokens(L.ptr(t), L.ptr(a), L.ptr(b), int(is64), L.ptr(self.frequencies), nf, 4,
                                          L.ptr(out), n, L.stream()), "gpt_scalar_tokens")
        W = 2 * nf + 1
        return out[:, :W] if s.dtype == torch.int64 or not s.is_floating_point() else out[:, 2 * W:3 * W]


class GPT(nn.Module):
    """Parameters of the transformer trunk (transformer.py:75-121) in the reference's module tree."""

    def __init__(self, input_parameter_sizes, output_parameter_sizes, input_parameter_names, n_layer=12, n_head=12,
                 n_embd=768, encoder_depth=1, decoder_depth=1, attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0,
                 chunk_size=None, split_policy="chunk"):
        super().__init__()
        self.input_splits = build_splits(input_parameter_sizes, split_policy, chunk_size)
        self.output_splits = build_splits(output_parameter_sizes, split_policy, chunk_size)
        block_size = len(self.input_splits)
        if split_policy == "layer_by_layer":
            assert len(input_parameter_names) == block_size
        if encoder_depth != 1 or decoder_depth != 1:
            raise NotImplementedError("the HIP path implements encoder_depth = decoder_depth = 1 "
                                      "(every shipped G.pt config, configs/train/*.yaml:41-42)")
        self.pos_emb = nn.Parameter(torch.zeros(1, block_size, n_embd))
        self.drop = nn.Dropout(embd_pdrop)
        self.blocks = nn.Sequential(*[Block(n_embd, n_head, attn_pdrop, resid_pdrop) for _ in range(n_layer)])
        self.block_size = block_size
        self.input_parameter_projections = nn.ModuleList(
            nn.Sequential(nn.Linear(k, n_embd, bias=False)) for k in self.input_splits)
        self.ln_in = nn.LayerNorm(n_embd)
        self.ln_f = nn.LayerNorm(n_embd)
        self.output_parameter_projections = nn.ModuleList(
            nn.Sequential(nn.Linear(n_embd, n, bias=False)) for n in self.output_splits)
        self.num_output_heads = len(self.output_splits)
        self.n_embd, self.n_head, self.n_layer = n_embd, n_head, n_layer
        self.dropout_p = max(attn_pdrop, resid_pdrop, embd_pdrop)
        self.apply(self._init_weights)

    build_splits = staticmethod(build_splits)
    flatten_parameter_sizes = staticmethod(flatten_parameter_sizes)

    def get_block_size(self):
        return self.block_size

    @staticmethod
    def _init_weights(module):
        """N(0, 0.02) Linear weights, zero biases, LayerNorm (1, 0) (transformer.py:153-161)."""
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    @staticmethod
    def configure_optimizers(nn_module, lr, wd, betas):
        """AdamW with two groups (transformer.py:163-213): Linear weights decay; biases, LayerNorm
        weights and pos_emb do not.  Parameters are ordered by name inside each group."""
        decay, no_decay = set(), set()
        for mod_name, mod in nn_module.named_modules():
            for p_name, _ in mod.named_parameters(recurse=False):
                full = f"{mod_name}.{p_name}" if mod_name else p_name
                if p_name.endswith("bias"):
                    no_decay.add(full)
                elif p_name.endswith("weight") and isinstance(mod, (nn.Linear, nn.Conv1d)):
                    decay.add(full)
                elif p_name.endswith("weight") and isinstance(mod, (nn.LayerNorm, nn.Embedding, FrequencyEmbedder)):
                    no_decay.add(full)
                elif p_name.endswith("pos_emb"):
                    no_decay.add(full)
        params = dict(nn_module.named_parameters())
        assert not (decay & no_decay), f"parameters in both groups: {decay & no_decay}"
        assert not (params.keys() - (decay | no_decay)), f"unassigned parameters: {params.keys() - (decay | no_decay)}"
        groups = [{"params": [params[n] for n in sorted(decay)], "weight_decay": wd},
                  {"params": [params[n] for n in sorted(no_decay)], "weight_decay": 0.0}]
        return torch.optim.AdamW(groups, lr=lr, betas=betas)

    def forward(self, x):
        raise NotImplementedError("the HIP path fuses the concatenation of transformer.py:458-459 away: call "
                                  "Gpt.forward(x, t, loss_target, loss_prev, x_prev)")


# ------------------------------------------------------------------------------------------ device engine
def _round_up(n, m):
    return (n + m - 1) // m * m


class _Engine:
    """Packed device weights + activation workspace of one (model, device, compute dtype).

    Packed copies (fused [query|key|value] weight and bias, bf16 casts, K-padded encoder weights) live in buffers allocated
    once per parameter layout and are RE-DERIVED from the live parameters on every `Gpt.engine()` call - one launch
    (gpt_pack_tensors, or gpt_pack_segments for a flat arena), i.e. once per forward / per sampling loop.  torch's version
    counters cannot be trusted for this: the reference's EMA update writes through `.data` (Gpt/utils.py:146-152), which
    does not bump them.  Only an owner that rewrites the packed operands itself together with the parameters (the fused
    optimizer pass of TrainStep) sets `managed` and calls `touched()` instead."""

    def __init__(self, model, device, code):
        self.model, self.device, self.code = model, device, code
        self.tdtype = L.torch_dtype(code)
        self.struct = None          # parameter layout the pointer tables were built for
        self.gen = 0                # bumped whenever the packed operands were re-derived (or rewritten by the owner)
        self.managed = False
        self.owner_sync = None      # managed engines: called before the operands are re-derived from the fp32 parameters
        self._versions = None
        self.keep_gen = 0           # bumped by every training forward (backward.py: one pending backward per engine)
        self.keep = []
        self.cmodel = None
        self.items = None
        self.ws = {}
        L.ensure_gemm_workspace(device)

    # ---- weights
    @property
    def sig(self):
        return (self.struct, self.gen)

    def touched(self):
        """The owner rewrote parameters and packed operands together (TrainStep.update)."""
        self.gen += 1

    def _structure(self, flat):
        return (id(flat) if flat is not None else 0,) + tuple(p.data_ptr() for p in self.model.parameters())

    def refresh(self):
        m = self.model
        flat = getattr(m, "_flat", None)
        if flat is not None and not flat.bound(m):
            flat = m._flat = None
        st = self._structure(flat)
        if st != self.struct:
            self._build(flat)
            self.struct = st
        elif self.managed and self._versions == tuple(p._version for p in m.parameters()):
            return                  # (a load_state_dict / optimizer.step behind the owner's back still shows in the counters)
        if self.managed and self.owner_sync is not None:
            self.owner_sync()       # sharded / overlapped owner: whole fp32 parameters before anything is derived from them
        self._repack(flat)

    def _repack(self, flat):
        if flat is not None:
            flat.repack(self.code)
        else:
            # skipped on the device when no parameter changed since the last pack (a 64-bit checksum pass over the sources: a
            # third of the pack's bytes, no host sync) - a host loop over model() calls, as the reference's
            # p_sample_loop_progressive, pays 0.3 instead of 1.0 ms per call at the mnist shapes for the .data-write safety
            table, n_items, n4 = self.items
            L.check(L.lib().gpt_pack_tensors_if_changed(L.ptr(table), n_items, n4, L.ptr(self._pack_state), L.stream()),
                    "gpt_pack_tensors_if_changed")
        self._versions = tuple(p._version for p in self.model.parameters())
        self.gen += 1

    def _build(self, flat):
        m, dec = self.model, self.model.decoder
        for p in m.parameters():
            if p.device != self.device or p.dtype != torch.float32:
                raise L.GptHipError("all G.pt parameters must be fp32 tensors on one GPU")
            if not p.is_contiguous() or p.data_ptr() % 16:
                raise L.GptHipError("G.pt parameters must be contiguous, 16-byte aligned tensors")
        keep, items = [], []
        dev, td, bf = self.device, self.tdtype, self.code == L.BF16

        def packed(src, dst=None, k=None, ld=None, dtype=None):
            """A packed copy of parameter `src` (re-derived by _repack); returns the destination tensor."""
            if dst is None:
                dst = torch.empty(src.shape, dtype=dtype or td, device=dev)
                keep.append(dst)
            items.append((src, dst, k or 1, ld or 1))
            return dst

        def operand(w):
            """GEMM operand of a Linear weight: the live fp32 parameter in fp32 mode, a bf16 copy in bf16 mode."""
            return L.ptr(packed(w)) if bf else L.ptr(w)

        d = dec.n_embd
        layers = (L.Layer * dec.n_layer)()
        for l, blk in enumerate(dec.blocks):
            a = blk.attn
            ly = layers[l]
            ly.ln1_w, ly.ln1_b = L.ptr(blk.ln1.weight), L.ptr(blk.ln1.bias)
            ly.ln2_w, ly.ln2_b = L.ptr(blk.ln2.weight), L.ptr(blk.ln2.bias)
            if flat is not None:
                pre = f"decoder.blocks.{l}."
                ly.wqkv, ly.bqkv = flat.weight_ptr(self.code, pre + "attn.query.weight"), flat.param_ptr(pre + "attn.query.bias")
                ly.wproj = flat.weight_ptr(self.code, pre + "attn.proj.weight")
                ly.w1, ly.w2 = flat.weight_ptr(self.code, pre + "mlp.0.weight"), flat.weight_ptr(self.code, pre + "mlp.2.weight")
            else:
                wqkv = torch.empty(3 * d, d, dtype=td, device=dev)
                bqkv = torch.empty(3 * d, dtype=torch.float32, device=dev)
                keep += [wqkv, bqkv]
                for i, lin in enumerate((a.query, a.key, a.value)):
                    packed(lin.weight, wqkv[i * d:(i + 1) * d])
                    packed(lin.bias, bqkv[i * d:(i + 1) * d])
                ly.wqkv, ly.bqkv = L.ptr(wqkv), L.ptr(bqkv)
                ly.wproj, ly.w1, ly.w2 = operand(a.proj.weight), operand(blk.mlp[0].weight), operand(blk.mlp[2].weight)
            ly.bproj, ly.b1, ly.b2 = L.ptr(a.proj.bias), L.ptr(blk.mlp[0].bias), L.ptr(blk.mlp[2].bias)
        cm = L.Model()
        cm.dtype, cm.d, cm.n_head, cm.n_layer = self.code, dec.n_embd, dec.n_head, dec.n_layer
        cm.T, cm.D = dec.block_size, m.D
        cm.nfreq, cm.n_scalar = m.scalar_embedder.frequencies.numel(), m.n_scalar
        cm.n_in_tok, cm.n_out_tok = dec.block_size, dec.num_output_heads
        cm.freqs, cm.pos_emb = L.ptr(m.scalar_embedder.frequencies), L.ptr(dec.pos_emb)
        cm.ln_in_w, cm.ln_in_b = L.ptr(dec.ln_in.weight), L.ptr(dec.ln_in.bias)
        cm.ln_f_w, cm.ln_f_b = L.ptr(dec.ln_f.weight), L.ptr(dec.ln_f.bias)
        for i, (kind, off, k) in enumerate(m.token_sources):
            kp = _round_up(k, 32)
            if flat is not None:
                wptr = flat.enc_ptr(self.code, i)
            else:
                w = dec.input_parameter_projections[i][0].weight
                wp = torch.zeros(w.shape[0], kp, dtype=td, device=dev)          # K padding stays zero: only [:, :k] is rewritten
                keep.append(wp)
                packed(w, wp, k, kp)
                wptr = L.ptr(wp)
            cm.enc_w[i], cm.enc_w_ld[i], cm.enc_k[i], cm.enc_src[i], cm.enc_off[i] = wptr, kp, k, kind, off
        off = 0
        for j, n in enumerate(dec.output_splits):
            if flat is not None:
                wptr = flat.weight_ptr(self.code, f"decoder.output_parameter_projections.{j}.0.weight")
            else:
                wptr = operand(dec.output_parameter_projections[j][0].weight)
            cm.dec_w[j], cm.dec_n[j], cm.dec_off[j] = wptr, n, off
            off += n
        cm.layers = C.cast(layers, C.POINTER(L.Layer))
        keep.append(layers)
        self.items = None
        if flat is None:
            arr = (L.PackItem * len(items))()
            first4 = 0
            for it, (src, dst, k, ld) in zip(arr, items):
                it.src, it.dst, it.first4, it.count = L.ptr(src), L.ptr(dst), first4, src.numel()
                it.k, it.ld, it.dst_dtype = k, ld, L.BF16 if dst.dtype == torch.bfloat16 else L.F32
                assert it.dst % 16 == 0
                first4 += (src.numel() + 3) // 4
            import numpy as np
            table = torch.from_numpy(np.frombuffer(bytes(arr), dtype=np.uint8).copy()).to(dev)
            self.items = (table, len(items), first4)
            self._pack_state = torch.zeros(4, dtype=torch.int64, device=dev)     # new destinations: the next pack must run
        ready = getattr(self, "_ready", None)
        self.keep, self.cmodel, self.clayers = keep, cm, layers
        self._ready = None
        if ready is not None:                   # a rebuild of the pointer tables must not drop waits that are still owed
            self.set_ready(*ready)

    def set_ready(self, layer_events, dec_event):
        """Operands delivered asynchronously (TrainStep's all-gather of the other ranks' shards): the forward waits for
        layer_events[l] (torch.cuda.Event or None) before block l's first kernel and for dec_event before the decoder Linears
        (gpt_layer_t.ready / gpt_model_t.dec_ready).  The events are kept alive here until they are replaced."""
        assert len(layer_events) == self.model.decoder.n_layer
        for ly, ev in zip(self.clayers, layer_events):
            ly.ready = ev.cuda_event if ev is not None else None
        self.cmodel.dec_ready = dec_event.cuda_event if dec_event is not None else None
        self._ready = (list(layer_events), dec_event)

    # ---- activations
    def workspace(self, B, keep_activations=False):
        key = bool(keep_activations)
        have = self.ws.get(key)
        if have is not None and have[0].B_max >= B:
            return have[0]
        m, dec = self.model, self.model.decoder
        d, T, Lr = dec.n_embd, dec.block_size, dec.n_layer
        M = B * T
        n = Lr if keep_activations else 1
        dev, td = self.device, self.tdtype
        bufs = dict(
            scal=torch.empty(B, m.n_scalar * (2 * m.scalar_embedder.frequencies.numel() + 1), dtype=torch.float32, device=dev),
            enc=torch.empty(M, d, dtype=torch.float32, device=dev),
            x=torch.empty((2 * Lr + 1) if keep_activations else 1, M, d, dtype=torch.float32, device=dev),
            h1=torch.empty(n, M, d, dtype=td, device=dev), qkv=torch.empty(n, M, 3 * d, dtype=td, device=dev),
            att=torch.empty(n, M, d, dtype=td, device=dev), h2=torch.empty(n, M, d, dtype=td, device=dev),
            act=torch.empty(n, M, 4 * d, dtype=td, device=dev), hf=torch.empty(M, d, dtype=td, device=dev))
        if keep_activations:
            bufs["pre"] = torch.empty(n, M, 4 * d, dtype=td, device=dev)
            bufs["stats"] = torch.empty(2 * Lr + 2, 2, M, dtype=torch.float32, device=dev)
        ws = L.Workspace()
        ws.B_max, ws.layer_stride = B, int(keep_activations)
        for k, v in bufs.items():
            setattr(ws, k, L.ptr(v))
        self.ws[key] = (ws, bufs)
        return ws

    def buffers(self, keep_activations=False):
        return self.ws[bool(keep_activations)][1]


def _as_loss(v, B, device):
    """(B,) or (B,1) loss scalars, fp32 or fp64 (DataLoader collates Python floats to float64)."""
    if not torch.is_tensor(v):
        v = torch.as_tensor(v, device=device)
    v = v.reshape(-1)
    if v.numel() == 1 and B > 1:
        v = v.expand(B)
    assert v.numel() == B, f"expected {B} loss scalars, got {v.numel()}"
    if v.dtype not in (torch.float32, torch.float64):
        v = v.float()
    return v.contiguous()


class Gpt(nn.Module):
    """The G.pt model (reference: transformer.py:337-464) on hand-written gfx950 kernels.

    compute dtype: "fp32" (default; exact-fp32 MFMA, the parity mode) or "bf16" (bf16 MFMA operands,
    fp32 accumulate / residual stream / statistics).  Select with `set_compute_dtype`, the
    GPT_AMD_DTYPE environment variable, or by calling under `torch.autocast('cuda')` (main.py:41, cfg.amp)."""

    def __init__(self, parameter_sizes, parameter_names, num_frequencies=128, max_freq_log2=20, predict_xstart=True,
                 absolute_loss_conditioning=False, **gpt_kwargs):
        super().__init__()
        self.predict_xstart = predict_xstart
        self.absolute_loss_conditioning = absolute_loss_conditioning
        in_sizes, out_sizes, in_names = self.compute_token_sizes(parameter_sizes, parameter_names, num_frequencies)
        self.decoder = GPT(in_sizes, out_sizes, in_names, **gpt_kwargs)
        self.scalar_embedder = FrequencyEmbedder(num_frequencies, max_freq_log2)
        self.use_global_residual = True
        for proj in self.decoder.output_parameter_projections:      # identity at initialisation (:363-367)
            proj[-1].weight.data.zero_()
        self.D = sum(flatten_parameter_sizes(parameter_sizes))
        self.n_scalar = 4 if absolute_loss_conditioning else 2
        self.token_sources = self._token_sources(num_frequencies)
        self._compute_dtype = os.environ.get("GPT_AMD_DTYPE", "fp32")
        self._engines = {}
        self._flat = None                # g.pt_amd/flat.py: FlatState once the parameters live in one arena

    # engines hold device pointers: never copy or pickle them with the module
    def __deepcopy__(self, memo):
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        from copy import deepcopy
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k == "_engines" else None if k == "_flat" else deepcopy(v, memo)
        return new

    def __getstate__(self):
        st = dict(self.__dict__)
        st["_engines"] = {}
        st["_flat"] = None
        return st

    @staticmethod
    def get_scalar_token_size(num_frequencies):
        return num_frequencies * 2 + 1

    def compute_token_sizes(self, parameter_sizes, parameter_names, num_frequencies):
        """Nested size lists fed to the trunk (transformer.py:376-407)."""
        scalar = [self.get_scalar_token_size(num_frequencies)]
        in_sizes = [list(parameter_sizes), list(parameter_sizes), scalar, scalar]
        names = list(parameter_names) * 2 + ["timestep_embedding", "loss_delta_embedding"]
        if self.absolute_loss_conditioning:
            in_sizes += [scalar, scalar]
            names += ["target_loss_embedding", "current_loss_embedding"]
        return in_sizes, [list(parameter_sizes)], names

    def _token_sources(self, num_frequencies):
        """Where each input token's columns live: (source kind, column offset, width).  The concatenated
        input [x | x_prev | t_emb | loss_embs] (transformer.py:458-459) is D + D + n_scalar*W wide; a token
        must not straddle two sources, which holds for every policy except 'chunk' on unaligned D."""
        W = self.get_scalar_token_size(num_frequencies)
        bounds = [(L.SRC_X, 0, self.D), (L.SRC_XPREV, self.D, 2 * self.D),
                  (L.SRC_SCALAR, 2 * self.D, 2 * self.D + self.n_scalar * W)]
        out, col = [], 0
        for k in self.decoder.input_splits:
            for kind, lo, hi in bounds:
                if lo <= col < hi:
                    if col + k > hi:
                        raise NotImplementedError("a token straddles two inputs (split_policy='chunk' with a "
                                                  "chunk size that does not divide the parameter count)")
                    out.append((kind, col - lo, k))
                    break
            col += k
        assert col == 2 * self.D + self.n_scalar * W
        assert len(out) <= L.MAX_TOKENS and len(out) <= 32, "the attention kernel handles at most 32 tokens"
        return out

    def configure_optimizers(self, lr, wd, betas):
        return GPT.configure_optimizers(self, lr, wd, betas)

    @torch.no_grad()
    def gradient_norm(self):
        """Global L2 norm of the gradients (transformer.py:415-425), one device reduction."""
        sq = torch.stack([p.grad.detach().float().pow(2).sum() for p in self.parameters()])
        return float(sq.sum().sqrt().item())

    # ---- compute dtype
    def set_compute_dtype(self, dtype):
        L.dtype_code(dtype)
        self._compute_dtype = dtype
        return self

    def _active_code(self):
        if torch.is_autocast_enabled():
            return L.BF16
        return L.dtype_code(self._compute_dtype)

    def engine(self, device=None, code=None):
        device = device if device is not None else self.decoder.pos_emb.device
        code = self._active_code() if code is None else code
        key = (str(device), code)
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = _Engine(self, device, code)
        eng.refresh()
        return eng

    # ---- forward
    def _prepare(self, x, t, loss_target, loss_prev, x_prev):
        assert x.dim() == 2 and x.shape[1] == self.D, f"x must be (N, {self.D})"
        assert x_prev is not None and loss_prev is not None, "G.pt needs loss_prev and x_prev"
        assert x_prev.shape == x.shape
        if not x.is_cuda:
            raise L.GptHipError("Gpt.forward: the HIP path needs GPU tensors; there is no CPU fallback")
        B, dev = x.shape[0], x.device
        x = x.detach().float().contiguous()
        x_prev = x_prev.detach().float().contiguous()
        t = t.reshape(-1)
        assert t.numel() == B and not t.is_floating_point(), "t must hold N integer timesteps"
        t = t.long().contiguous()
        lt, lp = _as_loss(loss_target, B, dev), _as_loss(loss_prev, B, dev)
        if lt.dtype != lp.dtype:
            lt, lp = lt.double(), lp.double()
        return x, t, lt, lp, x_prev

    def forward(self, x, t, loss_target, loss_prev=None, x_prev=None):
        """(N, D) denoised parameters; same contract as transformer.py:440-464."""
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self.training and self.decoder.dropout_p > 0:
            raise NotImplementedError("dropout > 0 is not implemented on the HIP path (every shipped config uses 0)")
        x, t, lt, lp, x_prev = self._prepare(x, t, loss_target, loss_prev, x_prev)
        if x.shape[0] == 0:                       # an empty batch: the reference returns an empty (0, D) tensor; no launch
            return torch.empty_like(x)
        if needs_grad:
            from .autograd import denoise_with_grad
            return denoise_with_grad(self, x, t, lt, lp, x_prev)
        eng = self.engine(x.device)
        B = x.shape[0]
        ws = eng.workspace(B)
        out = torch.empty_like(x)
        L.check(L.lib().gpt_denoiser_forward(C.byref(eng.cmodel), C.byref(ws), L.ptr(x), L.ptr(t), L.ptr(lt), L.ptr(lp),
                                             int(lt.dtype == torch.float64), L.ptr(x_prev), L.ptr(out), B, L.stream()),
                "gpt_denoiser_forward")
        return out
