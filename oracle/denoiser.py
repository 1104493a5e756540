"""Oracle restatement of the G.pt denoiser forward (TEST INFRASTRUCTURE, see oracle/__init__.py).

Functional form: ``forward(sd, plan, x, t, loss_target, loss_prev, x_prev)`` where ``sd`` is a
reference-layout state_dict.  Every function cites the reference lines it
follows (paths relative to /root/reference).
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- token plan
def flatten_sizes(sizes):
    """Gpt/models/transformer.py:244-249."""
    assert len(sizes) > 0
    if not isinstance(sizes[0], (list, tuple)):
        return list(sizes)
    return [p for g in sizes for p in g]


def build_splits(sizes, policy="chunk", chunk=None):
    """Gpt/models/transformer.py:251-300 (four tokenisation policies)."""
    if policy == "chunk":
        assert chunk is not None
        total = sum(flatten_sizes(sizes))
        out = [chunk] * (total // chunk)
        if total % chunk:
            out.append(total % chunk)
        return out
    if policy == "layer_by_layer":
        return flatten_sizes(sizes)
    if policy == "chunk_within_layer":
        assert chunk is not None
        out = []
        for n in flatten_sizes(sizes):
            out += [chunk] * (n // chunk)
            if n % chunk:
                out.append(n % chunk)
        return out
    if policy == "chunk_within_input":
        out = []
        for group in sizes:
            assert isinstance(group, (list, tuple))
            out += build_splits(group, "chunk", chunk)
        return out
    raise NotImplementedError(policy)


class Plan:
    """Token plan of one G.pt model: Gpt.compute_token_sizes (transformer.py:369-407) +
    GPT.__init__ (:90-93)."""

    def __init__(self, parameter_sizes, num_frequencies=128, max_freq_log2=20,
                 absolute_loss_conditioning=False, n_head=12, n_embd=768, n_layer=12,
                 chunk_size=None, split_policy="chunk", predict_xstart=True):
        scalar = [2 * num_frequencies + 1]                       # :370-374
        inp = [list(parameter_sizes), list(parameter_sizes), scalar, scalar]
        if absolute_loss_conditioning:
            inp += [scalar, scalar]
        out = [list(parameter_sizes)]
        self.input_splits = build_splits(inp, split_policy, chunk_size)
        self.output_splits = build_splits(out, split_policy, chunk_size)
        self.D = sum(parameter_sizes)
        self.T = len(self.input_splits)
        self.n_head, self.n_embd, self.n_layer = n_head, n_embd, n_layer
        self.num_frequencies, self.max_freq_log2 = num_frequencies, max_freq_log2
        self.absolute = absolute_loss_conditioning
        self.predict_xstart = predict_xstart


def frequencies(num_frequencies, max_freq_log2):
    """FrequencyEmbedder.__init__ (transformer.py:319-322): fp32 buffer 2**linspace."""
    return 2 ** torch.linspace(0, max_freq_log2, steps=num_frequencies)


# --------------------------------------------------------------------------- pieces
def freq_embed(s, freqs):
    """FrequencyEmbedder.forward (transformer.py:324-334) for (N,) or (N,1) input:
    cast to fp32 FIRST, multiply in fp32, full-range sin/cos, cat [sin, cos, s]."""
    n = s.shape[0]
    if s.dim() == 1:
        s = s.unsqueeze(1)
    su = s.unsqueeze(-1).to(torch.float)
    scaled = freqs.view(1, 1, -1) * su
    return torch.cat([torch.sin(scaled), torch.cos(scaled), su], dim=-1).view(n, -1)


def embed_loss(loss_target, loss_prev, freqs, absolute):
    """Gpt.embed_loss (transformer.py:427-438): the subtraction happens in the INPUT dtype."""
    parts = [freq_embed(loss_target - loss_prev, freqs)]
    if absolute:
        parts += [freq_embed(loss_target, freqs), freq_embed(loss_prev, freqs)]
    return torch.cat(parts, 1)


def attention(sd, pre, x, n_head):
    """SelfAttention.forward (transformer.py:34-51), dropout p=0."""
    B, T, C = x.shape
    hd = C // n_head
    q = F.linear(x, sd[pre + "query.weight"], sd[pre + "query.bias"]).view(B, T, n_head, hd).transpose(1, 2)
    k = F.linear(x, sd[pre + "key.weight"], sd[pre + "key.bias"]).view(B, T, n_head, hd).transpose(1, 2)
    v = F.linear(x, sd[pre + "value.weight"], sd[pre + "value.bias"]).view(B, T, n_head, hd).transpose(1, 2)
    att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
    att = F.softmax(att, dim=-1)
    y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
    return F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def block(sd, pre, x, n_head):
    """Block.forward (transformer.py:69-72): pre-LN, erf-GELU MLP."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[pre + "ln1.weight"], sd[pre + "ln1.bias"], 1e-5)
    x = x + attention(sd, pre + "attn.", h, n_head)
    h = F.layer_norm(x, (C,), sd[pre + "ln2.weight"], sd[pre + "ln2.bias"], 1e-5)
    h = F.linear(h, sd[pre + "mlp.0.weight"], sd[pre + "mlp.0.bias"])
    h = F.gelu(h)
    h = F.linear(h, sd[pre + "mlp.2.weight"], sd[pre + "mlp.2.bias"])
    return x + h


def encode(sd, plan, inp):
    """GPT.encode_parameters (transformer.py:215-228), encoder_depth=1."""
    C = plan.n_embd
    toks = []
    off = 0
    for i, k in enumerate(plan.input_splits):
        w = sd[f"decoder.input_parameter_projections.{i}.0.weight"]
        toks.append(F.linear(inp[:, off:off + k], w))
        off += k
    assert off == inp.shape[1]
    h = torch.stack(toks, dim=1)
    return F.layer_norm(h, (C,), sd["decoder.ln_in.weight"], sd["decoder.ln_in.bias"], 1e-5)


def decode(sd, plan, feats):
    """GPT.decode_parameters (transformer.py:230-242), decoder_depth=1: only the first
    len(output_splits) tokens are decoded."""
    outs = []
    for j in range(len(plan.output_splits)):
        w = sd[f"decoder.output_parameter_projections.{j}.0.weight"]
        outs.append(F.linear(feats[:, j, :], w))
    return torch.cat(outs, 1)


def forward(sd, plan, x, t, loss_target, loss_prev, x_prev, taps=None):
    """Gpt.forward (transformer.py:440-464) + GPT.forward (:302-314).

    ``taps``: optional dict that receives named intermediates (post ln_in+pos, per-block
    outputs, pre-decode features) for kernel-level parity tests."""
    freqs = sd["scalar_embedder.frequencies"]
    t_emb = freq_embed(t, freqs)
    l_emb = embed_loss(loss_target, loss_prev, freqs, plan.absolute)
    inp = torch.cat([x, x_prev, t_emb, l_emb], 1)
    h = encode(sd, plan, inp)
    assert h.shape[1] == plan.T
    h = h + sd["decoder.pos_emb"][:, :plan.T, :]
    if taps is not None:
        taps["scalar_tokens"] = torch.cat([t_emb, l_emb], 1)
        taps["embed"] = h.clone()
    for l in range(plan.n_layer):
        h = block(sd, f"decoder.blocks.{l}.", h, plan.n_head)
        if taps is not None:
            taps[f"block{l}"] = h.clone()
    C = plan.n_embd
    h = F.layer_norm(h, (C,), sd["decoder.ln_f.weight"], sd["decoder.ln_f.bias"], 1e-5)
    if taps is not None:
        taps["ln_f"] = h.clone()
    out = decode(sd, plan, h)
    return out + x_prev                                           # :462-463 global residual


# --------------------------------------------------------------------------- synthetic weights
def random_state_dict(plan, seed=0, std=0.02, dtype=torch.float32):
    """Seeded reference-layout state_dict with every tensor non-trivial:
    the reference zero-initialises pos_emb and the last decoder Linear (transformer.py:101,
    363-367), which would make a fresh model return x_prev and test nothing."""
    g = torch.Generator().manual_seed(seed)
    C, L = plan.n_embd, plan.n_layer

    def rn(*shape, s=std):
        return (torch.randn(*shape, generator=g) * s).to(dtype)

    sd = {"decoder.pos_emb": rn(1, plan.T, C)}
    for l in range(L):
        p = f"decoder.blocks.{l}."
        for ln in ("ln1", "ln2"):
            sd[p + ln + ".weight"] = 1.0 + rn(C, s=0.1)
            sd[p + ln + ".bias"] = rn(C, s=0.1)
        for nm in ("key", "query", "value", "proj"):
            sd[p + f"attn.{nm}.weight"] = rn(C, C)
            sd[p + f"attn.{nm}.bias"] = rn(C)
        sd[p + "mlp.0.weight"] = rn(4 * C, C)
        sd[p + "mlp.0.bias"] = rn(4 * C)
        sd[p + "mlp.2.weight"] = rn(C, 4 * C)
        sd[p + "mlp.2.bias"] = rn(C)
    for i, k in enumerate(plan.input_splits):
        sd[f"decoder.input_parameter_projections.{i}.0.weight"] = rn(C, k)
    sd["decoder.ln_in.weight"] = 1.0 + rn(C, s=0.1)
    sd["decoder.ln_in.bias"] = rn(C, s=0.1)
    sd["decoder.ln_f.weight"] = 1.0 + rn(C, s=0.1)
    sd["decoder.ln_f.bias"] = rn(C, s=0.1)
    for j, n in enumerate(plan.output_splits):
        sd[f"decoder.output_parameter_projections.{j}.0.weight"] = rn(n, C)
    sd["scalar_embedder.frequencies"] = frequencies(plan.num_frequencies, plan.max_freq_log2)
    return sd
