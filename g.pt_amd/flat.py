"""Flat parameter arenas of one G.pt replica.

The checkpoint ABI keeps ~235 separate tensors; on the device they become views of ONE fp32
arena laid out like the gradient arena of backward.py (query | key | value of a layer back to back, so the fused
[3d, d] operand is the storage itself).  The optimizer step, the EMA update, the gradient all-reduce and the
low-precision operand copies then each run as one pass over a flat buffer instead of 235 small launches.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .backward import GradArena


def _round_up(n, m):
    return (n + m - 1) // m * m


class FlatState:
    """fp32 arena holding every parameter of `model` (+ the packed operand copies per compute dtype)."""

    def __init__(self, model, device=None):
        self.device = device if device is not None else model.decoder.pos_emb.device
        self.arena = GradArena(model, self.device)
        self.numel = self.arena.numel
        self.params = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        self.names = [n for n, _ in model.named_parameters()]
        with torch.no_grad():
            for name, p in model.named_parameters():
                o, n = self.arena.slices[name]
                view = self.params[o:o + n].view_as(p)
                view.copy_(p.data)
                p.data = view
        decay_ids = {id(p) for p in model.configure_optimizers(lr=0.0, wd=0.0, betas=(0.9, 0.999)).param_groups[0]["params"]}
        self.decay = {n: int(id(p) in decay_ids) for n, p in model.named_parameters()}
        dec = model.decoder
        self.enc_names = [f"decoder.input_parameter_projections.{i}.0.weight" for i in range(len(dec.input_splits))]
        self.enc_k = list(dec.input_splits)
        self.d = dec.n_embd
        # packed region of the K-padded encoder weights, appended behind the arena-shaped part
        self.enc_off, off = [], self.numel
        for k in self.enc_k:
            self.enc_off.append(off)
            off += self.d * _round_up(k, 32)
        self.packed_numel = off
        self._packed, self._segments = {}, {}
        model._flat = self

    # ---- bookkeeping
    def bound(self, model):
        """True while every parameter still is its view of the arena (a later .to()/.cuda() re-allocates them)."""
        base = self.params.data_ptr()
        for name, p in model.named_parameters():
            if p.data_ptr() != base + 4 * self.arena.slices[name][0]:
                return False
        return True

    def like(self):
        return torch.zeros_like(self.params)

    def views(self, buf, model):
        return self.arena.views(buf, model)

    # ---- packed operand copies
    def segment_rows(self, code):
        """[(start, count, decay, k, ld, packed_off)] of every parameter in arena order (gpt_opt_segment_t, include/gpt_hip.h)."""
        rows = []
        enc = {n: i for i, n in enumerate(self.enc_names)}
        for name, (o, n) in sorted(self.arena.slices.items(), key=lambda kv: kv[1][0]):
            k, ld, packed_off = 1, 1, -1
            if name in enc:
                i = enc[name]
                k, ld, packed_off = self.enc_k[i], _round_up(self.enc_k[i], 32), self.enc_off[i] - self._base(code)
            elif code == L.BF16 and self.decay[name]:
                packed_off = o
            rows.append((o, n, self.decay[name], k, ld, packed_off))
        return rows

    def segment_table(self, code, lo, hi):
        """Device table of the segments that overlap arena elements [lo, hi), with their starts relative to `lo`: the table of
        an optimizer pass over params + lo ... (a data-parallel rank's shard, trainstep.py).  packed_off stays absolute."""
        rows = self.segment_rows(code)
        ends = [r[0] for r in rows[1:]] + [self.numel]
        sel = [r for r, e in zip(rows, ends) if r[0] < hi and e > lo]
        arr = (L.OptSegment * len(sel))()
        for s, (o, n, decay, k, ld, packed_off) in zip(arr, sel):
            s.start, s.count, s.decay, s.k, s.ld, s.packed_off = o - lo, n, decay, k, ld, packed_off
            # a partly covered segment is only safe for plain copies (the K-padded layout indexes from the segment's start)
            assert k == ld or (o >= lo and o + n <= hi) or packed_off < 0
            if packed_off >= 0 and k == ld:
                s.packed_off = packed_off        # element e of the pass maps to packed_off + (e - start)
        host = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        return torch.from_numpy(host).to(self.device), len(sel)

    def segments(self, code):
        seg = self._segments.get(code)
        if seg is None:
            seg = self._segments[code] = self.segment_table(code, 0, self.numel)
        return seg

    def _base(self, code):
        # fp32 mode packs only the encoder weights (every other operand is the arena itself)
        return 0 if code == L.BF16 else self.numel

    def packed(self, code):
        buf = self._packed.get(code)
        if buf is None:
            n = self.packed_numel - self._base(code)
            buf = self._packed[code] = torch.zeros(n, dtype=L.torch_dtype(code), device=self.device)
            self.repack(code)
        return buf

    def repack(self, code):
        buf = self._packed.get(code)
        if buf is None:
            return self.packed(code)
        seg, n_seg = self.segments(code)
        L.check(L.lib().gpt_pack_segments(L.ptr(self.params), L.ptr(buf), code, L.ptr(seg), n_seg, self.numel, L.stream()),
                "gpt_pack_segments")
        return buf

    def weight_ptr(self, code, name):
        """Device address of the GEMM operand of a Linear weight in compute dtype `code`."""
        o = self.arena.slices[name][0]
        if code == L.BF16:
            return self.packed(code).data_ptr() + 2 * o
        return self.params.data_ptr() + 4 * o

    def enc_ptr(self, code, i):
        es = 2 if code == L.BF16 else 4
        return self.packed(code).data_ptr() + es * (self.enc_off[i] - self._base(code))

    def param_ptr(self, name):
        return self.params.data_ptr() + 4 * self.arena.slices[name][0]
