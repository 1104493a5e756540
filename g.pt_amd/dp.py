"""Data-parallel layer: the reference's Gpt/distributed.py helper functions (live ones only, :35-72 and
:142-189) on torch.distributed, whose "nccl" backend is RCCL over xGMI on ROCm, plus the gradient
exchange used by the training step (reference: DistributedDataParallel, main.py:229-240).

One process per GPU; single-process calls short-circuit exactly as the reference's do."""
import os

import torch
import torch.distributed as dist


def _live():
    return dist.is_available() and dist.is_initialized()


def exchanging():
    """True when the collectives must run: more than one rank, or a one-rank process group under GPT_DP_REHEARSE=1
    (the RCCL calls of the multi-GPU path rehearsed on a one-GPU box: a one-rank SUM / gather / broadcast is the
    identity, so results must equal the single-process path's; tests/test_hip_training.py::test_one_rank_rccl_rehearsal_matches_single_process)."""
    return _live() and (dist.get_world_size() > 1 or os.environ.get("GPT_DP_REHEARSE") == "1")


def is_main_proc(local=False):
    return (not _live()) or dist.get_rank() == 0


def get_rank():
    return dist.get_rank() if _live() else 0


def get_world_size():
    return dist.get_world_size() if _live() else 1


def synchronize():
    if exchanging():
        dist.barrier()


def scaled_all_reduce(cfg, tensors):
    """In-place SUM all-reduce scaled by 1/num_gpus; returns the list (distributed.py:50-72).
    The tensors are reduced as ONE flat message instead of one collective each."""
    if cfg.num_gpus == 1:
        return tensors
    if len(tensors) == 1:
        dist.all_reduce(tensors[0])
        tensors[0].mul_(1.0 / cfg.num_gpus)
        return tensors
    # one flat message per dtype: every tensor is reduced and scaled in its OWN dtype, as the reference's per-tensor
    # all_reduce + mul_ does (float64 meters stay float64); non-float tensors take the reference's per-tensor route, with its behaviour
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, group in by_dtype.items():
        if not dtype.is_floating_point or len(group) == 1:
            for t in group:
                dist.all_reduce(t)
                t.mul_(1.0 / cfg.num_gpus)
            continue
        flat = torch.cat([t.reshape(-1) for t in group])
        dist.all_reduce(flat)
        flat.mul_(1.0 / cfg.num_gpus)
        off = 0
        for t in group:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return tensors


def all_gather(input, cat=True, dim=0):
    """distributed.py:170-183."""
    if not exchanging():
        return input if cat else input.unsqueeze(dim)
    parts = [torch.empty_like(input) for _ in range(get_world_size())]
    dist.all_gather(parts, input.contiguous())
    return torch.cat(parts, dim=dim) if cat else torch.stack(parts, dim=dim)


def rank0_to_all(input, cat=True):
    """Rank 0's value on every rank (distributed.py:186-189) as a broadcast."""
    if not exchanging():
        return (input if cat else input.unsqueeze(0))[0]
    out = input.clone()
    dist.broadcast(out, src=0)
    return (out if cat else out.unsqueeze(0))[0]


def shard_rows(n_rows, rank=None, world=None):
    """Contiguous row block [lo, hi) of a batch sharded for sampling (no traffic until the final gather)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    per = (n_rows + world - 1) // world
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def gather_rows(local, n_rows):
    """Final gather of row-sharded samples: (B_local, D) per rank -> (n_rows, D) on every rank."""
    world = get_world_size()
    if not exchanging():
        return local
    per = (n_rows + world - 1) // world
    pad = torch.zeros(per, *local.shape[1:], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if _host_staged(pad):                   # gloo rehearsal with several ranks on one GPU: its device all_gather does not return
        h_pad = pad.cpu()
        h_parts = [torch.empty_like(h_pad) for _ in range(world)]
        dist.all_gather(h_parts, h_pad)
        return torch.cat(h_parts, 0)[:n_rows].to(local.device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat(parts, 0)[:n_rows]


def max_over_ranks(value, device):
    """MAX of a host scalar over the ranks (the timed region of bench.py: the slowest rank counts)."""
    if not exchanging():
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def broadcast_parameters(module, src=0):
    """Initial parameter broadcast from rank 0 (what DDP does at construction, main.py:231-236), coalesced into one
    message per dtype instead of one collective per tensor."""
    if not exchanging():
        return
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t.data)
    for tensors in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()


class GradBuckets:
    """Gradient all-reduce (SUM then / world) over a few large flat buckets.

    The parameters' .grad tensors are views into flat fp32 buckets, so the exchange is `n_buckets` large
    RCCL all-reduces with no packing copies; a bucket is launched (async) as soon as the backward has
    produced every gradient in it, overlapping the exchange with the rest of the backward."""

    def __init__(self, params, bucket_bytes=256 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, self.index = [], {}
        cur, cur_n = [], 0
        for p in reversed(self.params):                  # backward produces the last layers first
            cur.append(p)
            cur_n += p.numel()
            if cur_n * 4 >= bucket_bytes:
                self._close(cur)
                cur, cur_n = [], 0
        if cur:
            self._close(cur)
        self.pending = []

    def _close(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=torch.float32, device=plist[0].device)
        off = 0
        for p in plist:
            p.grad = flat[off:off + p.numel()].view_as(p)
            self.index[id(p)] = len(self.buckets)
            off += p.numel()
        self.buckets.append(flat)

    def zero(self):
        for b in self.buckets:
            b.zero_()

    def launch(self, bucket_id):
        if get_world_size() > 1:
            self.pending.append((bucket_id, dist.all_reduce(self.buckets[bucket_id], async_op=True)))

    def launch_all(self):
        for i in range(len(self.buckets)):
            self.launch(i)

    def wait(self):
        w = get_world_size()
        for i, h in self.pending:
            h.wait()
            self.buckets[i].mul_(1.0 / w)
        self.pending = []


def all_reduce_ranges(flat, ranges, async_op=True):
    """SUM all-reduce of contiguous [lo, hi) element ranges of one flat buffer, one collective per range, launched in
    the given order (the order the backward finishes them) and left in flight; returns the work handles.
    The 1/world scale is NOT applied (the fused optimizer pass folds it in).
    A process group without a device transport (gloo: the CPU rehearsal of the multi-GPU path) gets device buffers
    staged through the host, synchronously."""
    if not exchanging():
        return []
    if flat.is_cuda and dist.get_backend() != "nccl":
        for lo, hi in ranges:
            host = flat[lo:hi].cpu()
            dist.all_reduce(host)
            flat[lo:hi].copy_(host)
        return []
    return [dist.all_reduce(flat[lo:hi], async_op=async_op) for lo, hi in ranges]


def _host_staged(t):
    """True for a device tensor on a process group without a device transport (gloo: the rehearsal of the multi-GPU path
    with several ranks on one GPU): the collective then runs on a host copy, synchronously."""
    return t.is_cuda and dist.get_backend() != "nccl"


def reduce_scatter_into(out, inp, async_op=True):
    """out[c] = SUM over ranks of inp[rank * c : (rank + 1) * c]  (inp: world * c elements; no 1/world scale).  The first half
    of a ring all-reduce: each rank receives only the 1/world of the sum it is going to update.  Returns a work handle or None."""
    if not exchanging():
        out.copy_(inp[:out.numel()])
        return None
    assert inp.numel() == out.numel() * get_world_size()
    if _host_staged(inp):
        h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter_tensor(h_out, h_in)
        out.copy_(h_out)
        return None
    return dist.reduce_scatter_tensor(out, inp, async_op=async_op)


def all_gather_into(full, own, async_op=True):
    """full[rank * c : (rank + 1) * c] = every rank's `own` (c elements).  `own` may be the matching slice of `full` itself
    (in place: nothing is copied locally).  The second half of a ring all-reduce.  Returns a work handle or None."""
    if not exchanging():
        if own.data_ptr() != full.data_ptr():
            full[:own.numel()].copy_(own)
        return None
    assert full.numel() == own.numel() * get_world_size()
    if _host_staged(full):
        h_own, h_full = own.cpu(), torch.empty(full.shape, dtype=full.dtype)
        dist.all_gather_into_tensor(h_full, h_own)
        full.copy_(h_full)
        return None
    return dist.all_gather_into_tensor(full, own, async_op=async_op)


def wait_all(handles):
    for h in handles:
        if h is not None:
            h.wait()
