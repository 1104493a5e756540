"""Data-parallel layer on 2 CPU ranks (gloo): the helpers of Gpt/distributed.py:35-72, 142-189 and the gradient
exchange used by the training step.  The same code runs on RCCL ("nccl" backend) on the GPUs."""
import os
import socket
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, port, fn_name, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        globals()[fn_name](rank)
        out.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _run(fn_name):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, fn_name, out)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}:\n{msg}"


# ---------------------------------------------------------------------------------------------- rank bodies
def _body_helpers(rank):
    from gpt_amd import dp
    cfg = types.SimpleNamespace(num_gpus=WORLD)
    assert dp.get_rank() == rank and dp.get_world_size() == WORLD and dp.is_main_proc() == (rank == 0)
    # scaled_all_reduce: in place, mean over ranks, returns the list (distributed.py:50-72)
    a, b = torch.full((3,), float(rank + 1)), torch.full((2, 2), 10.0 * (rank + 1))
    res = dp.scaled_all_reduce(cfg, [a, b])
    assert res[0] is a and res[1] is b
    assert torch.allclose(a, torch.full((3,), 1.5)) and torch.allclose(b, torch.full((2, 2), 15.0))
    # mixed dtypes keep their own dtype and precision (float64 meters are not squeezed through fp32)
    d64 = torch.tensor([1.0 + 2.0 ** -40 * (rank + 1)], dtype=torch.float64)
    f32, f32b = torch.tensor([float(rank)]), torch.tensor([4.0 * rank, 8.0])
    dp.scaled_all_reduce(cfg, [d64, f32, f32b])
    assert d64.dtype == torch.float64 and float(d64) == 1.0 + 1.5 * 2.0 ** -40
    assert float(f32) == 0.5 and f32b.tolist() == [2.0, 8.0]
    one = [torch.tensor([2.0 * rank])]
    dp.scaled_all_reduce(cfg, one)
    assert float(one[0]) == 1.0
    # single-GPU config short-circuits
    c = torch.tensor([7.0])
    assert dp.scaled_all_reduce(types.SimpleNamespace(num_gpus=1), [c])[0] is c and float(c) == 7.0
    # all_gather cat / stack (distributed.py:170-183)
    x = torch.arange(4.0).view(2, 2) + 100 * rank
    g = dp.all_gather(x)
    assert g.shape == (4, 2) and torch.equal(g[:2], torch.arange(4.0).view(2, 2)) and torch.equal(g[2:], x - 100 * rank + 100)
    assert dp.all_gather(x, cat=False).shape == (2, 2, 2)
    # rank0_to_all (distributed.py:186-189)
    r0 = dp.rank0_to_all(torch.tensor([[1.0 + rank, 2.0]]))
    assert torch.equal(r0, torch.tensor([1.0, 2.0]))
    dp.synchronize()


def _body_rows(rank):
    from gpt_amd import dp
    n, D = 7, 5                                       # ragged: 4 rows on rank 0, 3 on rank 1
    lo, hi = dp.shard_rows(n)
    assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
    full = torch.arange(n * D, dtype=torch.float32).view(n, D)
    got = dp.gather_rows(full[lo:hi].clone(), n)
    assert torch.equal(got, full)
    # parameter broadcast from rank 0
    lin = torch.nn.Linear(4, 3)
    with torch.no_grad():
        for p in lin.parameters():
            p.fill_(float(rank + 1))
    dp.broadcast_parameters(lin)
    assert all(float(p.min()) == 1.0 and float(p.max()) == 1.0 for p in lin.parameters())


def _body_grads(rank):
    from gpt_amd import dp
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.GELU(), torch.nn.Linear(16, 4))
    ref.load_state_dict(net.state_dict())
    data = torch.randn(2 * 6, 8, generator=torch.Generator().manual_seed(1))
    # reference: the full batch on one process; mean loss over 12 rows == mean over ranks of the mean over 6 rows
    ref(data).pow(2).mean().backward()
    buckets = dp.GradBuckets(net.parameters(), bucket_bytes=256)
    assert len(buckets.buckets) > 1
    buckets.zero()
    net(data[rank * 6:(rank + 1) * 6]).pow(2).mean().backward()
    buckets.launch_all()
    buckets.wait()
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, atol=1e-6), (p.grad - q.grad).abs().max()
    # flat-arena exchange used by TrainStep: ranges reduced in backward order, scale applied by the caller
    flat = torch.arange(10.0) * (rank + 1)
    h = dp.all_reduce_ranges(flat, [(6, 10), (2, 6), (0, 2)])
    dp.wait_all(h)
    assert torch.equal(flat, torch.arange(10.0) * 3)
    # the two halves of that exchange on their own (the sharded optimizer step): rank r receives the sum of chunk r, updates it,
    # and the chunks are gathered back in place
    arena = torch.arange(12.0) * (rank + 1)
    own = torch.zeros(3)
    dp.wait_all([dp.reduce_scatter_into(own, arena[2:8])])
    assert torch.equal(own, (torch.arange(12.0) * 3)[2 + 3 * rank: 5 + 3 * rank])
    arena[2 + 3 * rank: 5 + 3 * rank] = own * 10
    dp.wait_all([dp.all_gather_into(arena[2:8], arena[2 + 3 * rank: 5 + 3 * rank])])
    assert torch.equal(arena[2:8], torch.arange(12.0)[2:8] * 30)
    assert torch.equal(arena[:2], torch.arange(2.0) * (rank + 1)) and torch.equal(arena[8:], torch.arange(12.0)[8:] * (rank + 1))
    half = torch.tensor([1.0, 2.0, 3.0, 4.0]).bfloat16() * (rank + 1)
    got = torch.zeros(2, dtype=torch.bfloat16)
    dp.wait_all([dp.reduce_scatter_into(got, half)])
    assert got.float().tolist() == ([3.0, 6.0] if rank == 0 else [9.0, 12.0])


# ---------------------------------------------------------------------------------------------- tests
def test_distributed_helpers_two_ranks():
    _run("_body_helpers")


def test_row_sharding_gather_and_broadcast_two_ranks():
    _run("_body_rows")


def test_gradient_exchange_matches_full_batch_two_ranks():
    _run("_body_grads")


def test_single_process_short_circuits():
    sys.path.insert(0, ROOT)
    from gpt_amd import dp
    assert dp.get_world_size() == 1 and dp.get_rank() == 0 and dp.is_main_proc()
    x = torch.arange(6.0).view(3, 2)
    assert dp.all_gather(x) is x and dp.gather_rows(x, 3) is x
    assert torch.equal(dp.rank0_to_all(x), x[0])
    assert dp.all_reduce_ranges(x.view(-1), [(0, 6)]) == []
    out = torch.zeros(6)
    assert dp.reduce_scatter_into(out, x.view(-1)) is None and torch.equal(out, x.view(-1))
    assert dp.all_gather_into(out, out) is None
    assert dp.shard_rows(10, rank=1, world=4) == (3, 6) and dp.shard_rows(10, rank=3, world=4) == (9, 10)
