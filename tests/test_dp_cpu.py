"""Three-rank gloo run of the data-parallel layer's host logic on CPU tensors (g.pt_amd/dp.py: the live helpers of the reference's
Gpt/distributed.py:35-72, 142-189 plus the exchanges the training step and the sharded sampler are built from).  tests/test_dp_gloo.py
holds the two-rank cases; an odd world size adds what two ranks cannot show: a non-integer mean rank, a ragged last row block of the
sharded sampler, chunk offsets beyond the first two ranks.  No GPU: every collective runs over gloo on 127.0.0.1."""
import os
import socket
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gpt_amd import dp
        assert dp.exchanging() and dp.get_rank() == rank and dp.get_world_size() == world
        assert dp.is_main_proc() == (rank == 0)
        cfg = types.SimpleNamespace(num_gpus=world)
        mean_rank = (world - 1) / 2.0

        # scaled_all_reduce: one tensor, several of one dtype (one flat message), mixed dtypes (each reduced in its own)
        a = torch.full((5,), float(rank))
        assert dp.scaled_all_reduce(cfg, [a])[0] is a and torch.allclose(a, torch.full((5,), mean_rank))
        b, c = torch.arange(6.).view(2, 3) * (rank + 1), torch.full((4,), 2.0 * rank)
        d = torch.full((3,), float(rank), dtype=torch.float64)
        dp.scaled_all_reduce(cfg, [b, c, d])
        s = sum(r + 1 for r in range(world)) / world
        assert torch.allclose(b, torch.arange(6.).view(2, 3) * s) and torch.allclose(c, torch.full((4,), 2.0 * mean_rank))
        assert d.dtype == torch.float64 and torch.allclose(d, torch.full((3,), mean_rank, dtype=torch.float64))

        # all_gather (cat / stack) and rank0_to_all
        g = dp.all_gather(torch.full((2, 3), float(rank)))
        assert g.shape == (2 * world, 3) and all(bool((g[2 * r:2 * r + 2] == r).all()) for r in range(world))
        assert dp.all_gather(torch.full((2, 3), float(rank)), cat=False).shape == (world, 2, 3)
        assert float(dp.rank0_to_all(torch.tensor([10.0 + rank, 20.0 + rank]))) == 10.0

        # sharded sampling: contiguous row blocks (ragged last block) and the final gather
        n_rows = 4 * world - 1
        lo, hi = dp.shard_rows(n_rows)
        blocks = [dp.shard_rows(n_rows, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n_rows and all(blocks[r][1] == blocks[r + 1][0] for r in range(world - 1))
        rows = torch.arange(n_rows * 2, dtype=torch.float32).view(n_rows, 2)
        assert torch.equal(dp.gather_rows(rows[lo:hi].clone(), n_rows), rows)

        # timing reduction of bench.py
        assert dp.max_over_ranks(1.0 + rank, "cpu") == float(world)

        # broadcast of rank 0's parameters (DDP construction, main.py:231-236), two dtypes
        torch.manual_seed(100 + rank)
        net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3))
        net.register_buffer("steps", torch.tensor([rank], dtype=torch.int64))
        dp.broadcast_parameters(net)
        torch.manual_seed(100)
        ref = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.LayerNorm(3))
        assert all(torch.equal(p, q) for p, q in zip(net.parameters(), ref.parameters())) and int(net.steps) == 0

        # gradient exchange of the replicated step: flat buckets (mean) and arena ranges (sum, scale left to the optimizer pass)
        params = [torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(2))]
        gb = dp.GradBuckets(params, bucket_bytes=40)
        assert len(gb.buckets) >= 2 and sum(b.numel() for b in gb.buckets) == 24
        for i, p in enumerate(params):
            p.grad.fill_(float(rank + i))
        gb.launch_all()
        gb.wait()
        assert all(torch.allclose(p.grad, torch.full_like(p.grad, mean_rank + i)) for i, p in enumerate(params))
        gb.zero()
        assert all(float(p.grad.abs().sum()) == 0.0 for p in params)
        flat = torch.arange(12.) + 100 * rank
        dp.wait_all(dp.all_reduce_ranges(flat, [(8, 12), (0, 4)]))
        tot = 100.0 * sum(range(world))
        assert torch.equal(flat[4:8], torch.arange(4., 8.) + 100 * rank)              # outside the ranges: untouched
        assert torch.equal(flat[:4], world * torch.arange(4.) + tot) and torch.equal(flat[8:], world * torch.arange(8., 12.) + tot)

        # the sharded step's pair: reduce-scatter of a bucket, all-gather of the updated chunks (in place)
        c = 3
        inp = torch.arange(world * c, dtype=torch.float32) * (rank + 1)
        own = torch.empty(c)
        dp.wait_all([dp.reduce_scatter_into(own, inp)])
        assert torch.equal(own, torch.arange(rank * c, (rank + 1) * c, dtype=torch.float32) * sum(r + 1 for r in range(world)))
        full = torch.full((world * c,), -1.0)
        full[rank * c:(rank + 1) * c] = own
        dp.wait_all([dp.all_gather_into(full, full[rank * c:(rank + 1) * c])])
        assert torch.equal(full, torch.arange(world * c, dtype=torch.float32) * sum(r + 1 for r in range(world)))
        dp.synchronize()
        open(os.path.join(out_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [3])
def test_dp_helpers_over_gloo_odd_world(tmp_path, world):
    mp.start_processes(_rank_main, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(world)]
