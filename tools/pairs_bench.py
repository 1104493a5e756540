#!/usr/bin/env python3
"""Batch-fetch rate of the HBM-resident checkpoint store on a synthetic MNIST-shaped dataset.
Usage: python tools/pairs_bench.py [--runs 200] [--ckpts 200] [--batch 1024]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpt_amd import data as gdata  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    ap.add_argument("--ckpts", type=int, default=200)
    ap.add_argument("--batch", type=int, default=1024)
    a = ap.parse_args()
    D = 7960
    with tempfile.TemporaryDirectory() as root:
        gdata.write_synthetic_dataset(root, "mnist_loss", a.runs, a.ckpts)
        t0 = time.time()
        ds = gdata.CheckpointStore(root, "mnist_loss", split="train", num_test_runs=0, permute_augment=True)
        torch.cuda.synchronize()
        gb = ds.store.numel() * 4 / 1e9
        print(f"ingest: {ds.store.shape[0]} checkpoints, {gb:.2f} GB in {time.time() - t0:.1f} s")
        idx = torch.randint(0, len(ds), (a.batch,)).tolist()
        for name, rng in (("reference RNG order", None), ("vectorised draws", np.random.default_rng(0))):
            ds.get_batch(idx, rng)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(10):
                b = ds.get_batch(idx, rng)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rows = torch.randint(0, ds.store.shape[0], (a.batch,), device="cuda")
            perms = torch.stack([torch.randperm(10) for _ in range(a.batch)]).int()
            e0.record()
            ds._gather(rows, rows, True, perms)
            e1.record()
            torch.cuda.synchronize()
            print(f"{name}: {dt * 1e3:.2f} ms per batch of {a.batch} pairs (host + device); index build + gather alone "
                  f"{e0.elapsed_time(e1) * 1e3:.0f} us = {a.batch * D * (16 + 16) / e0.elapsed_time(e1) / 1e6:.0f} GB/s; "
                  f"shape {tuple(b['parameters_0'].shape)}")


if __name__ == "__main__":
    main()
