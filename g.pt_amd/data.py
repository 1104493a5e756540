"""Checkpoint-pair input pipeline with the whole split resident in HBM.

The reference feeds training from LMDBs through a `num_workers: 1` CPU DataLoader (Gpt/data/dataset_lmdb.py,
Gpt/utils.py:80-91): per sample two value look-ups, a normalisation and a permutation gather on the host.  An MI355X holds
288 GB: the whole MNIST checkpoint set (68 GB) - or a rank's share of a larger one - fits next to the model.  So the split
is read ONCE into one [n_checkpoints, D] fp32 tensor on the GPU, and a training batch is one `gpt_pair_gather` launch
(row gather + normalisation + permutation augmentation, HBM-bound) driven by two small index vectors; only the pair
sampling itself (two integers per sample) stays on the host, where it reproduces the reference's draws bit for bit.

`CheckpointStore` keeps ParameterDataset's constructor arguments, attributes (`parameter_sizes`, `runs`, `min_val`,
`max_val`, `optimal_test_loss`, `epoch_size`, `normalize`/`unnormalize`/`get_range`, `get_run_losses`, `get_run_network`)
and batch keys (`parameters_0/1`, `<metric>_0/1`, `checkpoint_key_0/1`, `run_name`, `step_spacing`).

Storage back-ends (`open_db(path) -> mapping key -> 1-D fp32 tensor`): the reference's own `Gpt.data.database.Database`
where py-lmdb is installed (value = raw little-endian fp32 bytes under "<run>/ep{E:03d}_it{I:05d}.pt"-style keys, database.py:
51-62), or the flat pair `checkpoints.f32` + `checkpoints.json` written by `write_flat_db` (one sequential read per
directory, no B-tree walk)."""
import json
import os
from glob import glob

import numpy as np
import torch

from . import _lib as L
from . import permute as P

FLAT_DATA, FLAT_INDEX = "checkpoints.f32", "checkpoints.json"


class FlatDb:
    """key -> fp32 tensor view over one memory-mapped file of little-endian fp32 values (the LMDB value format)."""

    def __init__(self, path):
        with open(os.path.join(path, FLAT_INDEX)) as f:
            self.index = json.load(f)["keys"]
        self.data = np.memmap(os.path.join(path, FLAT_DATA), dtype="<f4", mode="r")

    def keys(self):
        return list(self.index.keys())

    def __contains__(self, key):
        return key in self.index

    def __getitem__(self, key):
        off, n = self.index[key]
        return torch.from_numpy(np.array(self.data[off:off + n]))


def write_flat_db(path, items):
    """items: iterable of (key, 1-D tensor) -> the flat pair under `path` (e.g. items = iter(Database(path)))."""
    os.makedirs(path, exist_ok=True)
    index, off = {}, 0
    with open(os.path.join(path, FLAT_DATA), "wb") as f:
        for key, value in items:
            a = value.detach().cpu().to(torch.float32).contiguous().numpy().astype("<f4", copy=False)
            f.write(a.tobytes())
            index[key] = [off, int(a.size)]
            off += int(a.size)
    with open(os.path.join(path, FLAT_INDEX), "w") as f:
        json.dump({"keys": index}, f)


def write_synthetic_dataset(root, dataset_name="mnist_loss", runs=64, ckpts=64, train_metric="avg_test_loss", seed=0, std=0.25):
    """A checkpoint dataset of the reference's directory layout (one directory per run: runs.json + the flat value file)
    filled with seeded noise: for benchmarks and smoke runs where the real 68-275 GB sets are not present."""
    sizes = {"mlp": [7840, 10, 100, 10], "cnn": [432, 4608, 320, 10]}[_ARCH[dataset_name]["kind"]]
    D, g = sum(sizes), torch.Generator().manual_seed(seed)
    opt_key = f"optimal_{train_metric}".replace("_avg", "")
    for r in range(runs):
        run, ck, items = f"r{r:05d}", {}, []
        vals = torch.randn(ckpts, D, generator=g) * std
        for i in range(ckpts):
            key = f"{run}/ep{i:03d}_it00000.pt"
            items.append((key, vals[i]))
            ck[key] = {train_metric: 2.3 - 2.0 * i / ckpts}
        items.append((items[0][0] + "_arch", torch.tensor(sizes, dtype=torch.float32)))
        write_flat_db(os.path.join(root, run), items)
        series = [v[train_metric] for v in ck.values()]
        with open(os.path.join(root, run, "runs.json"), "w") as f:
            json.dump({"checkpoints": ck, "metadata": {train_metric: series, opt_key: min(series),
                                                       "min_parameter_val": float(vals.min()), "max_parameter_val": float(vals.max())}}, f)


def default_open_db(path):
    if os.path.exists(os.path.join(path, FLAT_INDEX)):
        return FlatDb(path)
    try:
        import lmdb  # noqa: F401
    except ImportError as e:
        raise L.GptHipError(f"{path}: no {FLAT_INDEX} and py-lmdb is not installed; convert the LMDB once with "
                            "g.pt_amd.data.write_flat_db on a machine that has it") from e
    return _LmdbDb(path)


class _LmdbDb:
    """Read-only view of a reference LMDB (Gpt/data/database.py:51-57, 92-100)."""

    def __init__(self, path):
        import lmdb
        self.env = lmdb.open(path, map_size=10485760, readonly=True, readahead=False, meminit=False, create=False, lock=False)

    def keys(self):
        with self.env.begin() as txn, txn.cursor() as cur:
            return [k.decode() for k in cur.iternext(values=False)]

    def __contains__(self, key):
        with self.env.begin() as txn:
            return txn.get(key.encode()) is not None

    def __getitem__(self, key):
        with self.env.begin() as txn:
            v = txn.get(key.encode())
        if v is None:
            raise KeyError(f"Key not in database: {key}")
        return torch.frombuffer(bytearray(v), dtype=torch.float32)


_ARCH = {"mnist_loss": P.MLP_ARCH, "mnist_error": P.MLP_ARCH, "cifar10_loss": P.CNN_ARCH, "cifar10_error": P.CNN_ARCH,
         "cartpole": P.ACTOR_ARCH}
_MINIMIZE = {"mnist_loss": True, "mnist_error": True, "cifar10_loss": True, "cifar10_error": True,
             "cartpole": False}                                                                     # Gpt/tasks.py:36-81


class CheckpointStore:
    def __init__(self, dataset_dir="checkpoint_datasets/mnist", dataset_name="mnist_loss", split="train",
                 max_train_runs=1000000, num_test_runs=500, target_epoch_size=806400, train_metric="avg_test_loss",
                 min_step_spacing=1, max_step_spacing=None, normalizer_name="openai", openai_coeff=4.185,
                 single_run_debug=False, min_val=None, max_val=None, permute_augment=False, verify_done=False,
                 device="cuda", open_db=default_open_db, generator=None):
        if dataset_name not in _ARCH:
            raise NotImplementedError(f"task {dataset_name!r}: the MNIST MLP, CIFAR-10 ConvNet and Cartpole policy tasks are built")
        if normalizer_name not in ("openai", "none"):
            raise KeyError(normalizer_name)                                     # normalization.py:66-69
        lmdb_paths = sorted(glob(f"{dataset_dir}/*"))                           # dataset_lmdb.py:48-50
        if len(lmdb_paths) == 0:
            raise FileNotFoundError(dataset_dir)
        if verify_done:
            lmdb_paths = [p for p in lmdb_paths if os.path.exists(os.path.join(p, "DONE.txt"))]
        run_path = {}
        for p in lmdb_paths:                                                    # one directory may hold several runs
            for run in os.path.basename(p).split("_"):
                run_path[run] = p
        runs = list(run_path.keys())
        assert split in ["train", "test"]
        assert num_test_runs < len(runs)
        if split == "train":
            runs = runs[:-num_test_runs] if num_test_runs > 0 else runs
            runs = runs[:max_train_runs]
        else:
            runs = runs[-num_test_runs:] if num_test_runs > 0 else []
        self.runs, self.split = runs, split
        self.run_lmdb_path = {run: run_path[run] for run in runs}
        self.run_jsons = []
        for run in runs:
            with open(os.path.join(self.run_lmdb_path[run], "runs.json")) as f:
                data = json.load(f)
            self.run_jsons.append(data[run] if run in data else data)            # fused jsons (:84-86)
        self.num_runs = len(runs)
        self.dataset_name, self.train_metric = dataset_name, train_metric
        self.min_step_spacing, self.max_step_spacing = min_step_spacing, max_step_spacing
        self.single_run_debug, self.generator = single_run_debug, generator
        self.arch = _ARCH[dataset_name]
        self.use_augment = bool(permute_augment)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.GptHipError("CheckpointStore keeps the split in GPU memory; there is no CPU path")

        # ---- one pass over the databases: every checkpoint of the split into one [n, D] device tensor
        dbs = {p: open_db(p) for p in dict.fromkeys(self.run_lmdb_path.values())}
        first = list(self.run_jsons[0]["checkpoints"].keys())[0]
        self.parameter_sizes = dbs[self.run_lmdb_path[runs[0]]][f"{first}_arch"].long().tolist()
        self.D = int(sum(self.parameter_sizes))
        assert self.D == P.arch_dim(self.arch), f"{dataset_name}: {self.D} parameters, the task net has {P.arch_dim(self.arch)}"
        self.num_checkpoints = [len(j["checkpoints"]) for j in self.run_jsons]
        self.checkpoint_names = [list(j["checkpoints"].keys()) for j in self.run_jsons]
        self.run_offset = np.concatenate([[0], np.cumsum(self.num_checkpoints)]).astype(np.int64)
        total = int(self.run_offset[-1])
        self.store = torch.empty(total, self.D, dtype=torch.float32, device=self.device)
        stage = torch.empty(min(total, 4096), self.D, dtype=torch.float32).pin_memory()
        metric, row = np.empty(total, dtype=np.float64), 0
        for r, run in enumerate(runs):
            db, ck = dbs[self.run_lmdb_path[run]], self.run_jsons[r]["checkpoints"]
            for name in self.checkpoint_names[r]:
                v = db[name]
                assert v.numel() == self.D, f"{name}: {v.numel()} values, expected {self.D}"
                stage[row % stage.shape[0]].copy_(v)
                metric[row] = ck[name][train_metric]
                row += 1
                if row % stage.shape[0] == 0 or row == total:
                    n = (row - 1) % stage.shape[0] + 1
                    self.store[row - n:row].copy_(stage[:n])
        self.metric = metric

        reduce_fn = min if _MINIMIZE[dataset_name] else max
        self.optimal_test_loss = self.reduce_metadata(f"optimal_{train_metric}".replace("_avg", ""), reduce_fn)
        self.min_val, self.max_val = min_val, max_val
        self.min_val, self.max_val = self.get_range(normalize=False)
        self.openai_coeff = float(openai_coeff) if normalizer_name == "openai" else None
        assert target_epoch_size >= self.num_runs, \
            f"target_epoch_size ({target_epoch_size}) is less than the number of runs ({self.num_runs})"
        self.epoch_size = self.num_runs * (target_epoch_size // self.num_runs)

    # ---- ParameterDataset's helpers
    def __len__(self):
        return self.epoch_size

    def reduce_metadata(self, key, reduce_fn=max):
        return reduce_fn(j["metadata"][key] for j in self.run_jsons)

    def get_range(self, normalize=True):
        if self.min_val is None and self.max_val is None:
            lo, hi = self.reduce_metadata("min_parameter_val", min), self.reduce_metadata("max_parameter_val", max)
        else:
            lo, hi = self.min_val, self.max_val
        if normalize and self.openai_coeff is not None:
            lo, hi = lo * self.openai_coeff, hi * self.openai_coeff
        return lo, hi

    def normalize(self, weights):
        return weights * self.openai_coeff if self.openai_coeff is not None else weights

    def unnormalize(self, normalized_weights):
        return normalized_weights / self.openai_coeff if self.openai_coeff is not None else normalized_weights

    def get_run_losses(self, run_index):
        if self.single_run_debug:
            run_index = 0
        return torch.tensor(self.run_jsons[run_index]["metadata"][self.train_metric])

    def get_run_network(self, run_index, iter=0, normalize=True, augment=False):
        if self.single_run_debug:
            run_index = 0
        row = torch.tensor([self.run_offset[run_index] + range(self.num_checkpoints[run_index])[iter]], device=self.device)
        perms = P.draw_permutations(self.arch, [None]) if (augment and self.use_augment) else None
        return self._gather(row, row, normalize, perms)[0][0]

    # ---- pair sampling (host; the reference's draws) and the batch fetch (device)
    def sample_pairs(self, run_indices):
        """Per item, in order: step spacing, first index (torch.randint on self.generator, dataset_lmdb.py:195-207).
        -> (rows_0, rows_1, spacings, run indices) as int64 numpy arrays."""
        n = len(run_indices)
        r0, r1, sp_out, ri = (np.empty(n, dtype=np.int64) for _ in range(4))
        for k, run_index in enumerate(run_indices):
            run_index = 0 if self.single_run_debug else int(run_index) % self.num_runs
            nc = self.num_checkpoints[run_index]
            max_sp = (nc - 1) if self.max_step_spacing is None else self.max_step_spacing
            sp = torch.randint(low=self.min_step_spacing, high=max_sp + 1, size=(), generator=self.generator).item()
            i0 = torch.randint(low=0, high=nc - sp, size=(), generator=self.generator).item()
            if self.single_run_debug:
                i0 = 0
            r0[k], r1[k] = self.run_offset[run_index] + i0, self.run_offset[run_index] + i0 + sp
            sp_out[k], ri[k] = sp, run_index
        return r0, r1, sp_out, ri

    def sample_pairs_fast(self, run_indices, rng):
        """The same distribution drawn for the whole batch at once from a numpy Generator (not the reference's RNG
        stream): for loaders that only need i.i.d. pairs, ~50x less host time per batch than the item-by-item draws."""
        ri = np.zeros(len(run_indices), dtype=np.int64) if self.single_run_debug else \
            np.asarray(run_indices, dtype=np.int64) % self.num_runs
        nc = np.asarray(self.num_checkpoints, dtype=np.int64)[ri]
        max_sp = nc - 1 if self.max_step_spacing is None else np.full_like(nc, self.max_step_spacing)
        sp = rng.integers(self.min_step_spacing, max_sp + 1)
        i0 = np.zeros_like(sp) if self.single_run_debug else rng.integers(0, nc - sp)
        r0 = self.run_offset[ri] + i0
        return r0, r0 + sp, sp, ri

    def _gather(self, rows0, rows1, normalize=True, perms=None):
        """perms: int32 [n, perm_len] hidden-unit permutations already drawn (permute.draw_permutations) or None."""
        n = rows0.numel()
        index = P.build_index(self.arch, perms, self.device) if perms is not None else None
        out0 = torch.empty(n, self.D, dtype=torch.float32, device=self.device)
        out1 = torch.empty(n, self.D, dtype=torch.float32, device=self.device)
        scale = self.openai_coeff if (normalize and self.openai_coeff is not None) else 1.0
        L.check(L.lib().gpt_pair_gather(L.ptr(self.store), self.D, self.store.shape[0], L.ptr(rows0), L.ptr(rows1), L.ptr(index),
                                        scale, L.ptr(out0), L.ptr(out1), n, self.D, L.stream()), "gpt_pair_gather")
        return out0, out1

    def get_batch(self, run_indices, rng=None):
        """The collated batch the reference's DataLoader yields for these dataset indices.  With augmentation the RNG is
        consumed exactly as by consecutive __getitem__ calls: per item spacing, first index, then the hidden-unit
        permutation(s) from the GLOBAL torch RNG (aug_fn(..., seed=None), dataset_lmdb.py:225).
        rng: a numpy Generator -> the vectorised draws of sample_pairs_fast (same distribution, its own stream)."""
        if rng is not None:
            r0, r1, sp, ri = self.sample_pairs_fast(run_indices, rng)
            perms = None
            if self.use_augment:
                widths = [self.arch["hidden"]] if self.arch["kind"] == "mlp" else [self.arch["c1"], self.arch["c2"]]
                perms = torch.from_numpy(np.concatenate([np.argsort(rng.random((len(ri), w)), axis=1) for w in widths],
                                                        axis=1).astype(np.int32))
        elif not self.use_augment:
            r0, r1, sp, ri = self.sample_pairs(run_indices)
            perms = None
        else:
            parts, drawn = [], []
            for idx in run_indices:
                parts.append(self.sample_pairs([idx]))
                drawn.append(P.draw_permutations(self.arch, [None])[0])
            r0, r1, sp, ri = (np.concatenate([p[k] for p in parts]) for k in range(4))
            perms = torch.stack(drawn)
        rows0 = torch.from_numpy(r0).to(self.device, non_blocking=True)
        rows1 = torch.from_numpy(r1).to(self.device, non_blocking=True)
        p0, p1 = self._gather(rows0, rows1, True, perms)
        m = self.train_metric
        names = self.checkpoint_names
        loc0, loc1 = r0 - self.run_offset[ri], r1 - self.run_offset[ri]
        return {"parameters_0": p0, "parameters_1": p1,
                "checkpoint_key_0": [names[r][i] for r, i in zip(ri, loc0)],
                "checkpoint_key_1": [names[r][i] for r, i in zip(ri, loc1)],
                "run_name": [self.runs[r] for r in ri], "step_spacing": torch.from_numpy(sp),
                f"{m}_0": torch.from_numpy(self.metric[r0]), f"{m}_1": torch.from_numpy(self.metric[r1])}

    def __getitem__(self, run_index):
        """One sample with ParameterDataset.__getitem__'s keys (the batch path above is the fast one)."""
        b = self.get_batch([run_index])
        m = self.train_metric
        return {"parameters_0": b["parameters_0"][0], "parameters_1": b["parameters_1"][0],
                "checkpoint_key_0": b["checkpoint_key_0"][0], "checkpoint_key_1": b["checkpoint_key_1"][0],
                "run_name": b["run_name"][0], "step_spacing": int(b["step_spacing"][0]),
                f"{m}_0": float(b[f"{m}_0"][0]), f"{m}_1": float(b[f"{m}_1"][0])}

    def batches(self, batch_size, seed=0, rank=0, world=1, drop_last=True, fast=False):
        """One epoch of shuffled dataset indices in this rank's share (RandomSampler for one GPU, DistributedSampler's
        strided share of one permutation otherwise: Gpt/utils.py:80-91) -> batches as get_batch returns them."""
        g = torch.Generator().manual_seed(int(seed))
        order = torch.randperm(self.epoch_size, generator=g)
        if world > 1:
            per = (self.epoch_size + world - 1) // world
            order = torch.cat([order, order[:per * world - self.epoch_size]])[rank::world]
        stop = order.numel() - (order.numel() % batch_size if drop_last else 0)
        rng = np.random.default_rng([int(seed), int(rank)]) if fast else None
        for i in range(0, stop, batch_size):
            yield self.get_batch(order[i:i + batch_size].tolist(), rng)
