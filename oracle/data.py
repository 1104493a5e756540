"""TEST INFRASTRUCTURE ONLY - CPU restatement of the checkpoint-pair dataset (Gpt/data/dataset_lmdb.py).

Follows ParameterDataset.__post_init__ (:41-122: run discovery from the directory names, train/test split, runs.json,
parameter sizes from the "<first checkpoint>_arch" entry, metadata reductions, epoch size), __getitem__ (:184-240: step
spacing, then the first index, from torch.randint on the dataset's generator; normalise; permutation augmentation on both
checkpoints with the GLOBAL torch RNG, seed=None) and OpenAINormalization (Gpt/data/normalization.py:13-37).  The LMDB
behind Gpt/data/database.py:51-57 is replaced by any mapping key -> fp32 tensor (py-lmdb is not installed in this image);
pinned by tests/golden/g13_pairs.npz, produced by the reference class over the same synthetic store."""
import json
import os
from glob import glob

import torch

from . import permute as op


class RefPairDataset:
    def __init__(self, dataset_dir, open_db, split="train", max_train_runs=1000000, num_test_runs=500,
                 target_epoch_size=806400, train_metric="avg_test_loss", min_step_spacing=1, max_step_spacing=None,
                 normalizer_name="openai", openai_coeff=4.185, min_val=None, max_val=None, augment=None, minimize=True,
                 verify_done=False):
        lmdb_paths = sorted(glob(f"{dataset_dir}/*"))                               # :48
        if not lmdb_paths:
            raise FileNotFoundError
        if verify_done:
            lmdb_paths = [p for p in lmdb_paths if os.path.exists(os.path.join(p, "DONE.txt"))]
        run_path = {}
        for p in lmdb_paths:                                                        # :58-61
            for run in os.path.basename(p).split("_"):
                run_path[run] = p
        runs = list(run_path.keys())
        assert split in ("train", "test") and num_test_runs < len(runs)
        if split == "train":                                                        # :67-71
            runs = runs[:-num_test_runs] if num_test_runs > 0 else runs
            runs = runs[:max_train_runs]
        else:
            runs = runs[-num_test_runs:] if num_test_runs > 0 else []
        self.runs, self.run_path = runs, {r: run_path[r] for r in runs}
        self.run_jsons = []
        for run in runs:                                                            # :79-87
            with open(os.path.join(self.run_path[run], "runs.json")) as f:
                data = json.load(f)
            self.run_jsons.append(data[run] if run in data else data)
        self.dbs = {p: open_db(p) for p in set(self.run_path.values())}
        first = list(self.run_jsons[0]["checkpoints"].keys())[0]
        self.parameter_sizes = self.db(0)[f"{first}_arch"].long().tolist()          # :97-98
        self.num_checkpoints = [len(j["checkpoints"]) for j in self.run_jsons]
        reduce_fn = min if minimize else max
        self.optimal_test_loss = reduce_fn(j["metadata"][f"optimal_{train_metric}".replace("_avg", "")] for j in self.run_jsons)
        if min_val is None and max_val is None:                                     # :142-147
            min_val = min(j["metadata"]["min_parameter_val"] for j in self.run_jsons)
            max_val = max(j["metadata"]["max_parameter_val"] for j in self.run_jsons)
        self.min_val, self.max_val = min_val, max_val
        self.coeff = openai_coeff if normalizer_name == "openai" else None
        self.augment = augment                                                      # None | "mlp" | "cnn"
        self.train_metric, self.min_step_spacing, self.max_step_spacing = train_metric, min_step_spacing, max_step_spacing
        assert target_epoch_size >= len(runs)
        self.epoch_size = len(runs) * (target_epoch_size // len(runs))              # :122
        self.generator = None

    def db(self, run_index):
        return self.dbs[self.run_path[self.runs[run_index]]]

    def normalize(self, x):
        return x * self.coeff if self.coeff is not None else x

    def get_range(self):
        if self.coeff is None:
            return self.min_val, self.max_val
        return self.min_val * self.coeff, self.max_val * self.coeff

    def __len__(self):
        return self.epoch_size

    def __getitem__(self, run_index):
        run_index = run_index % len(self.runs)
        ckpts = self.run_jsons[run_index]["checkpoints"]
        names = list(ckpts.keys())
        n = self.num_checkpoints[run_index]
        max_sp = (n - 1) if self.max_step_spacing is None else self.max_step_spacing
        sp = torch.randint(low=self.min_step_spacing, high=max_sp + 1, size=(), generator=self.generator).item()   # :195-200
        i0 = torch.randint(low=0, high=n - sp, size=(), generator=self.generator).item()                           # :202-207
        k0, k1 = names[i0], names[i0 + sp]
        p0, p1 = self.normalize(self.db(run_index)[k0]), self.normalize(self.db(run_index)[k1])
        if self.augment == "mlp":                                                   # aug_fn(..., seed=None): global RNG
            idx = torch.from_numpy(op.mlp_index(*op.draw_mlp(None)))
            p0, p1 = p0[idx], p1[idx]
        elif self.augment == "cnn":
            idx = torch.from_numpy(op.cnn_index(*op.draw_cnn(None)))
            p0, p1 = p0[idx], p1[idx]
        m = self.train_metric
        return {"parameters_0": p0, "parameters_1": p1, "checkpoint_key_0": k0, "checkpoint_key_1": k1,
                "run_name": self.runs[run_index], "step_spacing": sp, f"{m}_0": ckpts[k0][m], f"{m}_1": ckpts[k1][m]}
