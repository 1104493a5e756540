"""On-device permutation augmentation (reference: Gpt/data/augment.py:25-75 with
data_gen/train_mnist.py:28-50, data_gen/train_cifar10.py:27-47 and data_gen/train_rl.py:51-74).

The hidden-unit permutations are drawn on the host exactly as the reference draws them
(torch.randperm on torch.Generator().manual_seed(seed), conv1 before conv2), so the flat index maps are
bit-identical for the same seed; the maps themselves and the gather run on the GPU for whole batches."""
import ctypes as C

import torch

from . import _lib as L

MLP_ARCH = dict(kind="mlp", n_in=784, hidden=10, n_out=10)            # data_gen/train_mnist.py:58-63
CNN_ARCH = dict(kind="cnn", c_in=3, c1=16, c2=32, ksz=3, n_out=10)     # data_gen/train_cifar10.py:55-60
# Cartpole policy: VisActorCritic("Cartpole", 4, 1), pi_hid_sizes [32, 32] (data_gen/train_rl.py:95-121, Gpt/tasks.py:75)
ACTOR_ARCH = dict(kind="chain", prefix=1, dims=[4, 32, 32, 1])


def arch_dim(arch):
    if arch["kind"] == "chain":
        d = arch["dims"]
        return arch["prefix"] + sum(d[l + 1] * d[l] + d[l + 1] for l in range(len(d) - 1))
    if arch["kind"] == "mlp":
        return arch["hidden"] * arch["n_in"] + arch["hidden"] + arch["n_out"] * arch["hidden"] + arch["n_out"]
    kk = arch["ksz"] ** 2
    return arch["c1"] * arch["c_in"] * kk + arch["c2"] * arch["c1"] * kk + arch["n_out"] * arch["c2"] + arch["n_out"]


def draw_permutations(arch, seeds):
    """int32 host tensor [n, perm_len]: per sample the hidden-unit permutations back to back.
    seeds: iterable of ints (or None entries: the global torch RNG, as augment.py:40-43)."""
    widths = ([arch["hidden"]] if arch["kind"] == "mlp" else arch["dims"][1:-1] if arch["kind"] == "chain"
              else [arch["c1"], arch["c2"]])
    rows = []
    for seed in seeds:
        gen = torch.Generator().manual_seed(int(seed)) if seed is not None else None
        rows.append(torch.cat([torch.randperm(w, generator=gen) for w in widths]))
    return torch.stack(rows).to(torch.int32)


def build_index(arch, perms, device):
    """int64 [n, D] flat gather maps on `device` from the drawn permutations."""
    perms = perms.to(device=device, dtype=torch.int32).contiguous()
    n, D = perms.shape[0], arch_dim(arch)
    index = torch.empty(n, D, dtype=torch.int64, device=device)
    if arch["kind"] == "chain":
        dims = (C.c_int * len(arch["dims"]))(*arch["dims"])
        st = L.lib().gpt_perm_index_chain(L.ptr(perms), L.ptr(index), n, arch["prefix"], len(arch["dims"]), dims, L.stream())
    elif arch["kind"] == "mlp":
        st = L.lib().gpt_perm_index_mlp(L.ptr(perms), L.ptr(index), n, arch["n_in"], arch["hidden"], arch["n_out"],
                                        L.stream())
    else:
        st = L.lib().gpt_perm_index_cnn(L.ptr(perms), L.ptr(index), n, arch["c_in"], arch["c1"], arch["c2"],
                                        arch["ksz"], arch["n_out"], L.stream())
    L.check(st, "gpt_perm_index")
    return index


def gather(nets, index):
    """Applies the same per-sample index to every tensor of `nets` (each (n, D) fp32 on the GPU)."""
    nets = [t.contiguous() for t in nets]
    n, D = index.shape
    outs = [torch.empty_like(t) for t in nets]
    k = len(nets)
    src = (C.c_void_p * k)(*[L.ptr(t) for t in nets])
    dst = (C.c_void_p * k)(*[L.ptr(t) for t in outs])
    L.check(L.lib().gpt_perm_gather(src, dst, k, L.ptr(index), n, D, L.stream()), "gpt_perm_gather")
    return outs


def random_permute_batch(nets, arch, seeds):
    """Batch form of random_permute_flat: nets = list of (n, D) GPU tensors (e.g. both checkpoints of each
    pair); one function-preserving permutation per sample, shared by all tensors of that sample."""
    index = build_index(arch, draw_permutations(arch, seeds), nets[0].device)
    return gather(nets, index)
