"""Oracle restatement of the permutation-augmentation index maps (TEST INFRASTRUCTURE, see
oracle/__init__.py).  Integer work: parity bar is BIT-EXACT.

Follows Gpt/data/augment.py:25-75 (random_permute_flat: one flat gather index, built block by
block in state_dict order) with the task permutation functions
data_gen/train_mnist.py:28-50 (random_permute_mlp), data_gen/train_cifar10.py:27-47
(random_permute_cnn) and data_gen/train_rl.py:51-74 (random_permute_actor).  Hidden-unit permutations come from torch's CPU ``randperm`` on a
``torch.Generator().manual_seed(seed)`` (augment.py:40-43), so the oracle draws them the same way.
"""
import numpy as np
import torch


def _randperm(n, gen):
    return torch.randperm(n, generator=gen).numpy().astype(np.int64)


def draw_mlp(seed, hidden=10):
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    return (_randperm(hidden, gen),)


def draw_cnn(seed, ws=(16, 32)):
    """conv1's permutation is drawn BEFORE conv2's (train_cifar10.py:39,42)."""
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    p1 = _randperm(ws[0], gen)
    p2 = _randperm(ws[1], gen)
    return p1, p2


def mlp_index(p, n_in=784, n_out=10):
    """MLP fc1 (H,n_in), fc1.b (H), fc2 (n_out,H), fc2.b (n_out): rows of fc1/fc1.b permuted
    ('out'), columns of fc2 permuted ('in'), fc2.b registered unchanged."""
    H = len(p)
    i = np.arange(H)[:, None]
    j = np.arange(n_in)[None, :]
    fc1 = (p[i] * n_in + j).reshape(-1)
    off = H * n_in
    fc1b = off + p
    off += H
    fc2 = (off + np.arange(n_out)[:, None] * H + p[None, :]).reshape(-1)
    off += n_out * H
    fc2b = off + np.arange(n_out)
    return np.concatenate([fc1, fc1b, fc2, fc2b]).astype(np.int64)


def cnn_index(p1, p2, c_in=3, k=3, n_out=10):
    """ConvNet conv1 (C1,c_in,k,k) 'out'; conv2 (C2,C1,k,k) 'both' = [:, p1][p2];
    fc (n_out,C2) 'in'; fc.b registered."""
    C1, C2 = len(p1), len(p2)
    kk = k * k
    r = np.arange(c_in * kk)[None, :]
    conv1 = (p1[:, None] * (c_in * kk) + r).reshape(-1)
    off = C1 * c_in * kk
    o = p2[:, None, None]
    i = p1[None, :, None]
    q = np.arange(kk)[None, None, :]
    conv2 = (off + o * (C1 * kk) + i * kk + q).reshape(-1)
    off += C2 * C1 * kk
    fc = (off + np.arange(n_out)[:, None] * C2 + p2[None, :]).reshape(-1)
    off += n_out * C2
    fcb = off + np.arange(n_out)
    return np.concatenate([conv1, conv2, fc, fcb]).astype(np.int64)


def draw_actor(seed, hidden=(32, 32)):
    """One permutation per hidden layer of the policy, drawn input layer first (train_rl.py:64-73)."""
    gen = torch.Generator().manual_seed(seed) if seed is not None else None
    return tuple(_randperm(h, gen) for h in hidden)


def actor_index(perms, obs=4, act=1):
    """VisActorCritic (train_rl.py:95-121): log_std (act) registered unchanged, then the actor's Linears in state_dict order:
    input layer weight + bias 'out' (rows permuted), hidden layers weight 'both' = [:, p_prev][p_new] and bias 'out',
    output layer weight 'in' (columns permuted by the last hidden permutation), bias registered (train_rl.py:61-73)."""
    dims = [obs] + [len(p) for p in perms] + [act]
    parts, off = [np.arange(act)], act
    for l in range(len(dims) - 1):
        n_in, n_out = dims[l], dims[l + 1]
        pin = perms[l - 1] if l > 0 else np.arange(n_in)
        pout = perms[l] if l < len(perms) else np.arange(n_out)
        parts.append((off + pout[:, None] * n_in + pin[None, :]).reshape(-1))
        off += n_out * n_in
        parts.append(off + pout)
        off += n_out
    return np.concatenate(parts).astype(np.int64)


def permute(nets, index):
    """augment.py:72: the same gather is applied to every net of a pair."""
    return [n[index] for n in nets]
