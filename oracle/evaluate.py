"""TEST INFRASTRUCTURE ONLY - CPU restatement of the sampled-network quality metric.

moduleify (Gpt/vis.py:539-570) loads a flat parameter vector into the task net in state_dict order
(fc1.weight, fc1.bias, fc2.weight, fc2.bias for data_gen/train_mnist.py:58-71) and batch_test_fn (vis.py:398-413) calls
the task's test_epoch (train_mnist.py:324-332) once per net.  Pinned by tests/golden/g10_mlp_eval.npz."""
import torch
import torch.nn.functional as F


def mlp_layout(in_dim=784, hidden=10, classes=10):
    """Offsets of fc1.bias, fc2.weight, fc2.bias in the flat vector and its length D."""
    off_b1 = hidden * in_dim
    off_w2 = off_b1 + hidden
    off_b2 = off_w2 + classes * hidden
    return off_b1, off_w2, off_b2, off_b2 + classes


def mlp_eval(nets, imgs, labels, hidden=10, classes=10):
    """(loss [B], err% [B]) of B flat MNIST-MLP parameter vectors on one batch of images."""
    x = torch.flatten(imgs, 1).float()
    in_dim = x.shape[1]
    off_b1, off_w2, off_b2, D = mlp_layout(in_dim, hidden, classes)
    assert nets.shape[1] == D
    losses, errs = [], []
    for v in nets:
        w1, b1 = v[:off_b1].view(hidden, in_dim), v[off_b1:off_w2]
        w2, b2 = v[off_w2:off_b2].view(classes, hidden), v[off_b2:D]
        logits = F.linear(F.relu(F.linear(x, w1, b1)), w2, b2)
        losses.append(F.cross_entropy(logits, labels, reduction="none").mean())
        errs.append((logits.argmax(1) != labels).float().mean() * 100)
    return torch.stack(losses), torch.stack(errs)
