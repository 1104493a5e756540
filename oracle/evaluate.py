"""TEST INFRASTRUCTURE ONLY - CPU restatement of the sampled-network quality metric.

moduleify (Gpt/vis.py:539-570) loads a flat parameter vector into the task net in state_dict order
(fc1.weight, fc1.bias, fc2.weight, fc2.bias for data_gen/train_mnist.py:58-71) and batch_test_fn (vis.py:398-413) calls
the task's test_epoch (train_mnist.py:324-332) once per net.  Pinned by tests/golden/g10_mlp_eval.npz.
The CIFAR-10 ConvNet (data_gen/train_cifar10.py:53-68, test_epoch :426-434) likewise; pinned by g12_cnn_eval.npz."""
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


CNN_SHAPES = (("conv1.weight", (16, 3, 3, 3)), ("conv2.weight", (32, 16, 3, 3)), ("fc.weight", (10, 32)), ("fc.bias", (10,)))


def cnn_eval(nets, imgs, labels):
    """(loss [B], err% [B]) of B flat CIFAR-10 ConvNet vectors (state_dict order of train_cifar10.py:55-60) on imgs
    [N, 3, 32, 32]: relu(conv1 s2 p1) -> relu(conv2 s1 p1) -> global average pool -> fc (train_cifar10.py:62-68)."""
    assert nets.shape[1] == 5370
    x = imgs.float()
    losses, errs = [], []
    for v in nets:
        w, off = {}, 0
        for name, shape in CNN_SHAPES:
            n = 1
            for d in shape:
                n *= d
            w[name] = v[off:off + n].view(shape)
            off += n
        h = F.relu(F.conv2d(x, w["conv1.weight"], stride=2, padding=1))
        h = F.relu(F.conv2d(h, w["conv2.weight"], stride=1, padding=1))
        logits = F.linear(h.mean(dim=(2, 3)), w["fc.weight"], w["fc.bias"])
        losses.append(F.cross_entropy(logits, labels, reduction="none").mean())
        errs.append((logits.argmax(1) != labels).float().mean() * 100)
    return torch.stack(losses), torch.stack(errs)


# ------------------------------------------------------------------------------------------ prompting chains
def batch_linspace(start, end, steps):
    """Gpt/vis.py:416-424, as written: one torch.linspace per row, from Python floats."""
    return torch.stack([torch.linspace(s.item(), e.item(), steps=steps) for s, e in zip(start, end)], 0)


def one_step_prompting(task_net, synth_fn, test_fn, best_prompt, minimize=True, dvo_steps=20, prompt_start_coeff=1.0):
    """Gpt/vis.py:359-395 (regularly sampled prompts).  synth_fn(desired, current, nets) -> evolved nets;
    test_fn(nets [N, D]) -> losses [N] (the per-net batch_test_fn loop, vis.py:398-413, folded into test_fn)."""
    n = task_net.size(0)
    base = test_fn(task_net).float()
    desired = batch_linspace(torch.full((n,), float(best_prompt)), base * prompt_start_coeff, dvo_steps)
    if not minimize:
        desired = desired.flip(1,)
    evolved = synth_fn(desired.flatten(), base.repeat_interleave(dvo_steps, dim=0), task_net.repeat_interleave(dvo_steps, dim=0))
    observed = torch.cat([base.view(n, 1), test_fn(evolved).float().view(n, dvo_steps)], 1)
    return desired, observed, evolved


def recursive_prompting(task_net, synth_fn, test_fn, recursive_prompt, best_prompt=None, minimize=True, interp_alg="zero",
                        max_steps=2, step_divider=5):
    """Gpt/vis.py:325-356."""
    n = task_net.size(0)
    cur_loss, cur_net = test_fn(task_net).float(), task_net
    desired, observed, nets = [], [cur_loss], []
    sign = -1 if minimize else 1
    for _ in range(max_steps):
        if interp_alg == "zero":
            want = torch.full((n,), float(recursive_prompt))
        elif interp_alg == "step":
            want = cur_loss + sign * (best_prompt / float(step_divider))
        else:
            raise NotImplementedError
        desired.append(want)
        cur_net = synth_fn(want, cur_loss, cur_net)
        cur_loss = test_fn(cur_net).float()
        observed.append(cur_loss)
        nets.append(cur_net)
    return torch.stack(desired, 1), torch.stack(observed, 1), nets
