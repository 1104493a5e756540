"""Batched evaluation of sampled task networks, the sampling harness around p_sample_loop and the two evaluation chains
built on it (one-step and recursive prompting, Gpt/vis.py:325-395, BASELINE.json configs[2]).

`synth` / `create_thresholding_fn` keep the signatures of Gpt/vis.py:190-252.  `mlp_test_losses` replaces
moduleify + batch_test_fn + test_epoch (vis.py:398-413, 539-570; data_gen/train_mnist.py:324-332) for the MNIST MLP:
all B sampled nets are scored in two launches (one exact-fp32 MFMA GEMM for fc1 of every net, one fused head kernel)
instead of B Python iterations.  `cnn_test_losses` does the same for the CIFAR-10 ConvNet (data_gen/train_cifar10.py:53-68,
426-434) in one fused kernel (both convolutions, pooling, fc and the loss; conv2 on fp32 MFMAs)."""
import torch

from . import _lib as L


def create_thresholding_fn(thresholding, param_range):
    """'none': identity; 'static': clamp to param_range after every denoising step (vis.py:190-207)."""
    from .ddpm import StaticThreshold
    if thresholding == "none":
        return lambda x: x
    if thresholding == "static":
        return StaticThreshold(param_range[0], param_range[1])       # fused into the posterior kernel
    raise NotImplementedError(thresholding)


def synth(diffusion, G, loss_target, loss_prev, w_prev, clip_denoised=False, param_range=None, thresholding="none",
          **p_sample_loop_kwargs):
    """A sample from p(theta* | prompted loss, starting loss, starting theta) (vis.py:210-252)."""
    assert loss_target.size(0) == loss_prev.size(0) == w_prev.size(0)
    denoised_fn = None
    if param_range is not None:
        assert param_range[0] < param_range[1]
        denoised_fn = create_thresholding_fn(thresholding, param_range)
        clip_denoised = False
    model_kwargs = {"loss_target": loss_target, "loss_prev": loss_prev, "x_prev": w_prev}
    return diffusion.p_sample_loop(G, w_prev.shape, clip_denoised=clip_denoised, model_kwargs=model_kwargs,
                                   device=w_prev.device, denoised_fn=denoised_fn, **p_sample_loop_kwargs)


def mlp_layout(in_dim=784, hidden=10, classes=10):
    off_b1 = hidden * in_dim
    off_w2 = off_b1 + hidden
    off_b2 = off_w2 + classes * hidden
    return off_b1, off_w2, off_b2, off_b2 + classes


@torch.no_grad()
def mlp_test_losses(nets, images, labels, hidden=10, classes=10):
    """(loss [B], err% [B]): mean cross-entropy and top-1 error of B flat MLP parameter vectors (state_dict order
    fc1.weight, fc1.bias, fc2.weight, fc2.bias) on `images` [N, ...] / `labels` [N].  Un-normalise the vectors first
    (OpenAINormalization.unnormalize, Gpt/data/normalization.py:27-29) if they come straight from G.pt."""
    lib = L.lib()
    if not nets.is_cuda:
        raise L.GptHipError("mlp_test_losses: the HIP path needs GPU tensors; there is no CPU fallback")
    dev = nets.device
    x = torch.flatten(images, 1).to(dev, torch.float32).contiguous()
    n_img, in_dim = x.shape
    off_b1, off_w2, off_b2, D = mlp_layout(in_dim, hidden, classes)
    assert nets.dim() == 2 and nets.shape[1] == D, f"expected (B, {D}) parameter vectors"
    nets = nets.float().contiguous()
    B = nets.shape[0]
    labels = labels.to(dev, torch.long).contiguous()
    assert labels.numel() == n_img
    # K and N granularity of gpt_gemm_nt (fp32): pad the feature dim to 16 and the image count to 8 with zeros
    kp, np_ = (in_dim + 15) // 16 * 16, (n_img + 7) // 8 * 8
    w1 = torch.zeros(B * hidden, kp, device=dev)
    w1[:, :in_dim] = nets[:, :off_b1].reshape(B * hidden, in_dim)
    xp = x
    if kp != in_dim or np_ != n_img:
        xp = torch.zeros(np_, kp, device=dev)
        xp[:n_img, :in_dim] = x
    H = torch.empty(B * hidden, np_, device=dev)
    L.check(lib.gpt_gemm_nt(L.F32, L.ptr(w1), kp, L.ptr(xp), kp, None, L.ptr(H), np_, None, 0, B * hidden, np_, kp,
                            L.EPI_STORE_F32, L.stream()), "gpt_gemm_nt")
    loss = torch.empty(B, device=dev)
    err = torch.empty(B, device=dev)
    L.check(lib.gpt_mlp_head(L.ptr(H), np_, L.ptr(nets), D, off_b1, off_w2, off_b2, L.ptr(labels), L.ptr(loss), L.ptr(err),
                             B, hidden, classes, n_img, L.stream()), "gpt_mlp_head")
    return loss, err


CNN_D = 5370          # conv1.weight (16,3,3,3) | conv2.weight (32,16,3,3) | fc.weight (10,32) | fc.bias (10)


@torch.no_grad()
def cnn_test_losses(nets, images, labels):
    """(loss [B], err% [B]): mean cross-entropy and top-1 error of B flat CIFAR-10 ConvNet parameter vectors (state_dict
    order, train_cifar10.py:55-60) on `images` [N, 3, 32, 32] (already colour-normalised, :158-160) / `labels` [N]."""
    lib = L.lib()
    if not nets.is_cuda:
        raise L.GptHipError("cnn_test_losses: the HIP path needs GPU tensors; there is no CPU fallback")
    dev = nets.device
    assert nets.dim() == 2 and nets.shape[1] == CNN_D, f"expected (B, {CNN_D}) parameter vectors"
    assert images.dim() == 4 and tuple(images.shape[1:]) == (3, 32, 32), "expected (N, 3, 32, 32) images"
    x = images.to(dev, torch.float32).contiguous()
    nets = nets.float().contiguous()
    labels = labels.to(dev, torch.long).contiguous()
    B, n_img = nets.shape[0], x.shape[0]
    assert labels.numel() == n_img and n_img > 0
    assert int(labels.min()) >= 0 and int(labels.max()) < 10
    groups = (B + 3) // 4                     # one workgroup = 4 nets x one chunk of the images
    chunks = max(1, min((n_img + 7) // 8, (1024 + groups - 1) // groups))
    partial = torch.empty(B, chunks, 2, device=dev)
    loss = torch.empty(B, device=dev)
    err = torch.empty(B, device=dev)
    L.check(lib.gpt_cnn_eval(L.ptr(x), L.ptr(labels), L.ptr(nets), CNN_D, L.ptr(partial), chunks, L.ptr(loss), L.ptr(err),
                             B, n_img, L.stream()), "gpt_cnn_eval")
    return loss, err


# ------------------------------------------------------------------------------------------ prompting chains (vis.py:269-431)
TASK_PROMPTS = {     # Gpt/tasks.py:36-82: minimize / best_prompt / recursive_prompt of the tasks whose nets this path scores
    "mnist_loss": dict(minimize=True, best_prompt=0.0, recursive_prompt=0.0),
    "mnist_error": dict(minimize=True, best_prompt=5.0, recursive_prompt=5.0),
    "cifar10_loss": dict(minimize=True, best_prompt=1.2, recursive_prompt=1.4),
    "cifar10_error": dict(minimize=True, best_prompt=35.0, recursive_prompt=45.0),
}


def batched_test_fn(kind, images, labels, metric="loss", unnormalize=None):
    """A test_fn for the chains below that scores ALL nets of a batch at once on the GPU (mlp_test_losses / cnn_test_losses)
    instead of the reference's one-net-at-a-time moduleify + test_epoch loop (vis.py:398-413, 539-570).
    kind: 'mlp' | 'cnn'; metric: 'loss' | 'error' (test_epoch returns (loss, error): Gpt/tasks.py:38,47);
    unnormalize: openai_coeff, samples are divided by it first (Gpt/data/normalization.py:27-29)."""
    scorer = {"mlp": mlp_test_losses, "cnn": cnn_test_losses}[kind]
    which = {"loss": 0, "error": 1}[metric]

    def test_fn(nets):
        if unnormalize is not None:
            nets = nets / unnormalize
        return scorer(nets, images, labels)[which]

    test_fn.batched = True
    return test_fn


def batch_test_fn(test_fn, net_batch, return_as_tensor=True, check_dims=True, device=None):
    """vis.py:398-413.  A test_fn marked `.batched` (batched_test_fn above) gets the whole (N, D) batch in one call; any
    other callable is applied net by net to (1, D) slices, as the reference does."""
    if check_dims:
        assert net_batch.dim() == 2
    if getattr(test_fn, "batched", False):
        result = test_fn(net_batch).float()
        return result if return_as_tensor else result.tolist()
    result = [test_fn(net.unsqueeze(0)) for net in net_batch]
    if return_as_tensor:
        result = torch.tensor(result, dtype=torch.float, device=device if device is not None else net_batch.device)
    return result


def batch_linspace(start, end, steps, device=None):
    """vis.py:416-424: row i = torch.linspace(start[i], end[i], steps) in fp32, for all rows in one expression.
    torch.linspace's element formula is kept (step = (end - start) / (steps - 1); the first half counts up from start, the
    second half down from end): rows equal the reference's per-row calls to the last ulp (torch.linspace itself differs by
    an ulp between its vectorised CPU kernel and its GPU kernel, so nothing stricter is defined)."""
    assert start.dim() == end.dim() == 1 and start.size(0) == end.size(0)
    start, end = start.float(), end.float()
    if device is not None:
        start, end = start.to(device), end.to(device)
    if steps == 1:
        return start.view(-1, 1).clone()
    step = ((end - start) / float(steps - 1)).view(-1, 1)
    idx = torch.arange(steps, device=start.device)
    up = start.view(-1, 1) + step * idx.float()
    down = end.view(-1, 1) - step * (steps - 1 - idx).float()
    return torch.where(idx.view(1, -1) < steps // 2, up, down)


def scalar_expand(scalar, num, device):
    """vis.py:427-431 (the reference hard-codes 'cuda')."""
    return torch.tensor(scalar, dtype=torch.float, device=device).view(1).repeat(num)


@torch.inference_mode()
def one_step_prompting(task_net, synth_fn, diffusion, G, test_fn, metadata, tasknet_index=0, device=None, dvo_steps=20,
                       desired_losses=None, prompt_start_coeff=1.0, **ignore):
    """One update sampled from G.pt per (net, prompt) pair (vis.py:359-395): every net is repeated dvo_steps times
    (repeat_interleave) with prompts regularly spaced between best_prompt and its current loss, so the sampler sees a batch
    of N * dvo_steps rows (16 x 20 = 320 MNIST, 64 x 20 = 1280 CIFAR-10: configs/test/*.yaml).  Returns
    (desired [N, dvo_steps], observed [N, 1 + dvo_steps])."""
    device = task_net.device if device is None else device
    num_nets = task_net.size(0)
    base_loss = batch_test_fn(test_fn, task_net, device=device)
    if desired_losses is None:
        start_loss = base_loss
        best_prompt = scalar_expand(metadata["best_prompt"], num_nets, device)
        desired_losses = batch_linspace(best_prompt, base_loss * prompt_start_coeff, steps=dvo_steps, device=device)
        if not metadata["minimize"]:
            desired_losses = desired_losses.flip(1,)
    else:
        desired_losses = desired_losses[tasknet_index: tasknet_index + num_nets]
        start_loss = desired_losses[:, 0].to(device)
        desired_losses = desired_losses[:, :dvo_steps].to(device)
    num_evos = desired_losses.size(1)
    desired_losses_in = desired_losses.flatten()
    current_losses = start_loss.repeat_interleave(num_evos, dim=0)
    nets_in = task_net.repeat_interleave(num_evos, dim=0)
    evolved_nets = synth_fn(diffusion, G, desired_losses_in, current_losses, nets_in)
    observed_losses = batch_test_fn(test_fn, evolved_nets, device=device).view(num_nets, num_evos)
    observed_losses = torch.cat([base_loss.view(num_nets, 1), observed_losses], 1)
    return desired_losses, observed_losses


@torch.inference_mode()
def recursive_prompting(task_net, synth_fn, diffusion, G, test_fn, metadata, tasknet_index=0, device=None, interp_alg="zero",
                        max_steps=2, step_divider=5, **ignore):
    """max_steps chained sampling loops (vis.py:325-356): each sample is scored and fed back as the next loop's starting
    parameters x_prev / starting loss.  Returns (desired [N, max_steps], observed [N, 1 + max_steps])."""
    device = task_net.device if device is None else device
    num_nets = task_net.size(0)
    current_loss = batch_test_fn(test_fn, task_net, device=device)
    current_net = task_net
    desired_losses, observed_losses = [], [current_loss]
    sign = -1 if metadata["minimize"] else 1
    for step in range(max_steps):
        if interp_alg == "zero":
            desired_loss = scalar_expand(metadata["recursive_prompt"], num_nets, device)
        elif interp_alg == "step":
            desired_loss = current_loss + sign * (metadata["best_prompt"] / float(step_divider))
        else:
            raise NotImplementedError
        desired_losses.append(desired_loss)
        current_net = synth_fn(diffusion, G, desired_loss, current_loss, current_net)
        current_loss = batch_test_fn(test_fn, current_net, device=device)
        observed_losses.append(current_loss)
    return torch.stack(desired_losses, 1), torch.stack(observed_losses, 1)


@torch.inference_mode()
def aggregated_evolve(evolve_fn, task_nets, net_mb_size, synth_fn, diffusion, G, test_fn, metadata, **evolve_kwargs):
    """vis.py:293-322: the nets of this rank in mini-batches of net_mb_size, then one all_gather of the two small result
    tables over the ranks (every rank holds its own nets, main.py:271: the sampling itself needs no traffic)."""
    from . import dp
    desired, observed, index = [], [], 0
    for batch in torch.split(task_nets, net_mb_size, dim=0):
        d, o = evolve_fn(batch, synth_fn, diffusion, G, test_fn, metadata, tasknet_index=index, **evolve_kwargs)
        desired.append(d)
        observed.append(o)
        index += batch.size(0)
    desired, observed = torch.cat(desired, 0), torch.cat(observed, 0)
    dp.synchronize()
    return dp.all_gather(desired, dim=0).cpu(), dp.all_gather(observed, dim=0).cpu()


@torch.inference_mode()
def probe_G(evolve_fn, task_nets, net_mb_size, synth_fn, diffusion, G, test_fn, metadata, interp_algs, **evolve_kwargs):
    """vis.py:269-290."""
    if len(interp_algs) == 0:
        return [], []
    desired, observed = [], []
    for alg in interp_algs:
        d, o = aggregated_evolve(evolve_fn, task_nets, net_mb_size, synth_fn, diffusion, G, test_fn, metadata, interp_alg=alg,
                                 **evolve_kwargs)
        desired.append(d)
        observed.append(o)
    return torch.stack(desired), torch.stack(observed)


def r_squared(preds, targets):
    """Prompt-alignment score (vis.py:255-266): R^2 per row, averaged over the batch."""
    assert preds.size() == targets.size() and preds.dim() > 1
    avg = targets.mean(dim=1, keepdim=True)
    var = (targets - avg).pow(2).sum(dim=1)
    err = (targets - preds).pow(2).sum(dim=1)
    return (1.0 - err / (var + 1e-8)).mean().item()
