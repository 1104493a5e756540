"""Batched evaluation of sampled task networks and the sampling harness around p_sample_loop.

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
