"""torch.autograd bridges for the training path (loss -> backward through the HIP kernels)."""
import torch

from . import _lib as L


class _MseRows(torch.autograd.Function):
    """mean_flat((target - out)^2) (Gpt/diffusion/diffusion_utils.py:8-12 over gaussian_diffusion.py:773)
    as one row-reduction kernel forward and one elementwise kernel backward."""

    @staticmethod
    def forward(ctx, target, out):
        B = out.shape[0]
        o = out.detach().float().contiguous().view(B, -1)
        tg = target.view(B, -1)
        mse = torch.empty(B, dtype=torch.float32, device=o.device)
        L.check(L.lib().gpt_mse_rows(L.ptr(tg), L.ptr(o), L.ptr(mse), B, o.shape[1], L.stream()), "gpt_mse_rows")
        ctx.save_for_backward(tg, o)
        ctx.shape = out.shape
        return mse

    @staticmethod
    def backward(ctx, g):
        tg, o = ctx.saved_tensors
        B, D = o.shape
        grad = torch.empty_like(o)
        L.check(L.lib().gpt_mse_rows_bwd(L.ptr(tg), L.ptr(o), L.ptr(g.float().contiguous()), L.ptr(grad), B, D,
                                         L.stream()), "gpt_mse_rows_bwd")
        return None, grad.view(ctx.shape)


def mse_rows(target, out):
    return _MseRows.apply(target, out)


def denoise_with_grad(model, x, t, lt, lp, x_prev):
    from .backward import DenoiseFunction
    return DenoiseFunction.run(model, x, t, lt, lp, x_prev)
