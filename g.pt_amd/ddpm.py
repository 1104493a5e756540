"""DDPM process for G.pt on MI355X: schedule tables on the host in float64, every per-step tensor
operation in fused HIP kernels (include/gpt_hip.h: gpt_q_sample, gpt_posterior_step, gpt_mse_rows,
gpt_sample_loop).

Keeps the reference's names and call signatures (Gpt/diffusion/__init__.py:5-43,
gaussian_diffusion.py:138-867, respace.py:8-125, timestep_sampler.py:23-64) so `main.py`, `Gpt/vis.py`
and `playground.py` call it unchanged.  Out of scope (NotImplementedError): learned variances and the
variational-bound losses, which G.pt never enables (main.py:216-219 builds learn_sigma=False).
"""
import ctypes as C
import enum
import math

import numpy as np
import torch

from . import _lib as L
from .denoiser import Gpt


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self in (LossType.KL, LossType.RESCALED_KL)


# ------------------------------------------------------------------------------------------ schedules
def get_beta_schedule(beta_schedule, *, beta_start, beta_end, num_diffusion_timesteps):
    """gaussian_diffusion.py:59-89."""
    n = num_diffusion_timesteps
    if beta_schedule == "quad":
        betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=np.float64) ** 2
    elif beta_schedule == "linear":
        betas = np.linspace(beta_start, beta_end, n, dtype=np.float64)
    elif beta_schedule in ("warmup10", "warmup50"):
        frac = 0.1 if beta_schedule == "warmup10" else 0.5
        betas = beta_end * np.ones(n, dtype=np.float64)
        w = int(n * frac)
        betas[:w] = np.linspace(beta_start, beta_end, w, dtype=np.float64)
    elif beta_schedule == "const":
        betas = beta_end * np.ones(n, dtype=np.float64)
    elif beta_schedule == "jsd":
        betas = 1.0 / np.linspace(n, 1, n, dtype=np.float64)
    else:
        raise NotImplementedError(beta_schedule)
    assert betas.shape == (n,)
    return betas


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:119-135."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    """gaussian_diffusion.py:92-116."""
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return get_beta_schedule("linear", beta_start=scale * 0.0001, beta_end=scale * 0.02,
                                 num_diffusion_timesteps=num_diffusion_timesteps)
    if schedule_name == "squaredcos_cap_v2":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """Retained timesteps of a respaced process (respace.py:8-58)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, taken = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            taken.append(start + round(pos))
            pos += stride
        start += size
    return set(taken)


class StaticThreshold:
    """denoised_fn that clamps the x0 prediction to [lo, hi] ('static' thresholding, Gpt/vis.py:190-207).
    Passing an instance (instead of an opaque closure) lets p_sample_loop fuse the clamp into the
    posterior-update kernel and keep the whole loop in one host call."""

    def __init__(self, lo, hi):
        self.lo, self.hi = float(lo), float(hi)

    def __call__(self, x):
        return torch.clamp(x, self.lo, self.hi)


def _unwrap(model):
    """(Gpt or None, timestep map or None) behind _WrappedModel / DistributedDataParallel wrappers."""
    tmap = None
    if isinstance(model, _WrappedModel):
        tmap, model = model.timestep_map, model.model
    if hasattr(model, "module") and isinstance(getattr(model, "module"), Gpt):
        model = model.module
    return (model if isinstance(model, Gpt) else None), tmap


# ------------------------------------------------------------------------------------------ the process
class GaussianDiffusion:
    """Training and sampling utilities (reference: gaussian_diffusion.py:138-852)."""

    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type):
        self.model_mean_type, self.model_var_type, self.loss_type = model_mean_type, model_var_type, loss_type
        betas = np.array(betas, dtype=np.float64)
        assert betas.ndim == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.append(1.0, ac[:-1])
        self.alphas_cumprod_next = np.append(ac[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:])
        ) if len(self.posterior_variance) > 1 else np.array([])
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - ac)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)
        self._coef_cache = {}

    # ---- device tables
    def _model_variance_tables(self):
        if self.model_var_type == ModelVarType.FIXED_LARGE:
            var = np.append(self.posterior_variance[1], self.betas[1:])
            return var, np.log(var)
        if self.model_var_type == ModelVarType.FIXED_SMALL:
            return self.posterior_variance, self.posterior_log_variance_clipped
        raise NotImplementedError("learned variances are outside the G.pt hot path (main.py:217: learn_sigma=False)")

    def coef_table(self, device):
        """fp32 [num_timesteps, 8] device table (layout: include/gpt_hip.h GPT_C_*): float64 -> fp32 once,
        the same rounding the reference applies per gather (gaussian_diffusion.py:864)."""
        key = str(device)
        tab = self._coef_cache.get(key)
        if tab is None:
            var, logvar = self._model_variance_tables()
            cols = [self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, self.posterior_mean_coef1,
                    self.posterior_mean_coef2, logvar, self.sqrt_recip_alphas_cumprod,
                    self.sqrt_recipm1_alphas_cumprod, var]
            host = torch.from_numpy(np.stack(cols, axis=1)).float().contiguous()
            tab = self._coef_cache[key] = host.to(device)
        return tab

    # ---- forward process
    def q_sample(self, x_start, t, noise=None):
        """x_t ~ q(x_t | x_0) (gaussian_diffusion.py:209-224), one fused kernel."""
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        B = x_start.shape[0]
        x0 = x_start.detach().float().contiguous().view(B, -1)
        nz = noise.detach().float().contiguous().view(B, -1)
        out = torch.empty_like(x0)
        L.check(L.lib().gpt_q_sample(L.ptr(x0), L.ptr(nz), L.ptr(t.long().contiguous()),
                                     L.ptr(self.coef_table(x0.device)), L.ptr(out), B, x0.shape[1], L.stream()),
                "gpt_q_sample")
        return out.view(x_start.shape)

    def q_mean_variance(self, x_start, t):
        """q(x_t | x_0): (mean, variance, log_variance), all of x_start's shape (gaussian_diffusion.py:196-207); the two
        variances are broadcast views of the per-sample scalars."""
        B = x_start.shape[0]
        t = t.long().contiguous()
        x0 = x_start.detach().float().contiguous()
        mean = self.q_sample(x0, t, noise=torch.zeros_like(x0))           # sqrt(ac) * x0 + sqrt(1 - ac) * 0
        shape = [B] + [1] * (x_start.dim() - 1)
        var = torch.from_numpy(1.0 - self.alphas_cumprod).to(x0.device)[t].float()
        logvar = torch.from_numpy(self.log_one_minus_alphas_cumprod).to(x0.device)[t].float()
        return mean, var.view(shape).expand_as(x0), logvar.view(shape).expand_as(x0)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        """q(x_{t-1} | x_t, x_0) (gaussian_diffusion.py:226-246): the mean through the fused posterior kernel."""
        assert x_start.shape == x_t.shape
        B = x_t.shape[0]
        t = t.long().contiguous()
        x0, xt = x_start.detach().float().contiguous(), x_t.detach().float().contiguous()
        mean, _ = self._posterior(x0, xt, None, t, None, is_xstart=True)
        shape = [B] + [1] * (x_t.dim() - 1)
        var = torch.from_numpy(self.posterior_variance).to(xt.device)[t].float()
        logvar = torch.from_numpy(self.posterior_log_variance_clipped).to(xt.device)[t].float()
        return mean, var.view(shape).expand_as(xt), logvar.view(shape).expand_as(xt)

    def _prior_bpd(self, x_start):
        """Prior KL term of the variational bound in bits per dimension, one value per row (gaussian_diffusion.py:783-797;
        Gpt/data/prepare_checkpoints.py:98 prints its mean for the chosen openai_coeff)."""
        B = x_start.shape[0]
        x0 = x_start.detach().float().contiguous().view(B, -1)
        out = torch.empty(B, dtype=torch.float32, device=x0.device)
        s = float(np.float32(self.sqrt_alphas_cumprod[-1]))
        lv = float(np.float32(self.log_one_minus_alphas_cumprod[-1]))
        L.check(L.lib().gpt_prior_bpd(L.ptr(x0), s, lv, L.ptr(out), B, x0.shape[1], L.stream()), "gpt_prior_bpd")
        return out

    # ---- reverse process
    def _posterior(self, model_output, x, noise, t, clamp, want_xstart=False, is_xstart=None):
        if is_xstart is None:
            is_xstart = self.model_mean_type == ModelMeanType.START_X
        B = x.shape[0]
        D = x[0].numel()
        out = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_xstart else None
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        L.check(L.lib().gpt_posterior_step(
            L.ptr(model_output), L.ptr(x), L.ptr(noise), L.ptr(t), L.ptr(self.coef_table(x.device)),
            int(is_xstart), int(clamp is not None), lo, hi,
            L.ptr(out), L.ptr(x0), B, D, L.stream()), "gpt_posterior_step")
        return out, x0

    def _check_types(self):
        if self.model_mean_type == ModelMeanType.PREVIOUS_X:
            raise NotImplementedError("PREVIOUS_X prediction is outside the G.pt hot path")
        self._model_variance_tables()

    def _predict_xstart_from_eps(self, x_t, t, eps):
        c = self.coef_table(x_t.device)[t]
        return c[:, L.C_RECIP, None] * x_t - c[:, L.C_RECIPM1, None] * eps

    def _predict_eps_from_xstart(self, x_t, t, pred_xstart):
        c = self.coef_table(x_t.device)[t]
        return (c[:, L.C_RECIP, None] * x_t - pred_xstart) / c[:, L.C_RECIPM1, None]

    def p_mean_variance(self, model, x, t, clip_denoised=True, denoised_fn=None, model_kwargs=None):
        """p(x_{t-1} | x_t) and the x0 prediction (gaussian_diffusion.py:248-326).  `variance` and
        `log_variance` are broadcast views of the per-sample scalars."""
        self._check_types()
        model_kwargs = model_kwargs or {}
        B = x.shape[0]
        assert t.shape == (B,)
        t = t.long().contiguous()
        x = x.detach().float().contiguous()
        model_output = model(x, t, **model_kwargs)
        if isinstance(model_output, tuple):
            model_output, extra = model_output
        else:
            extra = None
        model_output = model_output.detach().float().contiguous()
        model_output, clamp, is_x0 = self._apply_denoised_fn(model_output, x, t, clip_denoised, denoised_fn)
        mean, pred = self._posterior(model_output, x, None, t, clamp, want_xstart=True, is_xstart=is_x0)
        c = self.coef_table(x.device)[t]
        shape = [B] + [1] * (x.dim() - 1)
        return {"mean": mean, "variance": c[:, L.C_SPARE].view(shape).expand_as(x),
                "log_variance": c[:, L.C_LOGVAR].view(shape).expand_as(x), "pred_xstart": pred, "extra": extra}

    def _apply_denoised_fn(self, model_output, x, t, clip_denoised, denoised_fn):
        """Folds denoised_fn / clip_denoised into the kernel's clamp where possible; an opaque callable is
        applied on the x0 prediction first (then the kernel is handed an x0 prediction).
        Returns (tensor for the kernel, clamp or None, tensor-is-an-x0-prediction)."""
        clamp = None
        is_x0 = self.model_mean_type == ModelMeanType.START_X
        if isinstance(denoised_fn, StaticThreshold):
            clamp = (denoised_fn.lo, denoised_fn.hi)
            if clip_denoised:
                clamp = (min(max(clamp[0], -1.0), 1.0), min(max(clamp[1], -1.0), 1.0))
        elif denoised_fn is not None:
            if self.model_mean_type != ModelMeanType.START_X:
                model_output = self._predict_xstart_from_eps(x, t, model_output)
            model_output = denoised_fn(model_output).float().contiguous()
            if clip_denoised:
                clamp = (-1.0, 1.0)
            return model_output, clamp, True
        elif clip_denoised:
            clamp = (-1.0, 1.0)
        return model_output, clamp, is_x0

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        """One reverse step (gaussian_diffusion.py:370-411): the model call plus ONE fused kernel."""
        self._check_types()
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is never used by G.pt and is not on the HIP path")
        model_kwargs = model_kwargs or {}
        t = t.long().contiguous()
        x = x.detach().float().contiguous()
        model_output = model(x, t, **model_kwargs)
        if isinstance(model_output, tuple):
            model_output = model_output[0]
        model_output = model_output.detach().float().contiguous()
        noise = torch.randn_like(x)                       # drawn at every step, t == 0 included (:404)
        model_output, clamp, is_x0 = self._apply_denoised_fn(model_output, x, t, clip_denoised, denoised_fn)
        sample, pred = self._posterior(model_output, x, noise, t, clamp, want_xstart=True, is_xstart=is_x0)
        return {"sample": sample, "pred_xstart": pred}

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, step_noise=None, seed=None):
        """Samples x_0 (gaussian_diffusion.py:413-456).

        When `model` is a `Gpt` (optionally behind DDP) and denoised_fn is None or a `StaticThreshold`, the
        whole loop runs in ONE host call (gpt_sample_loop): no per-step Python, host sync or H2D copy.
        Extensions over the reference signature: `step_noise` (fp32 [num_timesteps, N, D], slab i = the noise
        drawn at index i) injects the per-step noise for parity runs; `seed` seeds the in-kernel Philox
        generator (default: drawn from torch's CPU generator, so torch.manual_seed makes runs repeatable)."""
        gpt, _ = _unwrap(model)
        fusable = (gpt is not None and cond_fn is None and not progress
                   and (denoised_fn is None or isinstance(denoised_fn, StaticThreshold))
                   and model_kwargs is not None and {"loss_target", "loss_prev", "x_prev"} <= set(model_kwargs)
                   and not (gpt.training and gpt.decoder.dropout_p > 0))
        if fusable:
            return self._fused_loop(gpt, shape, noise, clip_denoised, denoised_fn, model_kwargs, device, step_noise, seed)
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                     denoised_fn=denoised_fn, cond_fn=cond_fn, model_kwargs=model_kwargs,
                                                     device=device, progress=progress):
            final = sample
        return final["sample"]

    def _timestep_map(self):
        return list(range(self.num_timesteps))

    def _fused_loop(self, gpt, shape, noise, clip_denoised, denoised_fn, model_kwargs, device, step_noise, seed,
                    ddim_eta=None):
        self._check_types()
        dev = torch.device(device) if device is not None else next(gpt.parameters()).device
        if dev.type != "cuda":
            raise L.GptHipError("p_sample_loop: the HIP path needs a GPU device")
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        B, D = int(shape[0]), int(np.prod(shape[1:]))
        assert D == gpt.D
        x = (noise.detach().to(dev).float().clone() if noise is not None
             else torch.randn(B, D, device=dev)).contiguous().view(B, D)
        zeros_t = torch.zeros(B, dtype=torch.long, device=dev)
        xk, _, lt, lp, xp = gpt._prepare(x, zeros_t, model_kwargs["loss_target"], model_kwargs["loss_prev"],
                                         model_kwargs["x_prev"].to(dev))
        _, clamp, _ = self._apply_denoised_fn(x, x, zeros_t, clip_denoised, denoised_fn)
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        eng = gpt.engine(dev)
        ws = eng.workspace(B)
        n = self.num_timesteps
        tmap = (C.c_int64 * n)(*self._timestep_map())
        t_model, t_index = torch.empty_like(zeros_t), torch.empty_like(zeros_t)
        model_out = torch.empty_like(x)
        noise_buf = None
        if step_noise is not None:
            step_noise = step_noise.detach().to(dev).float().contiguous()
            assert step_noise.shape[0] == n and step_noise[0].numel() == B * D
        else:
            noise_buf = torch.empty_like(x)
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        px = int(self.model_mean_type == ModelMeanType.START_X)
        if ddim_eta is None:
            L.check(L.lib().gpt_sample_loop(
                C.byref(eng.cmodel), C.byref(ws), L.ptr(x), L.ptr(lt), L.ptr(lp), int(lt.dtype == torch.float64), L.ptr(xp),
                L.ptr(self.coef_table(dev)), tmap, n, n - 1, 0, px,
                int(clamp is not None), lo, hi, L.ptr(step_noise), int(seed or 0), L.ptr(t_model), L.ptr(t_index),
                L.ptr(model_out), L.ptr(noise_buf), B, L.stream()), "gpt_sample_loop")
        else:
            L.check(L.lib().gpt_ddim_sample_loop(
                C.byref(eng.cmodel), C.byref(ws), L.ptr(x), L.ptr(lt), L.ptr(lp), int(lt.dtype == torch.float64), L.ptr(xp),
                L.ptr(self.coef_table(dev)), L.ptr(self._ac_table(dev)), ddim_eta, tmap, n, n - 1, 0, px,
                int(clamp is not None), lo, hi, L.ptr(step_noise), int(seed or 0), L.ptr(t_model), L.ptr(t_index),
                L.ptr(model_out), L.ptr(noise_buf), B, L.stream()), "gpt_ddim_sample_loop")
        return x.view(*shape)

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False):
        """Generator over the per-step dicts (gaussian_diffusion.py:458-505); each step is the model call
        plus one fused posterior kernel."""
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        t = torch.empty(shape[0], dtype=torch.long, device=img.device)
        for i in indices:
            t.fill_(i)
            with torch.no_grad():
                out = self.p_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                    cond_fn=cond_fn, model_kwargs=model_kwargs)
                yield out
                img = out["sample"]

    # ---- DDIM (gaussian_diffusion.py:507-674): fused step kernel, and the whole loop in one
    # host call when the model is a Gpt
    def _ac_table(self, device):
        key = "ac:" + str(device)
        tab = self._coef_cache.get(key)
        if tab is None:
            tab = self._coef_cache[key] = torch.from_numpy(self.alphas_cumprod).float().contiguous().to(device)
        return tab

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0,
                    noise=None):
        """One DDIM step (gaussian_diffusion.py:507-552).  `noise` (extension): the N(0,1) draw of this step, for parity runs."""
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is never used by G.pt and is not on the HIP path")
        self._check_types()
        if model_kwargs is None:
            model_kwargs = {}
        assert t.shape == (x.shape[0],)
        x = x.detach().float().contiguous()
        t = t.long().contiguous()
        model_output = model(x, t, **model_kwargs)
        if isinstance(model_output, tuple):
            model_output = model_output[0]
        model_output = model_output.detach().float().contiguous()
        model_output, clamp, is_x0 = self._apply_denoised_fn(model_output, x, t, clip_denoised, denoised_fn)
        if noise is None:
            noise = torch.randn_like(x)          # drawn at every step, as the reference does (:539)
        B, D = x.shape[0], x[0].numel()
        out, x0 = torch.empty_like(x), torch.empty_like(x)
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        L.check(L.lib().gpt_ddim_step(
            L.ptr(model_output), L.ptr(x), L.ptr(noise.detach().float().contiguous()), L.ptr(t),
            L.ptr(self.coef_table(x.device)), L.ptr(self._ac_table(x.device)), int(is_x0), int(clamp is not None), lo, hi,
            float(eta), L.ptr(out), L.ptr(x0), B, D, L.stream()), "gpt_ddim_step")
        return {"sample": out, "pred_xstart": x0}

    def ddim_reverse_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                            eta=0.0):
        """x_{t+1} from x_t by the DDIM reverse ODE (gaussian_diffusion.py:559-592)."""
        assert eta == 0.0, "Reverse ODE only for deterministic path"
        if cond_fn is not None:
            raise NotImplementedError("cond_fn guidance is never used by G.pt and is not on the HIP path")
        self._check_types()
        model_kwargs = model_kwargs or {}
        assert t.shape == (x.shape[0],)
        x = x.detach().float().contiguous()
        t = t.long().contiguous()
        model_output = model(x, t, **model_kwargs)
        if isinstance(model_output, tuple):
            model_output = model_output[0]
        model_output = model_output.detach().float().contiguous()
        model_output, clamp, is_x0 = self._apply_denoised_fn(model_output, x, t, clip_denoised, denoised_fn)
        B, D = x.shape[0], x[0].numel()
        out, x0 = torch.empty_like(x), torch.empty_like(x)
        lo, hi = clamp if clamp is not None else (0.0, 0.0)
        L.check(L.lib().gpt_ddim_reverse_step(
            L.ptr(model_output), L.ptr(x), L.ptr(t), L.ptr(self.coef_table(x.device)), L.ptr(self._ac_table(x.device)),
            self.num_timesteps, int(is_x0), int(clamp is not None), lo, hi, L.ptr(out), L.ptr(x0), B, D, L.stream()),
            "gpt_ddim_reverse_step")
        return {"sample": out, "pred_xstart": x0}

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, step_noise=None, seed=None):
        """gaussian_diffusion.py:594-628.  Fused into one host call (gpt_ddim_sample_loop) under the same conditions as
        p_sample_loop; `step_noise` / `seed` as there."""
        gpt, _ = _unwrap(model)
        fusable = (gpt is not None and cond_fn is None and not progress
                   and (denoised_fn is None or isinstance(denoised_fn, StaticThreshold))
                   and model_kwargs is not None and {"loss_target", "loss_prev", "x_prev"} <= set(model_kwargs)
                   and not (gpt.training and gpt.decoder.dropout_p > 0))
        if fusable:
            return self._fused_loop(gpt, shape, noise, clip_denoised, denoised_fn, model_kwargs, device, step_noise, seed,
                                    ddim_eta=float(eta))
        final = None
        for sample in self.ddim_sample_loop_progressive(model, shape, noise=noise, clip_denoised=clip_denoised,
                                                        denoised_fn=denoised_fn, cond_fn=cond_fn,
                                                        model_kwargs=model_kwargs, device=device, progress=progress,
                                                        eta=eta):
            final = sample
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                     model_kwargs=None, device=None, progress=False, eta=0.0):
        if device is None:
            device = next(model.parameters()).device
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        t = torch.empty(shape[0], dtype=torch.long, device=img.device)
        for i in indices:
            t.fill_(i)
            with torch.no_grad():
                out = self.ddim_sample(model, img, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn,
                                       cond_fn=cond_fn, model_kwargs=model_kwargs, eta=eta)
                yield out
                img = out["sample"]

    # ---- training
    def training_losses(self, model, x_start, t, model_kwargs=None, noise=None):
        """{'mse': (N,), 'loss': (N,)} for one batch of timesteps (gaussian_diffusion.py:709-781, MSE branch)."""
        if self.loss_type not in (LossType.MSE, LossType.RESCALED_MSE):
            raise NotImplementedError("variational-bound losses are outside the G.pt hot path")
        self._check_types()
        model_kwargs = model_kwargs or {}
        if noise is None:
            noise = torch.randn_like(x_start)
        t = t.long().contiguous()
        x_t = self.q_sample(x_start, t, noise=noise)
        model_output = model(x_t, t, **model_kwargs)
        target = x_start if self.model_mean_type == ModelMeanType.START_X else noise
        assert model_output.shape == target.shape == x_start.shape
        from .autograd import mse_rows
        mse = mse_rows(target.detach().float().contiguous(), model_output)
        return {"mse": mse, "loss": mse}


class SpacedDiffusion(GaussianDiffusion):
    """A process that keeps a subset of the base timesteps (respace.py:61-110)."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base = GaussianDiffusion(**kwargs)
        last, new_betas = 1.0, []
        for i, ac in enumerate(base.alphas_cumprod):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)
        self._map_cache = {}

    def _timestep_map(self):
        return self.timestep_map

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.original_num_steps, self._map_cache)

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def p_sample(self, model, *args, **kwargs):
        return super().p_sample(self._wrap_model(model), *args, **kwargs)

    def ddim_reverse_sample(self, model, *args, **kwargs):
        return super().ddim_reverse_sample(self._wrap_model(model), *args, **kwargs)

    def ddim_sample(self, model, *args, **kwargs):
        return super().ddim_sample(self._wrap_model(model), *args, **kwargs)

    def training_losses(self, model, *args, **kwargs):
        return super().training_losses(self._wrap_model(model), *args, **kwargs)

    def _scale_timesteps(self, t):
        return t


class _WrappedModel:
    """Maps respaced timestep indices back to the base process before the model call (respace.py:113-125).
    The map lives on the device (cached), so the call adds no host-to-device copy."""

    def __init__(self, model, timestep_map, original_num_steps, cache=None):
        self.model, self.timestep_map, self.original_num_steps = model, timestep_map, original_num_steps
        self._cache = cache if cache is not None else {}

    def parameters(self):
        return self.model.parameters()

    def __call__(self, x, ts, **kwargs):
        key = (str(ts.device), ts.dtype)
        m = self._cache.get(key)
        if m is None:
            m = self._cache[key] = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        return self.model(x, m[ts], **kwargs)


def create_diffusion(*, steps=1000, learn_sigma=True, sigma_small=False, noise_schedule="linear", use_kl=False,
                     predict_xstart=False, rescale_learned_sigmas=False, timestep_respacing=""):
    """Factory with the reference's keywords (Gpt/diffusion/__init__.py:5-43)."""
    betas = get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = LossType.RESCALED_MSE
    else:
        loss_type = LossType.MSE
    if not timestep_respacing:
        timestep_respacing = [steps]
    if learn_sigma:
        var_type = ModelVarType.LEARNED_RANGE
    else:
        var_type = ModelVarType.FIXED_SMALL if sigma_small else ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
        model_mean_type=ModelMeanType.START_X if predict_xstart else ModelMeanType.EPSILON,
        model_var_type=var_type, loss_type=loss_type)


# ------------------------------------------------------------------------------------------ timestep samplers
class ScheduleSampler:
    """Importance sampler over timesteps (timestep_sampler.py:23-55): numpy's global RNG on the host."""

    def weights(self):
        raise NotImplementedError

    def sample(self, batch_size, device):
        w = self.weights()
        p = w / np.sum(w)
        idx = np.random.choice(len(p), size=(batch_size,), p=p)
        indices = torch.from_numpy(idx).long().to(device)
        weights = torch.from_numpy(1 / (len(p) * p[idx])).float().to(device)
        return indices, weights


class UniformSampler(ScheduleSampler):
    """timestep_sampler.py:58-64."""

    def __init__(self, diffusion):
        self.diffusion = diffusion
        self._weights = np.ones([diffusion.num_timesteps])

    def weights(self):
        return self._weights


def create_named_schedule_sampler(name, diffusion):
    if name == "uniform":
        return UniformSampler(diffusion)
    raise NotImplementedError(f"schedule sampler {name!r}: only 'uniform' is on the G.pt path (main.py:220)")
