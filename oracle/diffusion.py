"""Oracle restatement of the DDPM process used by G.pt (TEST INFRASTRUCTURE, see oracle/__init__.py).

float64 numpy schedule tables, fp32 torch-CPU per-step arithmetic, noise always INJECTED (CPU and GPU
generators produce different streams, SURVEY.md section 7 "RNG parity").
References are to /root/reference/Gpt/diffusion/.
"""
import numpy as np
import torch


def linear_betas(steps):
    """gaussian_diffusion.py:92-109 ('linear' named schedule)."""
    scale = 1000 / steps
    return np.linspace(scale * 0.0001, scale * 0.02, steps, dtype=np.float64)


def space_timesteps(num_timesteps, section_counts):
    """respace.py:8-58."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError("no integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("section too small")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class Tables:
    """GaussianDiffusion.__init__ (gaussian_diffusion.py:147-195) after SpacedDiffusion has
    re-derived the betas from the retained alphas_cumprod (respace.py:69-83)."""

    def __init__(self, steps=1000, respacing=""):
        base = linear_betas(steps)
        ac = np.cumprod(1.0 - base, axis=0)
        use = space_timesteps(steps, respacing if respacing else [steps])
        last, betas, tmap = 1.0, [], []
        for i, a in enumerate(ac):
            if i in use:
                betas.append(1 - a / last)
                last = a
                tmap.append(i)
        b = np.array(betas, dtype=np.float64)
        self.timestep_map = tmap
        self.betas = b
        self.T = len(b)
        al = 1.0 - b
        self.alphas_cumprod = np.cumprod(al, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(al) / (1.0 - self.alphas_cumprod)
        # FIXED_LARGE (gaussian_diffusion.py:292-295)
        self.fixed_large_log_variance = np.log(np.append(self.posterior_variance[1], b[1:]))


def _take(arr, t):
    """_extract_into_tensor (gaussian_diffusion.py:855-867): f64 gather then .float()."""
    return torch.from_numpy(arr)[t].float()[:, None]


def q_sample(tb, x0, t, noise):
    """gaussian_diffusion.py:209-224."""
    return _take(tb.sqrt_alphas_cumprod, t) * x0 + _take(tb.sqrt_one_minus_alphas_cumprod, t) * noise


def xstart_from_eps(tb, x_t, t, eps):
    """gaussian_diffusion.py:328-333."""
    return _take(tb.sqrt_recip_alphas_cumprod, t) * x_t - _take(tb.sqrt_recipm1_alphas_cumprod, t) * eps


def posterior_step(tb, model_out, x_t, t, noise, predict_xstart=True, clip=False, lo=None, hi=None):
    """p_mean_variance FIXED_LARGE (gaussian_diffusion.py:289-317) + p_sample (:404-410).
    ``lo``/``hi``: optional denoised_fn clamp (Gpt/vis.py:190-207, 'static' thresholding)."""
    x0 = model_out if predict_xstart else xstart_from_eps(tb, x_t, t, model_out)
    if lo is not None:
        x0 = x0.clamp(lo, hi)
    if clip:
        x0 = x0.clamp(-1, 1)
    mean = _take(tb.posterior_mean_coef1, t) * x0 + _take(tb.posterior_mean_coef2, t) * x_t
    logvar = _take(tb.fixed_large_log_variance, t)
    mask = (t != 0).float()[:, None]
    return mean + mask * torch.exp(0.5 * logvar) * noise, x0


def p_sample_loop(tb, model_fn, start, step_noise, predict_xstart=True, clip=False):
    """p_sample_loop_progressive (gaussian_diffusion.py:458-505) with injected noise.
    ``model_fn(x, t_mapped)``; ``step_noise(i)`` returns the N(0,1) tensor drawn at index i
    (the reference draws one even at i == 0, :404)."""
    x = start
    B = x.shape[0]
    tmap = torch.tensor(tb.timestep_map)
    for i in reversed(range(tb.T)):
        t = torch.full((B,), i, dtype=torch.long)
        out = model_fn(x, tmap[t])                                 # respace.py:120-125
        x, _ = posterior_step(tb, out, x, t, step_noise(i), predict_xstart, clip)
    return x


def _sqrt32(x):
    """Correctly rounded fp32 square root (numpy's is on every host; torch's vectorised CPU sqrt was seen 1 ulp off on
    the GPU box's host CPU, which matters only for a bit-exact check of the per-row scalars)."""
    return torch.from_numpy(np.sqrt(x.numpy().astype(np.float32)))


def ddim_step(tb, model_out, x_t, t, noise, eta=0.0, predict_xstart=True, clip=False, lo=None, hi=None):
    """ddim_sample (gaussian_diffusion.py:507-552), fp32 op by op."""
    x0 = model_out if predict_xstart else xstart_from_eps(tb, x_t, t, model_out)
    if lo is not None:
        x0 = x0.clamp(lo, hi)
    if clip:
        x0 = x0.clamp(-1, 1)
    eps = (_take(tb.sqrt_recip_alphas_cumprod, t) * x_t - x0) / _take(tb.sqrt_recipm1_alphas_cumprod, t)   # :335-338
    ab, abp = _take(tb.alphas_cumprod, t), _take(tb.alphas_cumprod_prev, t)
    sigma = eta * _sqrt32((1 - abp) / (1 - ab)) * _sqrt32(1 - ab / abp)
    mean = x0 * _sqrt32(abp) + _sqrt32(1 - abp - sigma ** 2) * eps
    mask = (t != 0).float()[:, None]
    return mean + mask * sigma * noise, x0


def ddim_reverse_step(tb, model_out, x_t, t, predict_xstart=True, clip=False):
    """ddim_reverse_sample (gaussian_diffusion.py:559-592), fp32 op by op."""
    x0 = model_out if predict_xstart else xstart_from_eps(tb, x_t, t, model_out)
    if clip:
        x0 = x0.clamp(-1, 1)
    eps = (_take(tb.sqrt_recip_alphas_cumprod, t) * x_t - x0) / _take(tb.sqrt_recipm1_alphas_cumprod, t)
    abn = _take(tb.alphas_cumprod_next, t)
    return x0 * _sqrt32(abn) + _sqrt32(1 - abn) * eps, x0


def q_mean_variance(tb, x0, t):
    """gaussian_diffusion.py:196-207 (per-sample scalars for the two variances)."""
    return (_take(tb.sqrt_alphas_cumprod, t) * x0, _take(1.0 - tb.alphas_cumprod, t)[:, 0],
            _take(tb.log_one_minus_alphas_cumprod, t)[:, 0])


def q_posterior_mean_variance(tb, x0, x_t, t):
    """gaussian_diffusion.py:226-246."""
    mean = _take(tb.posterior_mean_coef1, t) * x0 + _take(tb.posterior_mean_coef2, t) * x_t
    return mean, _take(tb.posterior_variance, t)[:, 0], _take(tb.posterior_log_variance_clipped, t)[:, 0]


def prior_bpd(tb, x0):
    """_prior_bpd (gaussian_diffusion.py:783-797) with normal_kl (diffusion_utils.py:15-41) against N(0, I)."""
    t = torch.full((x0.shape[0],), tb.T - 1, dtype=torch.long)
    mean1, lv1 = _take(tb.sqrt_alphas_cumprod, t) * x0, _take(tb.log_one_minus_alphas_cumprod, t).expand_as(x0)
    lv2 = torch.tensor(0.0)
    kl = 0.5 * (-1.0 + lv2 - lv1 + torch.exp(lv1 - lv2) + ((mean1 - 0.0) ** 2) * torch.exp(-lv2))
    return kl.mean(dim=1) / np.log(2.0)


def ddim_sample_loop(tb, model_fn, start, step_noise, eta=0.0, predict_xstart=True, clip=False):
    """ddim_sample_loop_progressive (gaussian_diffusion.py:630-674) with injected noise."""
    x = start
    B = x.shape[0]
    tmap = torch.tensor(tb.timestep_map)
    for i in reversed(range(tb.T)):
        t = torch.full((B,), i, dtype=torch.long)
        out = model_fn(x, tmap[t])
        x, _ = ddim_step(tb, out, x, t, step_noise(i), eta, predict_xstart, clip)
    return x


def training_mse(tb, model_fn, x_start, t, noise, predict_xstart=True):
    """training_losses MSE branch (gaussian_diffusion.py:721-725, 740-741, 765-777)."""
    x_t = q_sample(tb, x_start, t, noise)
    tmap = torch.tensor(tb.timestep_map)
    out = model_fn(x_t, tmap[t])
    target = x_start if predict_xstart else noise
    return ((target - out) ** 2).mean(dim=1), out, x_t
