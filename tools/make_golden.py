#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (build container
only: /root/reference does not exist on the GPU box, so only the resulting .npz files travel).

Run:  python tools/make_golden.py            (from the repo root, CPU only, ~1-2 min)

Every fixture is data: seeds/inputs and the reference's outputs.  Model weights are not stored;
they are regenerated from a seed by ``oracle.denoiser.random_state_dict`` and loaded into the
reference module with ``load_state_dict`` (a few weight checksums are stored to catch RNG drift).

One deviation from the reference is unavoidable on a CPU-only box: ``FrequencyEmbedder.forward``
hard-codes ``.to('cuda', torch.float)`` (Gpt/models/transformer.py:329); it is replaced here by the
same arithmetic on the input's own device.
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)

# stub modules that the data_gen scripts import at module level but the permutation code never uses
for name in ("simplejson", "lmdb", "torchvision", "torchvision.datasets", "torchvision.transforms"):
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

import Gpt.models.transformer as ref_tf                      # noqa: E402
import Gpt.diffusion.gaussian_diffusion as ref_gd            # noqa: E402
from Gpt.diffusion import create_diffusion                   # noqa: E402
from Gpt.diffusion.timestep_sampler import UniformSampler    # noqa: E402
from Gpt.data.augment import random_permute_flat             # noqa: E402

from oracle import denoiser as od                            # noqa: E402
from tests.cases import CASES, make_inputs, step_noise      # noqa: E402


def _freq_forward(self, x):
    n = x.size(0)
    if x.dim() == 1:
        x = x.unsqueeze(1)
    xu = x.unsqueeze(-1).to(self.frequencies.device, torch.float)
    scaled = self.frequencies.view(1, 1, -1) * xu
    return torch.cat([torch.sin(scaled), torch.cos(scaled), xu], dim=-1).view(n, -1)


ref_tf.FrequencyEmbedder.forward = _freq_forward


def ref_model(case, seed=0):
    c = CASES[case]
    m = ref_tf.Gpt(
        parameter_sizes=list(c["parameter_sizes"]), parameter_names=list(c["parameter_names"]),
        num_frequencies=128, max_freq_log2=14, predict_xstart=c.get("predict_xstart", True),
        absolute_loss_conditioning=True, n_embd=c["n_embd"], n_head=c["n_head"], n_layer=c["n_layer"],
        chunk_size=c["chunk_size"], split_policy=c["split_policy"], encoder_depth=1, decoder_depth=1,
        attn_pdrop=0.0, resid_pdrop=0.0, embd_pdrop=0.0)
    plan = od.Plan(c["parameter_sizes"], 128, 14, True, c["n_head"], c["n_embd"], c["n_layer"],
                   c["chunk_size"], c["split_policy"], c.get("predict_xstart", True))
    sd = od.random_state_dict(plan, seed=seed)
    m.load_state_dict(sd)
    return m.eval(), plan, sd


def slim(a, stride=37):
    """Large tensors are stored as a strided sample (tests apply the same stride: tests/cases.slim)."""
    a = np.asarray(a)
    return a.reshape(-1)[::stride].copy() if a.size > 20000 else a


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1e3:.1f} kB)")


# ------------------------------------------------------------------ G1: token plans
def g1_splits():
    out = {}
    tasks = {"mnist": ([7840, 10, 100, 10], 1000), "cifar": ([432, 4608, 320, 10], 576)}
    for tname, (sizes, chunk) in tasks.items():
        nested = [sizes, sizes, [257], [257], [257], [257]]
        for pol in ("chunk", "layer_by_layer", "chunk_within_layer", "chunk_within_input"):
            out[f"{tname}_{pol}_in"] = np.array(ref_tf.GPT.build_splits(nested, pol, chunk))
            out[f"{tname}_{pol}_out"] = np.array(ref_tf.GPT.build_splits([sizes], pol, chunk))
    save("g1_splits", **out)


# ------------------------------------------------------------------ G2: frequency embedding
def g2_freq():
    emb = ref_tf.FrequencyEmbedder(128, 14)
    t = torch.tensor([0, 1, 2, 500, 998, 999], dtype=torch.long)
    l32 = torch.tensor([0.0, 0.05, 1.2, 2.3, 35.0, 90.0], dtype=torch.float32)
    l64 = l32.double() + 1e-9
    save("g2_freq", frequencies=emb.frequencies.numpy(), t=t.numpy(), t_emb=emb(t).numpy(),
         l32=l32.numpy(), l32_emb=emb(l32).numpy(), l64=l64.numpy(), l64_emb=emb(l64).numpy(),
         l32_col_emb=emb(l32.view(-1, 1)).numpy())


# ------------------------------------------------------------------ G3: reduced-model forward + taps
def g3_forward(case):
    m, plan, sd = ref_model(case)
    x, t, lt, lp, xp = make_inputs(case, B=4, seed=11)
    taps = {}
    hooks = []
    dec = m.decoder
    hooks.append(dec.ln_in.register_forward_hook(lambda mod, i, o: taps.__setitem__("ln_in", o.detach().clone())))
    hooks.append(dec.blocks[0].attn.register_forward_hook(lambda mod, i, o: taps.__setitem__("attn0", o.detach().clone())))
    hooks.append(dec.blocks[0].register_forward_hook(lambda mod, i, o: taps.__setitem__("block0", o.detach().clone())))
    hooks.append(dec.ln_f.register_forward_hook(lambda mod, i, o: taps.__setitem__("ln_f", o.detach().clone())))
    with torch.no_grad():
        out = m(x, t, lt, lp, xp)
        # loss scalars may arrive as (N,1) and as float64
        out_col = m(x, t, lt.view(-1, 1).double(), lp.view(-1, 1).double(), xp)
    for h in hooks:
        h.remove()
    save(f"g3_forward_{case}", out=out.numpy(), out_col64=out_col.numpy(),
         ln_in=taps["ln_in"].numpy(), attn0=taps["attn0"].numpy(), block0=taps["block0"].numpy(),
         ln_f=taps["ln_f"].numpy(),
         w_checksum=np.array([float(sd["decoder.blocks.0.attn.key.weight"].double().sum()),
                              float(sd["decoder.pos_emb"].double().sum()),
                              float(sd["decoder.output_parameter_projections.0.0.weight"].double().sum())]))


# ------------------------------------------------------------------ G4: full-width single block
def g4_block():
    out = {}
    for name, (d, H) in {"mnist": (1536, 16), "cifar": (2048, 16)}.items():
        blk = ref_tf.Block(d, H)
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():
            for p in blk.parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1 + (1.0 if p is blk.ln1.weight or p is blk.ln2.weight else 0.0))
            xin = torch.randn(2, 26, d, generator=g)
            y = blk(xin)
        out[f"{name}_y"] = y.numpy()
    save("g4_block", **out)


# ------------------------------------------------------------------ G5: diffusion tables
def g5_tables():
    out = {}
    for tag, resp in (("full", ""), ("r50", "50"), ("ddim25", "ddim25")):
        d = create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000,
                             timestep_respacing=resp)
        for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                  "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                  "posterior_mean_coef2"):
            out[f"{tag}_{k}"] = getattr(d, k)
        out[f"{tag}_timestep_map"] = np.array(d.timestep_map)
    save("g5_tables", **out)


# ------------------------------------------------------------------ G6: q_sample / training loss / grads
def g6_training(case, predict_xstart=True):
    c = dict(CASES[case])
    m, plan, sd = ref_model(case)
    m.predict_xstart = predict_xstart
    d = create_diffusion(learn_sigma=False, predict_xstart=predict_xstart, noise_schedule="linear", steps=1000)
    x, _, lt, lp, xp = make_inputs(case, B=6, seed=21)
    g = torch.Generator().manual_seed(22)
    x_start = xp + 0.1 * torch.randn(xp.shape, generator=g)
    t = torch.tensor([0, 1, 17, 500, 998, 999])
    noise = torch.randn(xp.shape, generator=g)
    x_t = d.q_sample(x_start, t, noise=noise)
    m.train()
    terms = d.training_losses(m, x_start, t, model_kwargs=dict(loss_target=lt.double(), loss_prev=lp.double(), x_prev=xp),
                              noise=noise)
    loss = terms["loss"].mean()
    loss.backward()
    named = dict(m.named_parameters())
    keys = ["decoder.blocks.1.mlp.0.weight", "decoder.blocks.0.attn.query.bias", "decoder.ln_in.weight",
            "decoder.input_parameter_projections.3.0.weight", "decoder.output_parameter_projections.1.0.weight",
            "decoder.pos_emb", "decoder.blocks.0.ln1.bias", "decoder.input_parameter_projections.23.0.weight"]
    grads = {f"grad::{k}": slim(named[k].grad.numpy()) for k in keys}
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in m.parameters())))
    tag = "x0" if predict_xstart else "eps"
    save(f"g6_training_{case}_{tag}", x_start=x_start.numpy(), t=t.numpy(), noise=noise.numpy(), x_t=x_t.numpy(),
         mse=terms["mse"].detach().numpy(), loss=float(loss), grad_norm=gn, **grads)


# ------------------------------------------------------------------ G7: p_sample_loop with injected noise
def g7_sampling(case, respacing, tag, predict_xstart=True, clip=False):
    m, plan, sd = ref_model(case)
    d = create_diffusion(learn_sigma=False, predict_xstart=predict_xstart, noise_schedule="linear", steps=1000,
                         timestep_respacing=respacing)
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    start = step_noise(1234, -1, B, plan.D)
    order = iter(reversed(range(d.num_timesteps)))
    real = ref_gd.th.randn_like

    def fake_randn_like(x):
        return step_noise(1234, next(order), B, plan.D)

    traj = {}
    ref_gd.th.randn_like = fake_randn_like
    try:
        i = d.num_timesteps
        for out in d.p_sample_loop_progressive(
                m, (B, plan.D), noise=start, clip_denoised=clip,
                model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), device="cpu"):
            i -= 1
            if i in (d.num_timesteps - 1, d.num_timesteps // 2, 1, 0):
                traj[f"sample_at_{i}"] = out["sample"].numpy()
                traj[f"xstart_at_{i}"] = out["pred_xstart"].numpy()
    finally:
        ref_gd.th.randn_like = real
    save(f"g7_sampling_{case}_{tag}", num_steps=d.num_timesteps, **traj)


# ------------------------------------------------------------------ G11: ddim_sample_loop with injected noise (row f2)
def g11_ddim(case, respacing, tag, eta, predict_xstart=True, clip=False):
    m, plan, sd = ref_model(case)
    d = create_diffusion(learn_sigma=False, predict_xstart=predict_xstart, noise_schedule="linear", steps=1000,
                         timestep_respacing=respacing)
    B = 3
    _, _, lt, lp, xp = make_inputs(case, B=B, seed=31)
    start = step_noise(1234, -1, B, plan.D)
    order = iter(reversed(range(d.num_timesteps)))
    real = ref_gd.th.randn_like

    def fake_randn_like(x):
        return step_noise(1234, next(order), B, plan.D)

    traj = {}
    ref_gd.th.randn_like = fake_randn_like
    try:
        i = d.num_timesteps
        for out in d.ddim_sample_loop_progressive(
                m, (B, plan.D), noise=start, clip_denoised=clip,
                model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), device="cpu", eta=eta):
            i -= 1
            if i in (d.num_timesteps - 1, d.num_timesteps // 2, 1, 0):
                traj[f"sample_at_{i}"] = out["sample"].numpy()
                traj[f"xstart_at_{i}"] = out["pred_xstart"].numpy()
    finally:
        ref_gd.th.randn_like = real
    save(f"g11_ddim_{case}_{tag}", num_steps=d.num_timesteps, eta=eta, **traj)


# ------------------------------------------------------------------ G8: permutation index maps
def g8_permute():
    import data_gen.train_mnist as tm
    import data_gen.train_cifar10 as tc
    out = {}
    for name, net, fn in (("mlp", tm.MLP(10), tm.random_permute_mlp), ("cnn", tc.ConvNet(), tc.random_permute_cnn)):
        D = sum(p.numel() for p in net.state_dict().values())
        ident = torch.arange(D)
        maps = [random_permute_flat(ident, net, seed, fn).numpy() for seed in range(16)]
        out[name] = np.stack(maps).astype(np.int32)
        # function preservation probe: permuted net == original net on random input (sanity only)
    save("g8_permute", **out)


# ------------------------------------------------------------------ G15: Cartpole policy permutation maps (the third shipped task)
def _reference_defs(path, names):
    """Compiles the named top-level definitions of one reference file, read as text at generation time, WITHOUT importing the
    module: data_gen/train_rl.py imports gym / hydra / omegaconf / isaacgym at module level, none of which exists here, and
    its permutation code uses none of them.  Nothing of the file is kept in this repository."""
    import ast
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in body} == set(names)
    from Gpt.data.augment import permute_in, permute_out, permute_in_out
    import torch.nn as nn
    ns = {"torch": torch, "nn": nn, "np": np, "permute_in": permute_in, "permute_out": permute_out,
          "permute_in_out": permute_in_out, "_TASKS": {"Cartpole": None}}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def reference_actor():
    """The reference's VisActorCritic('Cartpole', 4, 1) (Gpt/tasks.py:75) and random_permute_actor.  The class's __init__ reads its
    layer widths with OmegaConf (absent here): the instance is made with __new__ and given the attributes __init__ would
    (train_rl.py:108-121: actor = Linear / SELU chain over pi_hid_sizes, then log_std), the widths read from the reference's
    own data_gen/configs/rl/train/Cartpole.yaml."""
    import yaml
    import torch.nn as nn
    ns = _reference_defs("/root/reference/data_gen/train_rl.py", ["random_permute_actor", "VisActorCritic"])
    with open("/root/reference/data_gen/configs/rl/train/Cartpole.yaml") as f:
        hid = yaml.safe_load(f)["policy"]["pi_hid_sizes"]
    cls = ns["VisActorCritic"]
    net = cls.__new__(cls)
    nn.Module.__init__(net)
    dims = [4] + list(hid) + [1]
    layers = []
    for l in range(len(dims) - 1):
        layers.append(nn.Linear(dims[l], dims[l + 1]))
        if l < len(dims) - 2:
            layers.append(nn.SELU())
    net.actor = nn.Sequential(*layers)
    net.log_std = nn.Parameter(torch.zeros(1))
    return net, ns["random_permute_actor"], hid


def g15_actor_permute():
    torch.manual_seed(15)
    net, fn, hid = reference_actor()
    keys = list(net.state_dict().keys())
    sizes = [v.numel() for v in net.state_dict().values()]
    from gpt_amd.configs import CARTPOLE_NAMES, CARTPOLE_SIZES
    assert keys == CARTPOLE_NAMES and sizes == CARTPOLE_SIZES, (keys, sizes)
    D = sum(sizes)
    ident = torch.arange(D)
    maps = np.stack([random_permute_flat(ident, net, seed, fn).numpy() for seed in range(16)]).astype(np.int32)
    # function preservation (sanity): the permuted flat vector, loaded back, computes the same policy mean
    flat = torch.cat([v.flatten() for v in net.state_dict().values()])
    obs = torch.randn(5, 4)
    want = net.actor(obs)
    perm = flat[torch.from_numpy(maps[3]).long()]
    off, sd = 0, {}
    for k, v in net.state_dict().items():
        sd[k] = perm[off:off + v.numel()].view(v.shape)
        off += v.numel()
    net.load_state_dict(sd)
    assert torch.allclose(net.actor(obs), want, atol=1e-6)
    save("g15_actor_permute", actor=maps, hidden=np.array(hid), sizes=np.array(sizes))


# ------------------------------------------------------------------ G9: one optimisation step
def g9_trainstep(case):
    from copy import deepcopy
    m, plan, sd = ref_model(case)
    ema = deepcopy(m)
    d = create_diffusion(learn_sigma=False, predict_xstart=True, noise_schedule="linear", steps=1000)
    _, _, lt, lp, xp = make_inputs(case, B=6, seed=41)
    g = torch.Generator().manual_seed(42)
    x_start = xp + 0.1 * torch.randn(xp.shape, generator=g)
    t = torch.tensor([3, 100, 250, 600, 900, 999])
    noise = torch.randn(xp.shape, generator=g)
    m.train()
    opt = m.configure_optimizers(lr=4e-4, wd=0.1, betas=(0.9, 0.999))
    n_decay, n_nodecay = len(opt.param_groups[0]["params"]), len(opt.param_groups[1]["params"])
    terms = d.training_losses(m, x_start, t, model_kwargs=dict(loss_target=lt, loss_prev=lp, x_prev=xp), noise=noise)
    loss = (terms["loss"] * torch.ones(6)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), 0.75)
    opt.step()
    # EMA accumulate (Gpt/utils.py:146-152) restated: ema = ema*decay + p*(1-decay)
    with torch.no_grad():
        pe, pm = dict(ema.named_parameters()), dict(m.named_parameters())
        for k in pe:
            pe[k].mul_(0.9999).add_(pm[k], alpha=1 - 0.9999)
    keys = ["decoder.blocks.1.mlp.0.weight", "decoder.blocks.0.attn.query.bias", "decoder.ln_in.weight",
            "decoder.input_parameter_projections.3.0.weight", "decoder.pos_emb"]
    after = {f"param::{k}": slim(pm[k].detach().numpy()) for k in keys}
    after.update({f"ema::{k}": slim(pe[k].detach().numpy()) for k in keys})
    save(f"g9_trainstep_{case}", x_start=x_start.numpy(), t=t.numpy(), noise=noise.numpy(), loss=float(loss),
         total_norm=float(total_norm), n_decay=n_decay, n_nodecay=n_nodecay, **after)


# ------------------------------------------------------------------ G10: functional MLP evaluation
def g10_mlp_eval():
    import data_gen.train_mnist as tm
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(51)
    nets = torch.randn(5, 7960, generator=g) * 0.1
    imgs = torch.randn(64, 1, 28, 28, generator=g)
    labels = torch.randint(0, 10, (64,), generator=g)
    losses, errs = [], []
    for v in nets:
        net = tm.MLP(10)
        sd = net.state_dict()
        off = 0
        new = {}
        for k, p in sd.items():                                   # moduleify (Gpt/vis.py:539-570)
            new[k] = v[off:off + p.numel()].view(p.shape)
            off += p.numel()
        net.load_state_dict(new)
        with torch.no_grad():
            logits = net(imgs)
            losses.append(F.cross_entropy(logits, labels, reduction="mean").item())
            errs.append((logits.argmax(1) != labels).float().mean().item() * 100)
    save("g10_mlp_eval", nets=nets.numpy(), imgs=imgs.numpy(), labels=labels.numpy(),
         loss=np.array(losses), err=np.array(errs))


# ------------------------------------------------------------------ G12: functional ConvNet evaluation
def g12_cnn_eval():
    import data_gen.train_cifar10 as tc
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(52)
    nets = torch.randn(6, 5370, generator=g) * 0.15
    imgs = torch.randn(96, 3, 32, 32, generator=g)
    labels = torch.randint(0, 10, (96,), generator=g)
    losses, errs = [], []
    for v in nets:
        net = tc.ConvNet()
        off, new = 0, {}
        for k, p in net.state_dict().items():                     # moduleify (Gpt/vis.py:539-570)
            new[k] = v[off:off + p.numel()].view(p.shape)
            off += p.numel()
        assert off == 5370
        net.load_state_dict(new)
        with torch.no_grad():
            logits = net.eval()(imgs)
            losses.append(F.cross_entropy(logits, labels, reduction="none").mean().item())   # test_epoch (:426-434)
            errs.append((logits.argmax(1) != labels).float().mean().item() * 100)
    save("g12_cnn_eval", nets=nets.numpy(), imgs=imgs.numpy(), labels=labels.numpy(),
         loss=np.array(losses), err=np.array(errs))


# ------------------------------------------------------------------ G13: checkpoint-pair dataset
def g13_pairs():
    """The reference's ParameterDataset over synthetic run directories.  py-lmdb is absent, so its Database is replaced
    by the dict the synthetic generator returns; Gpt.tasks / Gpt.download (task registry, downloader: they import wandb-era
    packages) are replaced by the two names dataset_lmdb.py uses from them."""
    import tempfile
    import data_gen.train_mnist as tm
    import data_gen.train_cifar10 as tc
    from tests.cases import PAIR_CASES, make_pair_dataset
    tasks = types.ModuleType("Gpt.tasks")
    tasks.TASK_METADATA = {
        "mnist_loss": {"constructor": lambda: tm.MLP(w_h=10), "aug_fn": tm.random_permute_mlp, "minimize": True},
        "cifar10_error": {"constructor": tc.ConvNet, "aug_fn": tc.random_permute_cnn, "minimize": True}}
    dl = types.ModuleType("Gpt.download")
    dl.find_checkpoints = lambda d: None
    sys.modules["Gpt.tasks"], sys.modules["Gpt.download"] = tasks, dl
    import Gpt.data.dataset_lmdb as ref_ds
    out = {}
    for case, c in PAIR_CASES.items():
        with tempfile.TemporaryDirectory() as root:
            dbs = make_pair_dataset(root, case)
            ref_ds.Database = lambda path, map_size=None: dbs[path]
            for split in ("train", "test"):
                ds = ref_ds.ParameterDataset(dataset_dir=root, dataset_name=c["dataset_name"], split=split,
                                             num_test_runs=c["num_test_runs"], train_metric=c["train_metric"],
                                             openai_coeff=c["openai_coeff"], permute_augment=c["augment"], download=False,
                                             target_epoch_size=1000)
                torch.manual_seed(c["seed"] + 1)
                items = [ds[i] for i in c["items"]]
                tag = f"{case}_{split}"
                m = c["train_metric"]
                out[f"{tag}_runs"] = np.array(ds.runs)
                out[f"{tag}_sizes"] = np.array(ds.parameter_sizes)
                out[f"{tag}_scalars"] = np.array([ds.min_val, ds.max_val, ds.optimal_test_loss, float(ds.epoch_size),
                                                  *ds.get_range(normalize=True)])
                out[f"{tag}_key0"] = np.array([it["checkpoint_key_0"] for it in items])
                out[f"{tag}_key1"] = np.array([it["checkpoint_key_1"] for it in items])
                out[f"{tag}_run"] = np.array([it["run_name"] for it in items])
                out[f"{tag}_spacing"] = np.array([it["step_spacing"] for it in items])
                out[f"{tag}_m0"] = np.array([it[f"{m}_0"] for it in items])
                out[f"{tag}_m1"] = np.array([it[f"{m}_1"] for it in items])
                out[f"{tag}_p0"] = np.stack([it["parameters_0"].numpy()[::7] for it in items])
                out[f"{tag}_p1"] = np.stack([it["parameters_1"].numpy()[::7] for it in items])
                out[f"{tag}_net"] = ds.get_run_network(len(ds.runs) - 1, iter=2, normalize=True, augment=False).numpy()[::7]
    save("g13_pairs", **out)


# ------------------------------------------------------------------ G14: remaining GaussianDiffusion surface
def g14_diffusion_misc():
    """ddim_reverse_sample, q_mean_variance, q_posterior_mean_variance and _prior_bpd of the reference on seeded inputs
    (the "model" is a closure returning a fixed tensor: these functions only post-process its output)."""
    out = {}
    g = torch.Generator().manual_seed(77)
    B, D = 4, 301
    x, x0, mo = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g) * 0.5, torch.randn(B, D, generator=g) * 0.7
    out["x"], out["x0"], out["mo"] = x.numpy(), x0.numpy(), mo.numpy()
    for tag, respacing, px, ts in (("full_x0", "", True, [0, 17, 500, 999]), ("r50_eps", "50", False, [0, 3, 25, 49])):
        d = create_diffusion(learn_sigma=False, predict_xstart=px, noise_schedule="linear", steps=1000,
                             timestep_respacing=respacing)
        t = torch.tensor(ts)
        r = d.ddim_reverse_sample(lambda xx, tt, **kw: mo, x, t, clip_denoised=(tag == "r50_eps"))
        out[f"{tag}_t"], out[f"{tag}_rev"], out[f"{tag}_rev_x0"] = t.numpy(), r["sample"].numpy(), r["pred_xstart"].numpy()
        m, v, lv = d.q_mean_variance(x0, t)
        out[f"{tag}_qm"], out[f"{tag}_qv"], out[f"{tag}_qlv"] = m.numpy(), v[:, 0].numpy(), lv[:, 0].numpy()
        m, v, lv = d.q_posterior_mean_variance(x0, x, t)
        out[f"{tag}_pm"], out[f"{tag}_pv"], out[f"{tag}_plv"] = m.numpy(), v[:, 0].numpy(), lv[:, 0].numpy()
        out[f"{tag}_bpd"] = d._prior_bpd(x0).numpy()
    save("g14_diffusion_misc", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        g14_diffusion_misc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g13":          # only the pair-dataset fixture (added after G1-G12)
        g13_pairs()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g12":          # only the ConvNet fixture (added after G1-G11)
        g12_cnn_eval()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cartpole":     # the third shipped task (added in round 3)
        g3_forward("cartpole_small")
        g15_actor_permute()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "g11":          # only the DDIM fixtures (added after G1-G10)
        g11_ddim("mnist_small", "ddim25", "ddim25_eta0", 0.0)
        g11_ddim("mnist_small", "50", "r50_eta05", 0.5)
        g11_ddim("cifar_small", "ddim25", "ddim25_eta1_eps_clip", 1.0, predict_xstart=False, clip=True)
        sys.exit(0)
    g1_splits()
    g2_freq()
    for case in ("mnist_small", "cifar_small", "cartpole_small"):
        g3_forward(case)
    g4_block()
    g5_tables()
    g6_training("mnist_small", True)
    g6_training("mnist_small", False)
    g6_training("cifar_small", True)
    g7_sampling("mnist_small", "", "full")
    g7_sampling("mnist_small", "50", "r50")
    g7_sampling("cifar_small", "50", "r50")
    g7_sampling("mnist_small", "50", "r50_eps_clip", predict_xstart=False, clip=True)
    g8_permute()
    g9_trainstep("mnist_small")
    g10_mlp_eval()
    g11_ddim("mnist_small", "ddim25", "ddim25_eta0", 0.0)
    g11_ddim("mnist_small", "50", "r50_eta05", 0.5)
    g11_ddim("cifar_small", "ddim25", "ddim25_eta1_eps_clip", 1.0, predict_xstart=False, clip=True)
    g12_cnn_eval()
    g13_pairs()
    g14_diffusion_misc()
    g15_actor_permute()
