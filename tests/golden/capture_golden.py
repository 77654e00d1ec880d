#!/usr/bin/env python3
"""Capture golden vectors by importing the REFERENCE's own Python modules.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference.  Outputs small .npz fixtures next to this script.  Only
inputs / expected outputs are stored -- weights are regenerated on both sides
from the deterministic fill rule in ``oracle/fill.py``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_golden.py

Two of the reference's unused imports are absent from this image
(``torchvision.transforms`` -- imported but never used by the search script --
and ``blobfile`` -- used only by checkpoint reading); empty stand-in modules
are registered for them *in this capture process only* so that the search
driver module can be imported to record ``reset_diffusion`` and the EA
trajectory.  ``EvolutionSearcher.__init__`` needs TensorFlow, so instances are
created with ``object.__new__`` and the attributes the recorded methods read.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/examples/guided_diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle.fill import fill_array  # noqa: E402

from guided_diffusion import gaussian_diffusion as gd  # noqa: E402
from guided_diffusion.nn import GroupNorm32, timestep_embedding  # noqa: E402
from guided_diffusion.respace import space_timesteps  # noqa: E402
from guided_diffusion.script_util import (classifier_defaults, create_classifier,  # noqa: E402
                                          create_gaussian_diffusion, create_model_and_diffusion,
                                          model_and_diffusion_defaults)
from guided_diffusion.unet import (AttentionBlock, QKVAttention, QKVAttentionLegacy,  # noqa: E402
                                   ResBlock)

torch.set_num_threads(8)


def fill_module(mod, prefix=""):
    with torch.no_grad():
        for k, v in mod.state_dict().items():
            v.copy_(torch.from_numpy(fill_array(prefix + k, tuple(v.shape))))


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={sorted(arrs)}")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ---------------------------------------------------------------- model configs
def cfg_m32(dynamic=True, legacy=False):
    d = model_and_diffusion_defaults()
    d.update(image_size=32, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
             attention_resolutions="16,8", num_head_channels=(-1 if legacy else 32),
             num_heads=(2 if legacy else 4), class_cond=True, learn_sigma=True,
             resblock_updown=True, use_scale_shift_norm=True,
             use_new_attention_order=not legacy, use_dynamic_unet=dynamic,
             noise_schedule="cosine")
    return d


def cfg_m64(dynamic=False):
    d = model_and_diffusion_defaults()
    d.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
             attention_resolutions="16", num_head_channels=32, class_cond=True,
             learn_sigma=True, resblock_updown=True, use_scale_shift_norm=True,
             use_new_attention_order=True, use_dynamic_unet=dynamic, noise_schedule="cosine")
    return d


def cfg_c64():
    d = classifier_defaults()
    d.update(image_size=64, classifier_width=64, classifier_depth=1)
    return d


# ---------------------------------------------------------------- 1-3: schedules
def cap_schedules():
    out = {}
    for key in ("ddim4", "ddim10", "4", "10,15", "25"):
        out["space_" + key.replace(",", "_")] = np.array(sorted(space_timesteps(1000, key)), dtype=np.int64)
    out["betas_cosine"] = gd.get_named_beta_schedule("cosine", 1000)
    out["betas_linear"] = gd.get_named_beta_schedule("linear", 1000)
    save("schedules", **out)

    # SpacedDiffusion tables for a respaced constructor (not the reset path)
    diff = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="cosine",
                                     timestep_respacing="ddim4")
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
             "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
             "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    out = {n: getattr(diff, n) for n in names}
    out["timestep_map"] = np.array(diff.timestep_map, dtype=np.int64)
    save("spaced_ddim4_cosine", **out)
    return names


def import_search_driver():
    for missing in ("torchvision", "torchvision.transforms", "blobfile"):
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import search_imagenet64_classifier_guidance as drv
    return drv


def cap_reset(names, drv):
    import copy
    for sched in ("cosine", "linear"):
        base = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule=sched)
        s = object.__new__(drv.EvolutionSearcher)
        s.base_diffusion = base
        s.active_diffusion = copy.deepcopy(base)
        cands = {"k4": [153, 424, 926, 690], "k6": [94, 834, 217, 944, 574, 354],
                 "k10": [3, 77, 140, 251, 333, 480, 611, 702, 850, 999], "k1": [500]}
        out = {}
        for tag, cand in cands.items():
            s.reset_diffusion(cand)
            a = s.active_diffusion
            out[f"{tag}_cand"] = np.array(cand, dtype=np.int64)
            out[f"{tag}_timestep_map"] = np.array(a.timestep_map, dtype=np.int64)
            out[f"{tag}_num_timesteps"] = np.array(a.num_timesteps)
            for n in names:
                out[f"{tag}_{n}"] = getattr(a, n)
        save(f"reset_diffusion_{sched}", **out)


# ---------------------------------------------------------------- 4-6: ops
def cap_ops():
    t = torch.tensor([0, 1, 250, 999])
    save("timestep_embedding", t=t.numpy(), dim192=timestep_embedding(t, 192).numpy(),
         dim32=timestep_embedding(t, 32).numpy(), dim33=timestep_embedding(t, 33).numpy())

    gn = GroupNorm32(32, 64)
    fill_module(gn, "gn.")
    x = rnd((2, 64, 8, 8), 11, 2.0) + 0.5
    save("groupnorm32", x=x.numpy(), y=gn(x).detach().numpy())

    qkv = rnd((2, 3 * 2 * 32, 64), 12)
    save("qkv_attention", qkv=qkv.numpy(), heads=np.array(2),
         new=QKVAttention(2)(qkv).numpy(), legacy=QKVAttentionLegacy(2)(qkv).numpy())
    qkv = rnd((2, 3 * 1 * 64, 256), 13)
    save("qkv_attention_d64", qkv=qkv.numpy(), heads=np.array(1),
         new=QKVAttention(1)(qkv).numpy(), legacy=QKVAttentionLegacy(1)(qkv).numpy())

    emb = rnd((2, 128), 14)
    out = {"emb": emb.numpy()}
    for tag, kw, cin, hw in (("plain", dict(), 64, 16), ("down", dict(down=True), 64, 16),
                             ("up", dict(up=True), 64, 8), ("skipconv", dict(out_channels=96), 64, 16),
                             ("noss", dict(use_scale_shift_norm=False), 64, 16)):
        kw.setdefault("use_scale_shift_norm", True)
        rb = ResBlock(cin, 128, 0.0, **kw).eval()
        fill_module(rb, f"rb_{tag}.")
        x = rnd((2, cin, hw, hw), 20 + len(tag))
        out[f"{tag}_x"] = x.numpy()
        with torch.no_grad():
            out[f"{tag}_y"] = rb(x, emb).numpy()
    save("resblock", **out)

    out = {}
    for tag, kw in (("new", dict(num_head_channels=32, use_new_attention_order=True)),
                    ("legacy", dict(num_heads=2, use_new_attention_order=False))):
        ab = AttentionBlock(64, **kw).eval()
        fill_module(ab, f"attn_{tag}.")
        x = rnd((2, 64, 8, 8), 31)
        out[f"{tag}_x"] = x.numpy()
        with torch.no_grad():
            out[f"{tag}_y"] = ab(x).detach().numpy()
    save("attention_block", **out)


# ---------------------------------------------------------------- 7: networks
def build(cfgd):
    m, d = create_model_and_diffusion(**cfgd)
    fill_module(m)
    return m.eval(), d


def cap_nets():
    x = rnd((2, 3, 32, 32), 41)
    t = torch.tensor([37, 901])
    y = torch.tensor([3, 977])
    m, _ = build(cfg_m32(dynamic=True))
    out = {"x": x.numpy(), "t": t.numpy(), "y": y.numpy(), "layer_num": np.array(m.layer_num)}
    skips = {"none": [], "a": [1, 4], "b": [0, 2, 5, 9, 13, 17, 21], "all": list(range(m.layer_num))}
    with torch.no_grad():
        for tag, sk in skips.items():
            out[f"skip_{tag}"] = np.array(sk, dtype=np.int64)
            out[f"out_{tag}"] = m(x, t, y, skip_layer=sk).numpy()
    save("unet_m32", **out)

    m, _ = build(cfg_m32(dynamic=False, legacy=True))
    with torch.no_grad():
        save("unet_m32_legacy", x=x.numpy(), t=t.numpy(), y=y.numpy(), out=m(x, t, y).numpy())

    x = rnd((2, 3, 64, 64), 42)
    m, _ = build(cfg_m64())
    with torch.no_grad():
        save("unet_m64", x=x.numpy(), t=t.numpy(), y=y.numpy(), out=m(x, t, y).numpy())

    clf = create_classifier(**cfg_c64())
    fill_module(clf)
    clf.eval()
    xin = x.clone().requires_grad_(True)
    logits = clf(xin, t)
    logp = torch.nn.functional.log_softmax(logits, dim=-1)
    sel = logp[range(2), y.view(-1)]
    g = torch.autograd.grad(sel.sum(), xin)[0]
    save("classifier_c64", x=x.numpy(), t=t.numpy(), y=y.numpy(),
         logits=logits.detach().numpy(), grad=g.numpy())


# ---------------------------------------------------------------- 8-9: steps + loops
def cap_sampling(drv):
    import copy
    import torch.nn.functional as F
    m, base = build(cfg_m64(dynamic=True))
    clf = create_classifier(**cfg_c64())
    fill_module(clf)
    clf.eval()
    y = torch.tensor([3, 977])
    scale = 1.0

    def cond_fn(x, t, y=None, **kw):
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            logits = clf(x_in, t)
            lp = F.log_softmax(logits, dim=-1)
            sel = lp[range(len(logits)), y.view(-1)]
            return torch.autograd.grad(sel.sum(), x_in)[0] * scale

    def model_fn(x, t, y=None, **kw):
        return m(x, t, y)

    s = object.__new__(drv.EvolutionSearcher)
    s.base_diffusion = base
    s.active_diffusion = copy.deepcopy(base)
    cand = [153, 424, 926, 690]
    s.reset_diffusion(cand)
    diff = s.active_diffusion
    x = rnd((2, 3, 64, 64), 51)
    out = {"x": x.numpy(), "y": y.numpy(), "cand": np.array(cand)}
    # single steps at index 2 and 0, with and without guidance
    for idx in (2, 0):
        t = torch.tensor([idx, idx])
        for guided in (False, True):
            cf = cond_fn if guided else None
            tag = f"i{idx}_{'g' if guided else 'u'}"
            torch.manual_seed(100 + idx)
            with torch.no_grad():
                o = diff.ddim_sample(model_fn, x, t, cond_fn=cf, model_kwargs={"y": y})
            out[f"ddim_{tag}_sample"] = o["sample"].numpy()
            out[f"ddim_{tag}_x0"] = o["pred_xstart"].numpy()
            torch.manual_seed(100 + idx)
            with torch.no_grad():
                o = diff.ddim_sample(model_fn, x, t, cond_fn=cf, model_kwargs={"y": y}, eta=0.7)
            out[f"ddim_eta_{tag}_sample"] = o["sample"].numpy()
            torch.manual_seed(100 + idx)
            with torch.no_grad():
                o = diff.p_sample(model_fn, x, t, cond_fn=cf, model_kwargs={"y": y})
            out[f"ddpm_{tag}_sample"] = o["sample"].numpy()
            out[f"ddpm_{tag}_x0"] = o["pred_xstart"].numpy()
            torch.manual_seed(100 + idx)
            out[f"noise_i{idx}"] = torch.randn_like(x).numpy()
    with torch.no_grad():
        out["model_out_i2"] = model_fn(x, torch.tensor([690, 690]), y).numpy()
    save("sampler_steps_m64", **out)

    # full loops, injected start noise, per-step noise = successive randn_like draws
    out = {"x_T": x.numpy(), "y": y.numpy(), "cand": np.array(cand)}
    for name, fn in (("ddim", diff.ddim_sample_loop), ("ddpm", diff.p_sample_loop)):
        for guided in (False, True):
            tag = f"{name}_{'g' if guided else 'u'}"
            torch.manual_seed(7)
            smp = fn(model_fn, (2, 3, 64, 64), noise=x, clip_denoised=True,
                     model_kwargs={"y": y}, cond_fn=cond_fn if guided else None,
                     device=torch.device("cpu"))
            out[f"{tag}_sample"] = smp.numpy()
            u8 = ((smp + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            out[f"{tag}_uint8"] = u8.numpy()
    torch.manual_seed(7)
    out["noises"] = np.stack([torch.randn_like(x).numpy() for _ in range(4)])
    # layer-skip candidate through the dynamic search script's model_fn convention
    skip_layers = [[1], [], [0, 5], [2, 3]]

    def model_fn_skip(x, t, y=None, skip_layers=None, **kw):
        sl = skip_layers[diff.timestep_map.index(int(t[0]))]
        return m(x, t, y, skip_layer=sl)

    def cond_fn_skip(x, t, y=None, skip_layers=None, **kw):
        return cond_fn(x, t, y=y)

    torch.manual_seed(7)
    smp = diff.ddim_sample_loop(model_fn_skip, (2, 3, 64, 64), noise=x, clip_denoised=True,
                                model_kwargs={"y": y, "skip_layers": skip_layers},
                                cond_fn=cond_fn_skip, device=torch.device("cpu"))
    out["ddim_g_skip_sample"] = smp.numpy()
    out["skip_layers"] = np.array([",".join(map(str, s_)) for s_ in skip_layers])
    save("sampler_loops_m64", **out)

    # unconditional model, uniform ddim4 respacing (BASELINE config 1 in miniature)
    d = cfg_m32(dynamic=False)
    d.update(class_cond=False, timestep_respacing="ddim4")
    m2, diff2 = build(d)
    x2 = rnd((2, 3, 32, 32), 52)
    out = {"x_T": x2.numpy()}
    for name, fn in (("ddim", diff2.ddim_sample_loop), ("ddpm", diff2.p_sample_loop)):
        torch.manual_seed(9)
        out[f"{name}_sample"] = fn(m2, (2, 3, 32, 32), noise=x2, clip_denoised=True,
                                   model_kwargs={}).numpy()
    torch.manual_seed(9)
    out["noises"] = np.stack([torch.randn_like(x2).numpy() for _ in range(4)])
    save("sampler_loops_m32_uncond", **out)


# ---------------------------------------------------------------- 10: EA trajectory
def cap_ea(drv):
    base = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="cosine")

    class A:
        pass
    args = A()
    args.max_epochs, args.select_num, args.population_num = 3, 4, 10
    args.m_prob, args.crossover_num, args.mutation_num = 0.25, 3, 5
    args.max_fid, args.thres = 48.0, 0.2
    args.use_ddim_init_x, args.use_ddim, args.time_step = True, True, 4
    drv.args = args

    class _Log:
        @staticmethod
        def log(*a, **k):
            pass
    drv.logger = _Log

    s = object.__new__(drv.EvolutionSearcher)
    s.args, s.base_diffusion, s.time_step = args, base, 4
    s.max_epochs, s.select_num, s.population_num = args.max_epochs, args.select_num, args.population_num
    s.m_prob, s.crossover_num, s.mutation_num = args.m_prob, args.crossover_num, args.mutation_num
    s.keep_top_k = {s.select_num: [], 50: []}
    s.epoch, s.candidates, s.vis_dict = 0, [], {}
    s.max_fid, s.thres, s.search_space = args.max_fid, args.thres, None
    evaluated = []

    def fitness(cand=None, args=None):
        evaluated.append(list(cand))
        c = np.sort(np.array(cand, dtype=np.float64))
        return float(np.abs(c - np.array([150., 420., 690., 930.])).sum() / 10.0)
    s.get_cand_fid = fitness
    random.seed(0)
    np.random.seed(0)
    s.search()
    top = s.keep_top_k[50]
    save("ea_trajectory", evaluated=np.array(evaluated, dtype=np.int64),
         final_candidates=np.array(s.candidates),
         top50=np.array(top), top50_fid=np.array([s.vis_dict[c]["fid"] for c in top]))
    print("EA evaluations:", len(evaluated), "first three:", evaluated[:3])


if __name__ == "__main__":
    names = cap_schedules()
    drv = import_search_driver()
    cap_reset(names, drv)
    cap_ops()
    cap_nets()
    cap_sampling(drv)
    cap_ea(drv)
