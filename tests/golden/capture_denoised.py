#!/usr/bin/env python3
"""Capture `denoised_fn` golden vectors by importing the REFERENCE's sampler (build container only).

`p_mean_variance` applies `denoised_fn` to the predicted x_0 before clipping (gaussian_diffusion.py:293-298); no launch script of
the reference passes one, so the fixture drives `ddim_sample` / `p_sample` / the two loops directly on the small 64x64 golden
model + classifier (tests/golden/capture_golden.py: cfg_m64, cfg_c64), with the deterministic function below -- restated in
tests/helpers.py (`denoised_fn_fixture`) for the test side.  Only inputs / outputs are stored.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_denoised.py
"""
import copy
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import capture_golden as cg  # noqa: E402  (imports the reference's modules)


def denoised_fn(x):
    return 0.8 * x + 0.25 * torch.tanh(3.0 * x)


def main():
    m, base = cg.build(cg.cfg_m64(dynamic=True))
    clf = cg.create_classifier(**cg.cfg_c64())
    cg.fill_module(clf)
    clf.eval()
    y = torch.tensor([977])

    def cond_fn(x, t, y=None, **kw):
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            lp = F.log_softmax(clf(x_in, t), dim=-1)
            return torch.autograd.grad(lp[range(len(lp)), y.view(-1)].sum(), x_in)[0]

    def model_fn(x, t, y=None, **kw):
        return m(x, t, y)

    drv = cg.import_search_driver()
    s = object.__new__(drv.EvolutionSearcher)
    s.base_diffusion, s.active_diffusion = base, copy.deepcopy(base)
    cand = [153, 424, 926, 690]
    s.reset_diffusion(cand)
    diff = s.active_diffusion
    x = cg.rnd((1, 3, 64, 64), 51)
    out = {"x": x.numpy(), "y": y.numpy(), "cand": np.array(cand)}
    for idx in (2, 0):
        t = torch.tensor([idx])
        for guided in (False, True):
            cf = cond_fn if guided else None
            tag = f"i{idx}_{'g' if guided else 'u'}"
            for clip in ((True, False) if (idx == 2 and guided) else (True,)):
                ctag = tag + ("" if clip else "_noclip")
                torch.manual_seed(100 + idx)
                with torch.no_grad():
                    o = diff.ddim_sample(model_fn, x, t, clip_denoised=clip, denoised_fn=denoised_fn, cond_fn=cf, model_kwargs={"y": y}, eta=0.3)
                out[f"ddim_{ctag}_sample"], out[f"ddim_{ctag}_x0"] = o["sample"].numpy(), o["pred_xstart"].numpy()
                torch.manual_seed(100 + idx)
                with torch.no_grad():
                    o = diff.p_sample(model_fn, x, t, clip_denoised=clip, denoised_fn=denoised_fn, cond_fn=cf, model_kwargs={"y": y})
                out[f"ddpm_{ctag}_sample"], out[f"ddpm_{ctag}_x0"] = o["sample"].numpy(), o["pred_xstart"].numpy()
        torch.manual_seed(100 + idx)
        out[f"noise_i{idx}"] = torch.randn_like(x).numpy()
    for name, fn in (("ddim", diff.ddim_sample_loop), ("ddpm", diff.p_sample_loop)):
        torch.manual_seed(7)
        smp = fn(model_fn, (1, 3, 64, 64), noise=x, clip_denoised=True, denoised_fn=denoised_fn, model_kwargs={"y": y}, cond_fn=cond_fn,
                 device=torch.device("cpu"))
        out[f"{name}_loop_sample"] = smp.numpy()
    torch.manual_seed(7)
    out["noises"] = np.stack([torch.randn_like(x).numpy() for _ in range(4)])
    cg.save("sampler_denoised_m64", **out)


if __name__ == "__main__":
    main()
