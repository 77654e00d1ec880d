"""Oracle: per-step sampler arithmetic and the two sample loops (TEST INFRASTRUCTURE).

float32 torch-CPU restatement of
  * ``p_mean_variance`` -- reference guided_diffusion/gaussian_diffusion.py:232-326
  * ``condition_mean`` / ``condition_score`` -- ibid. :356-393
  * ``p_sample`` / ``ddim_sample`` -- ibid. :395-439 / :536-584
  * ``p_sample_loop`` / ``ddim_sample_loop`` -- ibid. :441-534 / :624-716
  * ``_WrappedModel`` timestep mapping -- reference guided_diffusion/respace.py:115-127
  * the uint8 NHWC pack -- reference search_imagenet64_classifier_guidance.py:352-354

Noise is always *injected* (``noises[k]`` for the k-th executed step) so that the
oracle and the HIP path can be compared on identical draws.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np
import torch

from .schedule import OracleDiffusion


def _coef(table: np.ndarray, i: int) -> torch.Tensor:
    # _extract_into_tensor: float64 table -> float32 scalar (gaussian_diffusion.py:910-923)
    return torch.tensor(float(table[i]), dtype=torch.float64).float()


def _mapped_t(diff: OracleDiffusion, i: int, n: int) -> torch.Tensor:
    t = torch.full((n,), diff.timestep_map[i], dtype=torch.int64)
    if diff.rescale_timesteps:
        return t.float() * (1000.0 / diff.original_num_steps)
    return t


def mean_variance(diff: OracleDiffusion, model_out: torch.Tensor, x: torch.Tensor, i: int,
                  clip_denoised: bool = True, denoised_fn: Optional[Callable] = None):
    """model_out [N, C or 2C, H, W] -> dict(mean, variance, log_variance, pred_xstart)."""
    T = diff.tables
    C = x.shape[1]
    if diff.var_type == "learned_range":
        assert model_out.shape[1] == 2 * C
        eps_or_x0, v = torch.split(model_out, C, dim=1)
        lo = _coef(T["posterior_log_variance_clipped"], i)
        hi = _coef(np.log(T["betas"]), i)
        frac = (v + 1) / 2
        logvar = frac * hi + (1 - frac) * lo
        var = torch.exp(logvar)
    else:
        eps_or_x0 = model_out
        if diff.var_type == "fixed_large":
            vtab = np.append(T["posterior_variance"][1], T["betas"][1:])
            ltab = np.log(vtab)
        else:
            vtab, ltab = T["posterior_variance"], T["posterior_log_variance_clipped"]
        var = _coef(vtab, i).expand(x.shape)
        logvar = _coef(ltab, i).expand(x.shape)
    if diff.predict_xstart:
        x0 = eps_or_x0
    else:
        x0 = _coef(T["sqrt_recip_alphas_cumprod"], i) * x - _coef(T["sqrt_recipm1_alphas_cumprod"], i) * eps_or_x0
    if denoised_fn is not None:   # process_xstart: denoised_fn first, then the clip (gaussian_diffusion.py:293-298)
        x0 = denoised_fn(x0)
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    mean = _coef(T["posterior_mean_coef1"], i) * x0 + _coef(T["posterior_mean_coef2"], i) * x
    return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": x0}


def _eps_from_x0(diff, x, i, x0):
    T = diff.tables
    return (_coef(T["sqrt_recip_alphas_cumprod"], i) * x - x0) / _coef(T["sqrt_recipm1_alphas_cumprod"], i)


def ddim_step(diff: OracleDiffusion, model_out, x, i, grad=None, noise=None, eta=0.0,
              clip_denoised=True, denoised_fn=None):
    T = diff.tables
    out = mean_variance(diff, model_out, x, i, clip_denoised, denoised_fn)
    x0 = out["pred_xstart"]
    ab = _coef(T["alphas_cumprod"], i)
    if grad is not None:  # condition_score: x0 is NOT re-clamped
        eps = _eps_from_x0(diff, x, i, x0)
        eps = eps - (1 - ab).sqrt() * grad
        x0 = _coef(T["sqrt_recip_alphas_cumprod"], i) * x - _coef(T["sqrt_recipm1_alphas_cumprod"], i) * eps
    eps = _eps_from_x0(diff, x, i, x0)
    ab_prev = _coef(T["alphas_cumprod_prev"], i)
    sigma = eta * torch.sqrt((1 - ab_prev) / (1 - ab)) * torch.sqrt(1 - ab / ab_prev)
    mean_pred = x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev - sigma ** 2) * eps
    if noise is None:
        noise = torch.zeros_like(x)
    nz = 0.0 if i == 0 else 1.0
    return {"sample": mean_pred + nz * sigma * noise, "pred_xstart": x0}


def ddpm_step(diff: OracleDiffusion, model_out, x, i, grad=None, noise=None, clip_denoised=True, denoised_fn=None):
    out = mean_variance(diff, model_out, x, i, clip_denoised, denoised_fn)
    mean = out["mean"]
    if grad is not None:  # condition_mean
        mean = mean.float() + out["variance"] * grad.float()
    if noise is None:
        noise = torch.zeros_like(x)
    nz = 0.0 if i == 0 else 1.0
    return {"sample": mean + nz * torch.exp(0.5 * out["log_variance"]) * noise,
            "pred_xstart": out["pred_xstart"]}


def sample_loop(diff: OracleDiffusion, model_fn: Callable, x_T: torch.Tensor, *, use_ddim: bool,
                cond_fn: Optional[Callable] = None, noises: Optional[List[torch.Tensor]] = None,
                model_kwargs: Optional[dict] = None, clip_denoised: bool = True, eta: float = 0.0,
                return_all: bool = False, denoised_fn: Optional[Callable] = None):
    """model_fn(x, t_mapped, **kw) -> model_out;  cond_fn(x, t_mapped, **kw) -> grad."""
    kw = model_kwargs or {}
    img = x_T
    trail = [img]
    K = diff.num_timesteps
    for k, i in enumerate(reversed(range(K))):
        t = _mapped_t(diff, i, img.shape[0])
        with torch.no_grad():
            mo = model_fn(img, t, **kw)
        g = cond_fn(img, t, **kw) if cond_fn is not None else None
        nz = noises[k] if noises is not None else None
        with torch.no_grad():
            if use_ddim:
                img = ddim_step(diff, mo, img, i, g, nz, eta, clip_denoised, denoised_fn)["sample"]
            else:
                img = ddpm_step(diff, mo, img, i, g, nz, clip_denoised, denoised_fn)["sample"]
        trail.append(img)
    return trail if return_all else img


def pack_uint8_nhwc(sample: torch.Tensor) -> torch.Tensor:
    """((s+1)*127.5).clamp(0,255).to(uint8) (truncation) then NCHW -> NHWC."""
    return ((sample + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
