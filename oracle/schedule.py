"""Oracle: diffusion schedule tables in numpy float64 (TEST INFRASTRUCTURE).

Restates, in its own words:
  * ``get_named_beta_schedule`` / ``betas_for_alpha_bar``
    (reference guided_diffusion/gaussian_diffusion.py:18-62),
  * the coefficient tables of ``GaussianDiffusion.__init__`` (ibid. :118-169),
  * ``space_timesteps`` (reference guided_diffusion/respace.py:7-60),
  * the subset re-derivation of ``SpacedDiffusion.__init__`` (respace.py:71-85)
    and ``EvolutionSearcher.reset_diffusion``
    (reference search_imagenet64_classifier_guidance.py:200-255), including the
    K == 1 quirk where ``posterior_log_variance_clipped`` is the raw variance.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List

import numpy as np

TABLE_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
)


def named_betas(name: str, steps: int) -> np.ndarray:
    if name == "linear":
        k = 1000.0 / steps
        return np.linspace(k * 1e-4, k * 2e-2, steps, dtype=np.float64)
    if name == "cosine":
        def abar(u):
            return math.cos((u + 0.008) / 1.008 * math.pi / 2.0) ** 2
        out = np.empty(steps, dtype=np.float64)
        for i in range(steps):
            out[i] = min(1.0 - abar((i + 1) / steps) / abar(i / steps), 0.999)
        return out
    raise NotImplementedError(f"unknown beta schedule: {name}")


def spaced_steps(num_timesteps: int, section_counts) -> set:
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(s) for s in section_counts.split(",")]
    base, extra = divmod(num_timesteps, len(section_counts))
    picked: List[int] = []
    start = 0
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            picked.append(start + round(pos))
            pos += stride
        start += size
    return set(picked)


def tables_from_betas(betas: np.ndarray, log_clip_quirk: bool = False) -> Dict[str, np.ndarray]:
    """All 13 float64 tables for one beta sequence.

    ``log_clip_quirk`` reproduces reset_diffusion's K == 1 branch (the search
    scripts store the raw posterior variance instead of its log).  The base
    class constructor would index element 1 and fail, so K == 1 only exists on
    the reset path.
    """
    betas = np.asarray(betas, dtype=np.float64)
    assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    ac_next = np.append(ac[1:], 0.0)
    pv = betas * (1.0 - ac_prev) / (1.0 - ac)
    if len(pv) > 1:
        plv = np.log(np.append(pv[1], pv[1:]))
    elif log_clip_quirk:
        plv = pv
    else:
        raise IndexError("posterior_log_variance_clipped needs at least 2 steps")
    return {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "alphas_cumprod_next": ac_next,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": plv,
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }


def subset_betas(base_alphas_cumprod: np.ndarray, use_timesteps: Iterable[int]):
    """(new_betas, timestep_map) for a *set* of original timesteps."""
    keep = set(int(t) for t in use_timesteps)
    last = 1.0
    new_betas, tmap = [], []
    for i, a in enumerate(base_alphas_cumprod):
        if i in keep:
            new_betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return np.array(new_betas, dtype=np.float64), tmap


class OracleDiffusion:
    """Minimal stand-in for SpacedDiffusion: tables + timestep_map + flags."""

    def __init__(self, *, steps=1000, noise_schedule="linear", timestep_respacing="",
                 learn_sigma=False, sigma_small=False, predict_xstart=False,
                 rescale_timesteps=False):
        self.base_betas = named_betas(noise_schedule, steps)
        self.base_alphas_cumprod = np.cumprod(1.0 - self.base_betas, axis=0)
        self.original_num_steps = steps
        self.predict_xstart = predict_xstart
        self.var_type = "learned_range" if learn_sigma else ("fixed_small" if sigma_small else "fixed_large")
        self.rescale_timesteps = rescale_timesteps
        use = spaced_steps(steps, timestep_respacing or [steps])
        nb, self.timestep_map = subset_betas(self.base_alphas_cumprod, use)
        self.use_timesteps = set(use)
        self.tables = tables_from_betas(nb)
        self.num_timesteps = len(nb)

    def reset(self, cand: Iterable[int]):
        """reset_diffusion(cand): rebuild every table for a searched subset."""
        nb, self.timestep_map = subset_betas(self.base_alphas_cumprod, cand)
        self.use_timesteps = set(int(c) for c in cand)
        self.tables = tables_from_betas(nb, log_clip_quirk=True)
        self.num_timesteps = len(nb)
        return self
