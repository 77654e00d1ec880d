"""Host-side diffusion schedule tables (numpy float64), candidate re-spacing.

Mirrors, as plain host logic (no device work):
  * get_named_beta_schedule / betas_for_alpha_bar -- reference
    guided_diffusion/gaussian_diffusion.py:18-62
  * the 13 coefficient tables of GaussianDiffusion.__init__ -- ibid. :118-169
  * space_timesteps -- reference guided_diffusion/respace.py:7-60
  * SpacedDiffusion.__init__'s subset re-derivation -- respace.py:71-85
  * EvolutionSearcher.reset_diffusion -- reference
    search_imagenet64_classifier_guidance.py:200-255 (apply_candidate below)

The tables stay float64 numpy attributes of the diffusion object exactly as in
the reference, so the search drivers may keep overwriting them in place; the
sampler converts the handful of scalars it needs per step (float64 -> float32,
as _extract_into_tensor does) and passes them to the HIP kernel by value, so
there is no device-side cache to invalidate.
"""
from __future__ import annotations

import enum
import math
from typing import Iterable, List, Sequence, Tuple

import numpy as np


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


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    t = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / t) / alpha_bar(i / t), max_beta) for i in range(t)])


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    if schedule_name == "linear":
        scale = 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def space_timesteps(num_timesteps, section_counts):
    """Set of original timesteps to keep ("ddimN" = first integer stride giving N steps)."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, count in enumerate(section_counts):
        size = per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            steps.append(start + round(cur))
            cur += frac
        start += size
    return set(steps)


TABLES = (
    "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def install_tables(obj, betas: np.ndarray, allow_single_step: bool) -> None:
    """Set betas, num_timesteps and the 12 derived float64 arrays as attributes of `obj`."""
    betas = np.array(betas, dtype=np.float64)
    assert len(betas.shape) == 1, "betas must be 1-D"
    assert (betas > 0).all() and (betas <= 1).all()
    obj.betas = betas
    obj.num_timesteps = int(betas.shape[0])
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    obj.alphas_cumprod = ac
    obj.alphas_cumprod_prev = np.append(1.0, ac[:-1])
    obj.alphas_cumprod_next = np.append(ac[1:], 0.0)
    obj.sqrt_alphas_cumprod = np.sqrt(ac)
    obj.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - ac)
    obj.log_one_minus_alphas_cumprod = np.log(1.0 - ac)
    obj.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
    obj.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)
    pv = betas * (1.0 - obj.alphas_cumprod_prev) / (1.0 - ac)
    obj.posterior_variance = pv
    if len(pv) > 1 or not allow_single_step:
        obj.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:]))
    else:
        # reset_diffusion's K == 1 branch keeps the raw variance (search script :242-247)
        obj.posterior_log_variance_clipped = pv
    obj.posterior_mean_coef1 = betas * np.sqrt(obj.alphas_cumprod_prev) / (1.0 - ac)
    obj.posterior_mean_coef2 = (1.0 - obj.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - ac)


def subset_betas(base_alphas_cumprod: Sequence[float], use_timesteps: Iterable[int]) -> Tuple[np.ndarray, List[int]]:
    keep = set(use_timesteps)
    last, betas, tmap = 1.0, [], []
    for i, a in enumerate(base_alphas_cumprod):
        if i in keep:
            betas.append(1 - a / last)
            last = a
            tmap.append(i)
    return np.array(betas, dtype=np.float64), tmap


def apply_candidate(active, base, use_timesteps: Iterable[int]) -> None:
    """reset_diffusion: re-derive every table of `active` for a searched subset of `base`'s steps."""
    use = set(use_timesteps)
    betas, tmap = subset_betas(base.alphas_cumprod, use)
    active.use_timesteps = set(use)
    active.timestep_map = tmap
    install_tables(active, betas, allow_single_step=True)
