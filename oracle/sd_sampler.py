"""Oracle: the latent DDIM / PLMS samplers with a searched timestep list (TEST INFRASTRUCTURE -- never imported by
the product).  Plain torch-CPU float32 restatement of

  * ``make_beta_schedule`` / ``make_ddim_timesteps`` / ``make_ddim_sampling_parameters``
    -- reference "Stable Diffusion"/ldm/modules/diffusionmodules/util.py:21-43, 46-63, 66-78
  * ``LatentDiffusion.register_schedule`` tables -- ldm/models/diffusion/ddpm.py:117-137
  * ``DDIMSampler.sample / ddim_sampling / p_sample_ddim`` -- ldm/models/diffusion/ddim.py:60-203
  * ``PLMSSampler.sample / plms_sampling / p_sample_plms`` -- ldm/models/diffusion/plms.py:63-258

pinned by tests/golden/sd_samplers.npz (captured by running the reference's sampler classes, capture_sd_samplers.py).
``apply_model(x, t, c)`` is any callable.
"""
from __future__ import annotations

import numpy as np
import torch


def beta_schedule(schedule="linear", n=1000, linear_start=0.00085, linear_end=0.0120):
    if schedule != "linear":
        raise ValueError(schedule)
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=torch.float64) ** 2).numpy()


def alphas_cumprod_f32(**kw):
    return torch.tensor(np.cumprod(1.0 - beta_schedule(**kw), axis=0), dtype=torch.float32)


def uniform_timesteps(num_ddim, num_ddpm=1000):
    c = round(num_ddpm / num_ddim)
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def sampling_parameters(ac: torch.Tensor, steps, eta: float):
    steps = [int(s) for s in steps]
    a = ac[steps]
    a_prev = torch.cat([ac[:1], ac[steps[:-1]]])
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return sig, a, a_prev


def _guided(apply_model, x, t, c, uc, scale):
    if uc is None or scale == 1.0:
        return apply_model(x, t, c)
    eu, ec = apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uc, c])).chunk(2)
    return eu + scale * (ec - eu)


def _update(x, e, a, a_prev, sig, noise):
    x0 = (x - torch.sqrt(1 - a) * e) / torch.sqrt(a)
    xp = torch.sqrt(a_prev) * x0 + torch.sqrt(1.0 - a_prev - sig ** 2) * e
    if noise is not None:
        xp = xp + sig * noise
    return xp, x0


def ddim_sample(apply_model, ac, x_T, c, steps, eta=0.0, uc=None, scale=1.0, noises=None):
    """steps: ascending timestep list (``sorted(sampled_timestep)``); noises: optional list, one per step in loop order."""
    sig, a, a_prev = sampling_parameters(ac, steps, eta)
    x = x_T
    for i, step in enumerate(reversed(list(steps))):
        idx = len(steps) - i - 1
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        e = _guided(apply_model, x, t, c, uc, scale)
        x, _ = _update(x, e, a[idx], a_prev[idx], sig[idx], None if noises is None else noises[i])
    return x


def plms_sample(apply_model, ac, x_T, c, steps, uc=None, scale=1.0):
    sig, a, a_prev = sampling_parameters(ac, steps, 0.0)
    rng = list(reversed(list(steps)))
    x, old = x_T, []
    for i, step in enumerate(rng):
        idx = len(steps) - i - 1
        t = torch.full((x.shape[0],), int(step), dtype=torch.long)
        t_next = torch.full((x.shape[0],), int(rng[min(i + 1, len(rng) - 1)]), dtype=torch.long)
        e = _guided(apply_model, x, t, c, uc, scale)
        if len(old) == 0:
            xp, _ = _update(x, e, a[idx], a_prev[idx], sig[idx], None)
            ep = (e + _guided(apply_model, xp, t_next, c, uc, scale)) / 2
        elif len(old) == 1:
            ep = (3 * e - old[-1]) / 2
        elif len(old) == 2:
            ep = (23 * e - 16 * old[-1] + 5 * old[-2]) / 12
        else:
            ep = (55 * e - 59 * old[-1] + 37 * old[-2] - 9 * old[-3]) / 24
        x, _ = _update(x, ep, a[idx], a_prev[idx], sig[idx], None)
        old.append(e)
        if len(old) >= 4:
            old.pop(0)
    return x


def toy_model(x, t, c):
    """Deterministic stand-in for ``apply_model`` shared by the capture script and the tests: depends on x, t and c."""
    tt = t.to(torch.float32)[:, None, None, None]
    return (0.7 * x * torch.cos(tt * 0.003) + 0.3 * torch.roll(x, 1, -1) * torch.sin(tt * 0.002)
            + 0.1 * c.mean(dim=(1, 2))[:, None, None, None])


# ------------------------------------------------------------------ DPM-Solver++(2M), searched time points
# Restates dpm_solver/sampler.py:22-83 (DPMSolverSampler.sample: discrete VP schedule, classifier-free model_wrapper,
# predict_x0, multistep order 2, lower_order_final) and dpm_solver/dpm_solver.py: NoiseScheduleVP 'discrete' :99-108,
# 125-156, interpolate_fn :1144-1188, get_model_input_time :278-287, data_prediction_fn :380-400, first / second multistep
# updates :700-735 / 755-810, the ea_timesteps branch of sample() :1079-1118.
class DiscreteVP:
    def __init__(self, ac: torch.Tensor):
        self.log_alpha = 0.5 * torch.log(ac.to(torch.float64))
        self.N = len(ac)
        self.t = torch.linspace(0.0, 1.0, self.N + 1, dtype=torch.float64)[1:]

    def log_mean(self, t: float) -> float:
        k = int(torch.searchsorted(self.t, torch.tensor(t, dtype=torch.float64)))
        k = min(max(k, 1), self.N - 1)  # linear extrapolation from the boundary segments, like interpolate_fn
        x0, x1, y0, y1 = self.t[k - 1], self.t[k], self.log_alpha[k - 1], self.log_alpha[k]
        return float(y0 + (t - x0) * (y1 - y0) / (x1 - x0))

    def alpha(self, t):
        return float(np.exp(self.log_mean(t)))

    def std(self, t):
        return float(np.sqrt(1.0 - np.exp(2.0 * self.log_mean(t))))

    def lam(self, t):
        lm = self.log_mean(t)
        return lm - 0.5 * float(np.log(1.0 - np.exp(2.0 * lm)))


def dpm_time_points(ea_timesteps, n_total=1000):
    """sample()'s ea_timesteps branch: integer candidates index the ascending 1000-point uniform time grid, in the given
    order; float candidates are continuous times, sorted descending."""
    ea = list(ea_timesteps)
    if max(ea) > 1:
        full = torch.linspace(1.0, 1.0 / n_total, n_total + 1).tolist()
        full.reverse()
        return [float(np.float32(full[int(e)])) for e in ea]
    return [float(np.float32(t)) for t in sorted(ea, reverse=True)]


def dpm_sample(apply_model, ac, x_T, c, time_points, uc=None, scale=1.0):
    ns = DiscreteVP(ac)
    ts = list(time_points)
    steps = len(ts) - 1
    assert steps >= 2

    def data_pred(x, t):
        t_in = torch.full((x.shape[0],), (t - 1.0 / ns.N) * 1000.0, dtype=torch.float32)
        e = _guided(apply_model, x, t_in, c, uc, scale)
        return (x - ns.std(t) * e) / ns.alpha(t)

    def update(x, m_prev, m_cur, t_pp, s, t, order):
        h = ns.lam(t) - ns.lam(s)
        phi = ns.alpha(t) * (np.exp(-h) - 1.0)
        xt = (ns.std(t) / ns.std(s)) * x - phi * m_cur
        if order == 2:
            r0 = (ns.lam(s) - ns.lam(t_pp)) / h
            xt = xt - 0.5 * phi * (1.0 / r0) * (m_cur - m_prev)
        return xt

    x = x_T
    m_list = [data_pred(x, ts[0])]
    x = update(x, None, m_list[0], None, ts[0], ts[1], 1)
    m_list.append(data_pred(x, ts[1]))
    for step in range(2, steps + 1):
        order = min(2, steps + 1 - step) if steps < 15 else 2
        x = update(x, m_list[0], m_list[1], ts[step - 2], ts[step - 1], ts[step], order)
        m_list[0] = m_list[1]
        if step < steps:
            m_list[1] = data_pred(x, ts[step])
    return x
