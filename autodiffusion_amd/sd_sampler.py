"""Latent-diffusion samplers with a searched timestep list on the HIP path.

Host-side mirror of the reference's ``DDIMSampler`` / ``PLMSSampler`` ("Stable Diffusion"/ldm/models/diffusion/ddim.py:60-203,
plms.py:63-258) as AutoDiffusion drives them (scripts/search_ea.py:528-538): ``sampler.sample(S, batch_size, shape,
conditioning, ..., unconditional_guidance_scale, unconditional_conditioning, eta, x_T, sampled_timestep=cand)`` returning
``(samples, intermediates)``, attribute ``ddpm_num_timesteps``.  ``model`` is anything with the attributes the reference
samplers read from ``LatentDiffusion`` -- ``num_timesteps, betas, alphas_cumprod, alphas_cumprod_prev, device,
apply_model(x, t, c)`` -- e.g. ``LatentDiffusion`` below around the HIP latent UNet (``sd_unet.UNetModel``).

Every update (classifier-free-guidance combine, PLMS multistep blend, pred_x0, x_prev) is ONE ``adm_sd_step`` launch; the
per-step scalars are computed on the host in float32 exactly as the reference's float32 tables hold them.
"""
from __future__ import annotations

import ctypes as C
import itertools

import os

import numpy as np
import torch

from . import _lib
from ._lib import AdmError, SdStepCoefs, check


_CALL_IDS = itertools.count()  # one id per sample() call in this process: the context key handed to the model


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """The four named beta schedules of ldm/modules/diffusionmodules/util.py:21-43, float64 numpy."""
    grid = lambda lo, hi: np.linspace(lo, hi, n_timestep, dtype=np.float64)  # noqa: E731
    if schedule == "linear":        # linear in sqrt(beta): the latent-diffusion default
        return grid(linear_start ** 0.5, linear_end ** 0.5) ** 2
    if schedule == "sqrt_linear":
        return grid(linear_start, linear_end)
    if schedule == "sqrt":
        return grid(linear_start, linear_end) ** 0.5
    if schedule == "cosine":
        t = np.arange(n_timestep + 1, dtype=np.float64) / n_timestep + cosine_s
        abar = np.cos(t / (1 + cosine_s) * np.pi / 2) ** 2
        abar /= abar[0]
        return np.clip(1 - abar[1:] / abar[:-1], 0, 0.999)
    raise ValueError(f"schedule '{schedule}' unknown.")


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """util.py:46-63: the fallback grid when no searched list is given.  The reference ROUNDS the stride (it does not
    floor it) and shifts every step by one."""
    if ddim_discr_method == "uniform":
        stride = round(num_ddpm_timesteps / num_ddim_timesteps)
        steps = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        steps = (np.linspace(0, np.sqrt(num_ddpm_timesteps * .8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    return steps + 1


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """util.py:66-78 on float32 values: (sigmas, alphas, alphas_prev), numpy float32."""
    ac = np.asarray(alphacums, dtype=np.float32)
    steps = [int(t) for t in ddim_timesteps]
    alphas = ac[steps]
    alphas_prev = np.concatenate([ac[:1], ac[steps[:-1]]]).astype(np.float32)
    one = np.float32(1.0)
    sigmas = np.float32(eta) * np.sqrt((one - alphas_prev) / (one - alphas) * (one - alphas / alphas_prev))
    return sigmas.astype(np.float32), alphas, alphas_prev


class LatentDiffusion:
    """The slice of ``ldm.models.diffusion.ddpm.LatentDiffusion`` the samplers use (register_schedule ddpm.py:117-137,
    apply_model with ``conditioning_key: crossattn``): float32 schedule tables and the UNet call."""

    parameterization = "eps"

    def __init__(self, unet, timesteps=1000, beta_schedule="linear", linear_start=0.00085, linear_end=0.0120,
                 cosine_s=8e-3, device=None):
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end,
                                   cosine_s=cosine_s)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = int(timesteps)
        self.device = torch.device(device) if device is not None else getattr(unet, "device", torch.device("cpu"))
        self.betas = torch.tensor(betas, dtype=torch.float32, device=self.device)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32, device=self.device)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32, device=self.device)
        self.model = unet

    def apply_model(self, x_noisy, t, cond, context_key=None):
        if context_key is not None and getattr(self.model, "accepts_context_key", False):
            return self.model(x_noisy, t, context=cond, context_key=context_key)
        return self.model(x_noisy, t, context=cond)

    accepts_context_key = True


def _f32ptr(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise AdmError(f"{name}: expected a contiguous float32 device tensor (the HIP path has no CPU fallback)")
    return t.data_ptr()


def sd_step(x, eps, batch, cfg_scale, weights, hist, a_t, a_prev, sigma, noise=None, want_e=True):
    """One fused update.  eps: model output, [2*batch, ...] (unconditional half first) under guidance, else [batch, ...].
    weights/hist: multistep blend e' = w[0]*e + sum w[k]*hist[k-1].  a_t, a_prev, sigma: float32 scalars.
    Returns (x_prev, pred_x0, e)."""
    guided = eps.shape[0] == 2 * batch
    if not guided and eps.shape[0] != batch:
        raise AdmError(f"sd_step: model output batch {eps.shape[0]} is neither {batch} nor {2 * batch}")
    eps = eps.contiguous()
    eu, ec = (eps[:batch], eps[batch:]) if guided else (None, eps)
    x_prev, pred = torch.empty_like(x), torch.empty_like(x)
    e_out = torch.empty_like(x) if want_e else None
    one = np.float32(1.0)
    a_t, a_prev, sigma = np.float32(a_t), np.float32(a_prev), np.float32(sigma)
    co = SdStepCoefs()
    co.cfg_scale = float(cfg_scale)
    w = list(weights) + [0.0] * (4 - len(weights))
    co.w = (C.c_float * 4)(*[float(v) for v in w])
    co.sqrt_one_minus_at = float(np.sqrt(one - a_t))
    co.sqrt_at = float(np.sqrt(a_t))
    co.sqrt_a_prev = float(np.sqrt(a_prev))
    co.dir_coef = float(np.sqrt(one - a_prev - sigma * sigma))
    co.sigma = float(sigma)
    h = list(hist) + [None] * (3 - len(hist))
    check(_lib.load().adm_sd_step(_f32ptr(x, "x"), _f32ptr(eu, "eps"), _f32ptr(ec, "eps"), _f32ptr(h[0], "old_eps"),
                                  _f32ptr(h[1], "old_eps"), _f32ptr(h[2], "old_eps"), _f32ptr(noise, "noise"),
                                  x_prev.data_ptr(), pred.data_ptr(), None if e_out is None else e_out.data_ptr(),
                                  x.numel(), C.byref(co), torch.cuda.current_stream().cuda_stream), "adm_sd_step")
    return x_prev, pred, e_out


_SIDE_STREAMS = {}  # device -> second HIP stream for the unconditional half of a guided evaluation


class _LatentSampler:
    def __init__(self, model, schedule="linear", **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True, sampled_timestep=None):
        if sampled_timestep is None:
            self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps, verbose)
        else:
            self.ddim_timesteps = sampled_timestep
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.alphas_cumprod_host = ac.detach().to(torch.float32).cpu().numpy()
        self.ddim_sigmas, self.ddim_alphas, self.ddim_alphas_prev = make_ddim_sampling_parameters(
            self.alphas_cumprod_host, self.ddim_timesteps, ddim_eta, verbose)
        self.ddim_sqrt_one_minus_alphas = np.sqrt(np.float32(1.0) - self.ddim_alphas)

    def _unsupported(self, **kw):
        bad = [k for k, v in kw.items() if v]
        if bad:
            raise NotImplementedError(f"{type(self).__name__}: {bad} are not built on the HIP path (unused by search_ea.py)")

    # Classifier-free guidance evaluates the UNet on [uncond | cond] (ddim.py:177-181 concatenates them into one batch of
    # 2N).  At the search's n_samples (6 latents) most launches of that batch under-fill 256 CUs, so the two halves run as
    # two evaluations of N on two HIP streams instead (each stream replays its own captured graph: sd_unet.py keys graphs
    # and conditioning projections by the launching stream) and fill each other's tails.  Bit-identical: every kernel's
    # arithmetic per latent is independent of the batch it rides in.  ADM_SD_SPLIT_GUIDANCE=0 restores the single batch.
    split_guidance = os.environ.get("ADM_SD_SPLIT_GUIDANCE", "1") != "0"

    def _begin(self, c, uc, scale):
        """Per sample() call: the guidance batch [uncond | cond] of the conditioning is built once, and models that
        take a ``context_key`` (LatentDiffusion over the HIP UNet) are told that it stays fixed for the call's steps."""
        self._guided = not (uc is None or scale == 1.)
        self._split = bool(self._guided and self.split_guidance and c.is_cuda and getattr(self.model, "accepts_context_key", False))
        self._c, self._uc = c, uc
        self._c_in = torch.cat([uc, c]) if (self._guided and not self._split) else c
        self._key = ("sample", next(_CALL_IDS)) if getattr(self.model, "accepts_context_key", False) else None

    def _eps(self, x, t):
        if self._guided and self._split:
            cur = torch.cuda.current_stream(x.device)
            side = _SIDE_STREAMS.get(x.device)
            if side is None:
                side = _SIDE_STREAMS[x.device] = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                eu = self.model.apply_model(x, t, self._uc, context_key=(self._key, "u"))
            ec = self.model.apply_model(x, t, self._c, context_key=(self._key, "c"))
            cur.wait_stream(side)
            eu.record_stream(cur)
            return torch.cat([eu, ec])
        if self._guided:
            x, t = torch.cat([x] * 2), torch.cat([t] * 2)
        if self._key is not None:
            return self.model.apply_model(x, t, self._c_in, context_key=self._key)
        return self.model.apply_model(x, t, self._c_in)

    def _start(self, shape, x_T):
        device = self.model.betas.device
        if device.type != "cuda":
            raise AdmError("latent samplers: the model's tables are on the CPU (the HIP path has no CPU fallback)")
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device=device, dtype=torch.float32)
        return device, img.contiguous()

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, sampled_timestep=None, **kwargs):
        self._unsupported(quantize_x0=quantize_x0, mask=mask is not None, noise_dropout=noise_dropout > 0.,
                          score_corrector=score_corrector is not None)
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        sampled_timestep = self._order(sampled_timestep)
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose, sampled_timestep=sampled_timestep)
        C_, H, W = shape
        return self._loop(conditioning, (batch_size, C_, H, W), x_T=x_T, callback=callback, img_callback=img_callback,
                          log_every_t=log_every_t, temperature=temperature,
                          unconditional_guidance_scale=unconditional_guidance_scale,
                          unconditional_conditioning=unconditional_conditioning)


class DDIMSampler(_LatentSampler):
    def _order(self, sampled_timestep):
        return None if sampled_timestep is None else sorted(int(t) for t in sampled_timestep)  # ddim.py:93-94

    def _loop(self, cond, shape, x_T, callback, img_callback, log_every_t, temperature, unconditional_guidance_scale,
              unconditional_conditioning):
        device, img = self._start(shape, x_T)
        self._begin(cond, unconditional_conditioning, unconditional_guidance_scale)
        b = shape[0]
        timesteps = np.asarray(self.ddim_timesteps)
        total = timesteps.shape[0]
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        for i, step in enumerate(np.flip(timesteps)):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            eps = self._eps(img, ts)
            sigma = self.ddim_sigmas[index]
            noise = torch.randn(shape, device=device) * temperature if sigma != 0 else None  # ddim.py:198 noise_like
            img, pred_x0, _ = sd_step(img, eps, b, unconditional_guidance_scale, (1.0,), (), self.ddim_alphas[index],
                                      self.ddim_alphas_prev[index], sigma, noise, want_e=False)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates


class PLMSSampler(_LatentSampler):
    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True, sampled_timestep=None):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        super().make_schedule(ddim_num_steps, ddim_discretize, ddim_eta, verbose, sampled_timestep)

    def _order(self, sampled_timestep):
        return sampled_timestep  # plms.py takes the list as given; search_ea.py sorts its candidates (:480)

    def _loop(self, cond, shape, x_T, callback, img_callback, log_every_t, temperature, unconditional_guidance_scale,
              unconditional_conditioning):
        device, img = self._start(shape, x_T)
        b = shape[0]
        timesteps = np.asarray(self.ddim_timesteps)
        total = timesteps.shape[0]
        time_range = np.flip(timesteps)
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        uc, scale = unconditional_conditioning, unconditional_guidance_scale
        self._begin(cond, uc, scale)
        old_eps = []
        for i, step in enumerate(time_range):
            index = total - i - 1
            ts = torch.full((b,), int(step), device=device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), device=device, dtype=torch.long)
            a_t, a_prev, sigma = self.ddim_alphas[index], self.ddim_alphas_prev[index], self.ddim_sigmas[index]
            eps = self._eps(img, ts)
            if len(old_eps) == 0:    # pseudo improved Euler (2nd order): a second model call at the predicted point
                x_mid, _, e_t = sd_step(img, eps, b, scale, (1.0,), (), a_t, a_prev, sigma)
                eps2 = self._eps(x_mid, ts_next)
                img, pred_x0, _ = sd_step(img, eps2, b, scale, (0.5, 0.5), (e_t,), a_t, a_prev, sigma, want_e=False)
            else:                    # Adams-Bashforth of order 2 / 3 / 4 over the kept eps history (newest first)
                w = {1: (3 / 2, -1 / 2), 2: (23 / 12, -16 / 12, 5 / 12), 3: (55 / 24, -59 / 24, 37 / 24, -9 / 24)}[len(old_eps)]
                img, pred_x0, e_t = sd_step(img, eps, b, scale, w, tuple(reversed(old_eps)), a_t, a_prev, sigma)
            old_eps.append(e_t)
            if len(old_eps) >= 4:
                old_eps.pop(0)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img, intermediates


# ------------------------------------------------------------------ DPM-Solver++(2M)
class NoiseScheduleVP:
    """The 'discrete' VP schedule of dpm_solver.py:99-156: log(alpha_t) is the piecewise-linear interpolant of
    0.5*log(alphas_cumprod) over t = 1/N .. 1 (linear extrapolation outside, as interpolate_fn :1144-1188)."""

    def __init__(self, schedule="discrete", betas=None, alphas_cumprod=None):
        if schedule != "discrete":
            raise NotImplementedError("only the discrete schedule is used by DPMSolverSampler (sampler.py:60)")
        ac = np.asarray(torch.as_tensor(alphas_cumprod).detach().cpu(), dtype=np.float64) if alphas_cumprod is not None \
            else np.cumprod(1.0 - np.asarray(torch.as_tensor(betas).detach().cpu(), dtype=np.float64))
        self.schedule, self.total_N, self.T = schedule, len(ac), 1.0
        self.log_alpha_array = 0.5 * np.log(ac)
        self.t_array = np.linspace(0.0, 1.0, self.total_N + 1)[1:]

    def marginal_log_mean_coeff(self, t: float) -> float:
        k = min(max(int(np.searchsorted(self.t_array, t)), 1), self.total_N - 1)
        x0, x1 = self.t_array[k - 1], self.t_array[k]
        y0, y1 = self.log_alpha_array[k - 1], self.log_alpha_array[k]
        return float(y0 + (t - x0) * (y1 - y0) / (x1 - x0))

    def marginal_alpha(self, t):
        return float(np.exp(self.marginal_log_mean_coeff(t)))

    def marginal_std(self, t):
        return float(np.sqrt(1.0 - np.exp(2.0 * self.marginal_log_mean_coeff(t))))

    def marginal_lambda(self, t):
        lm = self.marginal_log_mean_coeff(t)
        return lm - 0.5 * float(np.log(1.0 - np.exp(2.0 * lm)))


def dpm_step(x, eps, batch, cfg_scale, m_prev, sigma_s, alpha_s, a, b0, b1):
    """One fused multistep update; returns (x_next, m) with m the data prediction at the current point."""
    guided = eps.shape[0] == 2 * batch
    if not guided and eps.shape[0] != batch:
        raise AdmError(f"dpm_step: model output batch {eps.shape[0]} is neither {batch} nor {2 * batch}")
    eps = eps.contiguous()
    eu, ec = (eps[:batch], eps[batch:]) if guided else (None, eps)
    x_next, m = torch.empty_like(x), torch.empty_like(x)
    check(_lib.load().adm_dpm_step(_f32ptr(x, "x"), _f32ptr(eu, "eps"), _f32ptr(ec, "eps"), _f32ptr(m_prev, "model_prev"),
                                   x_next.data_ptr(), m.data_ptr(), x.numel(), float(cfg_scale), float(sigma_s),
                                   float(alpha_s), float(a), float(b0), float(b1),
                                   torch.cuda.current_stream().cuda_stream), "adm_dpm_step")
    return x_next, m


class DPMSolverSampler(_LatentSampler):
    """dpm_solver/sampler.py:5-83: multistep DPM-Solver++ of order 2 in data-prediction form, ``lower_order_final``,
    classifier-free guidance, over K+1 searched time points (``sampled_timestep``: integers index the ascending
    1000-point uniform time grid in the given order; floats are continuous times, sorted descending --
    dpm_solver.py:1079-1091) or the uniform grid when none are given."""

    def __init__(self, model, **kwargs):
        super().__init__(model, **kwargs)
        self.alphas_cumprod = model.alphas_cumprod.detach().to(torch.float32)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, sampled_timestep=None, **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f"Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}")
        C_, H, W = shape
        device, x = self._start((batch_size, C_, H, W), x_T)
        ns = NoiseScheduleVP('discrete', alphas_cumprod=self.alphas_cumprod)
        steps, order = int(S), 2
        assert steps >= order
        if sampled_timestep is None:
            ts = [float(v) for v in np.linspace(ns.T, 1.0 / ns.total_N, steps + 1, dtype=np.float32)]
        else:
            ea = [v.item() if hasattr(v, "item") else v for v in sampled_timestep]
            if max(ea) > 1:
                full = np.linspace(ns.T, 1.0 / ns.total_N, 1000 + 1, dtype=np.float32)[::-1]
                ts = [float(full[int(e)]) for e in ea]
            else:
                ts = [float(np.float32(t)) for t in sorted(ea, reverse=True)]
        assert len(ts) - 1 == steps
        uc, scale = unconditional_conditioning, unconditional_guidance_scale

        self._begin(conditioning, uc, scale)

        def eps_at(x_, t):  # model_wrapper: discrete-time input (t - 1/N) * 1000, guidance batch = [uncond | cond]
            t_in = torch.full((batch_size,), (t - 1.0 / ns.total_N) * 1000.0, device=device, dtype=torch.float32)
            return self._eps(x_, t_in)

        m_prev, lam_prev = None, None
        for step in range(1, steps + 1):
            s, t = ts[step - 1], ts[step]
            if step == 1:
                step_order = 1
            else:
                step_order = min(order, steps + 1 - step) if steps < 15 else order  # lower_order_final
            lam_s, lam_t = ns.marginal_lambda(s), ns.marginal_lambda(t)
            h = lam_t - lam_s
            phi = ns.marginal_alpha(t) * (np.exp(-h) - 1.0)
            a = ns.marginal_std(t) / ns.marginal_std(s)
            if step_order == 2:
                r0 = (lam_s - lam_prev) / h
                b0, b1 = -phi * (1.0 + 0.5 / r0), 0.5 * phi / r0
            else:
                b0, b1 = -phi, 0.0
            x, m = dpm_step(x, eps_at(x, s), batch_size, scale, m_prev if step_order == 2 else None,
                            ns.marginal_std(s), ns.marginal_alpha(s), a, b0, b1)
            m_prev, lam_prev = m, lam_s
        return x, None
