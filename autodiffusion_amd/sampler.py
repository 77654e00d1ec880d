"""SpacedDiffusion: the reference's sampling interface over the fused HIP step kernel.

Drop-in mirror of the sampling half of ``GaussianDiffusion`` / ``SpacedDiffusion``
(reference guided_diffusion/gaussian_diffusion.py:441-534 p_sample_loop, :624-716
ddim_sample_loop incl. AutoDiffusion's ``return_all_images`` and start-noise yield;
respace.py:63-127 timestep mapping).  Per step, the whole of p_mean_variance +
condition_score / condition_mean + the x_{t-1} update is ONE launch of
``adm_ddim_step`` / ``adm_ddpm_step``; the model and cond_fn stay arbitrary
callables ``(x, t_mapped, **model_kwargs)`` exactly as in the reference.

``denoised_fn`` (applied to the predicted x_0 before the clip, :293-298) is honoured with two launches around the caller's function.
Training-time members (q_sample, training_losses, bpd, ddim_reverse_sample) are
out of scope: candidate evaluation never calls them.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops
from ._lib import StepCoefs
from .schedule import (LossType, ModelMeanType, ModelVarType, install_tables, subset_betas)


def step_coefs(tables, i: int, *, learned_range: bool, fixed: str = "large", predict_xstart=False,
               clip_denoised=True, eta=0.0) -> StepCoefs:
    """Pack step i's scalars (float64 -> float32, as _extract_into_tensor does).

    `tables` is any object/dict exposing the reference's table names."""
    g = (lambda n: tables[n]) if isinstance(tables, dict) else (lambda n: getattr(tables, n))
    c = StepCoefs()
    c.sqrt_recip_ac = float(g("sqrt_recip_alphas_cumprod")[i])
    c.sqrt_recipm1_ac = float(g("sqrt_recipm1_alphas_cumprod")[i])
    c.ac = float(g("alphas_cumprod")[i])
    c.ac_prev = float(g("alphas_cumprod_prev")[i])
    c.coef1 = float(g("posterior_mean_coef1")[i])
    c.coef2 = float(g("posterior_mean_coef2")[i])
    if learned_range:
        c.log_var_lo = float(g("posterior_log_variance_clipped")[i])
        c.log_var_hi = float(np.log(g("betas"))[i])
        c.fixed_var = 0.0
    elif fixed == "large":
        v = np.append(g("posterior_variance")[1], g("betas")[1:])
        c.fixed_var = float(v[i])
        c.log_var_lo = float(np.log(v)[i])
        c.log_var_hi = 0.0
    else:
        c.fixed_var = float(g("posterior_variance")[i])
        c.log_var_lo = float(g("posterior_log_variance_clipped")[i])
        c.log_var_hi = 0.0
    c.eta = float(eta)
    c.nonzero = int(i != 0)
    c.learned_range = int(learned_range)
    c.predict_xstart = int(predict_xstart)
    c.clip_denoised = int(clip_denoised)
    return c


_SIDE_STREAMS = {}  # device -> second HIP stream for the guidance gradient (kept out of the deep-copied objects)
_PARTITIONS = {}    # (device, spec) -> (UNet stream, guidance stream, description) | None


def cu_partition(device, spec: str):
    """Two CU-MASKED HIP streams for one guided step: the UNet on one part of the chip, the guidance gradient on the rest.

    Both networks' hot kernels are persistent one-block-per-CU kernels that fill a CU's LDS and registers, so on two ordinary
    streams they time-share every CU (the second stream only fills the first one's tails); on masked streams each runs undisturbed
    on its own CUs, and the memory-bound stretches of the gradient network leave their power budget to the other part's matrix
    cores.  spec = "G": G CUs (a multiple of 8: G / 8 of every XCD) go to the guidance stream, the rest to the UNet.  Results do
    not depend on the partition.  -> (unet_stream, guide_stream, description)."""
    key = (torch.device(device).index, spec)
    if key in _PARTITIONS:
        return _PARTITIONS[key]
    import ctypes as C
    from . import _lib
    g = int(spec)
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    if not (0 < g < ncu) or g % 8 or (ncu - g) % 8:
        raise ValueError(f"ADM_CU_PARTITION={spec!r}: expected a multiple of 8 with 0 < G < {ncu}")
    # mask bit i is CU i // 8 of XCD i % 8 (tools/cumask_probe.py on this pool: the driver deals the bits round-robin over the 8
    # XCDs, and a mask that leaves an XCD without a CU is ignored altogether): the top G bits = the last G / 8 CUs of EVERY XCD, so
    # both streams keep all 8 L2s and the kernels' `blockIdx & 7` XCD labels stay valid
    guide_bits = set(range(ncu - g, ncu))
    words = (ncu + 31) // 32

    def mk(bits):
        m = (C.c_uint32 * words)()
        for i in bits:
            m[i // 32] |= 1 << (i % 32)
        out = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.load().adm_stream_create_cumask(m, words, C.byref(out)), "adm_stream_create_cumask")
        return torch.cuda.ExternalStream(out.value, device=device)
    unet_s, guide_s = mk(set(range(ncu)) - guide_bits), mk(guide_bits)
    desc = f"CU-partitioned streams: UNet on {ncu - g} CUs, guidance gradient on {g} CUs ({g // 8} of every XCD's 32)"
    _PARTITIONS[key] = (unet_s, guide_s, desc)
    return _PARTITIONS[key]


class SpacedDiffusion:
    """A diffusion process over a subset of a base process's timesteps (sampling only)."""

    # eps(x_t) and the classifier-guidance gradient run concurrently on two HIP streams (bit-identical results;
    # ADM_OVERLAP_GUIDANCE=0 or setting the attribute to False restores the sequential order)
    overlap_guidance = os.environ.get("ADM_OVERLAP_GUIDANCE", "1") != "0"
    # "G": give the guidance gradient G CUs of its own and the UNet the rest (cu_partition); "" = two ordinary streams
    cu_partition_spec = os.environ.get("ADM_CU_PARTITION", "")

    def describe_overlap(self):
        if self.cu_partition_spec and self.overlap_guidance:
            return "timed region: " + cu_partition(torch.cuda.current_device(), self.cu_partition_spec)[2]
        return "timed region: UNet and classifier-guidance kernels overlap on two HIP streams"

    def __init__(self, use_timesteps, *, betas, model_mean_type, model_var_type, loss_type,
                 rescale_timesteps=False):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(betas)
        base_ac = np.cumprod(1.0 - np.array(betas, dtype=np.float64), axis=0)
        new_betas, self.timestep_map = subset_betas(base_ac, self.use_timesteps)
        install_tables(self, new_betas, allow_single_step=False)
        self.generator = None         # optional torch.Generator for the per-step noise draws
        self.last_uint8_nhwc = None   # uint8 NHWC pack of the last finished loop (fused into the final step)
        if model_mean_type == ModelMeanType.PREVIOUS_X:
            raise NotImplementedError("ModelMeanType.PREVIOUS_X is not produced by any reference factory")
        if model_var_type == ModelVarType.LEARNED:
            raise NotImplementedError("ModelVarType.LEARNED is not produced by any reference factory")

    # ------------------------------------------------------------------ helpers
    def _coefs(self, i, clip_denoised, eta=0.0):
        return step_coefs(
            self, i, learned_range=self.model_var_type == ModelVarType.LEARNED_RANGE,
            fixed="large" if self.model_var_type == ModelVarType.FIXED_LARGE else "small",
            predict_xstart=self.model_mean_type == ModelMeanType.START_X,
            clip_denoised=clip_denoised, eta=eta)

    def _mapped(self, t):
        """_WrappedModel: step index -> original timestep (float-rescaled if requested)."""
        map_tensor = torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)
        new_ts = map_tensor[t]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return new_ts

    def _step(self, kind, model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta=0.0,
              want_u8=False, index=None):
        if model_kwargs is None:
            model_kwargs = {}
        if index is None:  # direct p_sample / ddim_sample calls: read the index back from the tensor
            idx = t.tolist()
            if len(set(idx)) != 1:
                raise NotImplementedError("per-sample step indices: the sample loops always pass one index per batch")
            index = int(idx[0])
        i = index
        ts = self._mapped(t)
        x = x.contiguous()
        grad = None
        if cond_fn is not None and self.overlap_guidance and x.is_cuda and self.cu_partition_spec:
            # the two networks on disjoint sets of CUs (cu_partition): both depend on x_t only
            cur = torch.cuda.current_stream(x.device)
            unet_s, guide_s, _ = cu_partition(x.device, self.cu_partition_spec)
            unet_s.wait_stream(cur)
            guide_s.wait_stream(cur)
            with torch.cuda.stream(guide_s):
                grad = cond_fn(x, ts, **model_kwargs).float().contiguous()
            with torch.cuda.stream(unet_s):
                model_out = model(x, ts, **model_kwargs)
                if model_out.dtype != torch.float32:
                    model_out = model_out.float()
            cur.wait_stream(unet_s)
            cur.wait_stream(guide_s)
            grad.record_stream(cur)
            model_out.record_stream(cur)
            x.record_stream(unet_s)
            x.record_stream(guide_s)
        elif cond_fn is not None and self.overlap_guidance and x.is_cuda:
            # eps(x_t) and the guidance gradient both depend on x_t only: the gradient runs on a second HIP stream
            # and fills the dispatch gaps and tile-quantisation tails of the UNet's ~2000 launches (and vice versa)
            cur = torch.cuda.current_stream(x.device)
            side = _SIDE_STREAMS.get(x.device)
            if side is None:
                side = _SIDE_STREAMS[x.device] = torch.cuda.Stream(device=x.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                grad = cond_fn(x, ts, **model_kwargs).float().contiguous()
            model_out = model(x, ts, **model_kwargs)
            cur.wait_stream(side)
            grad.record_stream(cur)
        else:
            model_out = model(x, ts, **model_kwargs)
            if cond_fn is not None:
                grad = cond_fn(x, ts, **model_kwargs).float().contiguous()
        if model_out.dtype != torch.float32:
            model_out = model_out.float()
        # drawn every step, as the reference does (keeps the RNG stream aligned with it)
        if isinstance(self.generator, (list, tuple)):
            # several reference batches evaluated in one pass (CandidateEvaluator.sample_batches): [(generator, images), ...] --
            # every sub-batch draws from its own generator, so its images are those of its own separate evaluation
            noise = torch.cat([torch.randn((cnt,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype, generator=g_)
                               for g_, cnt in self.generator], 0)
        elif self.generator is not None:
            noise = torch.randn(x.shape, device=x.device, dtype=x.dtype, generator=self.generator)
        else:
            noise = torch.randn_like(x)
        coefs = self._coefs(i, clip_denoised, eta)
        model_out = model_out.contiguous()
        if denoised_fn is not None:
            # process_xstart (gaussian_diffusion.py:293-298): the caller's function acts on the predicted x_0 BEFORE the clip, in the
            # middle of what is otherwise one fused launch.  Two launches around it (no reference script passes one): the first,
            # unclipped and unguided, only yields x_0 = sqrt(1/abar) x - sqrt(1/abar - 1) eps; the second takes denoised_fn(x_0) as
            # a START_X model output -- the kernel's predict_xstart path is exactly `process_xstart(model_output)` followed by the
            # same posterior / guidance / update arithmetic (:299-306, :381-393)
            c = x.shape[1]
            if coefs.predict_xstart:
                x0_raw = model_out[:, :c]
            else:
                probe = self._coefs(i, False, 0.0)   # unclipped, eta 0, no guidance, no noise: the ddim kernel's pred_xstart output IS that x_0
                _, x0_raw, _ = ops.sampler_step("ddim", x, model_out, probe, None, None, want_xstart=True)
            x0_fn = denoised_fn(x0_raw).to(torch.float32)
            model_out = torch.cat([x0_fn, model_out[:, c:]], 1).contiguous() if coefs.learned_range else x0_fn.contiguous()
            coefs.predict_xstart = 1
        sample, x0, u8 = ops.sampler_step(kind, x, model_out, coefs, grad, noise, want_xstart=True, want_u8=want_u8)
        out = {"sample": sample, "pred_xstart": x0}
        if want_u8:
            out["uint8_nhwc"] = u8
        return out

    # ------------------------------------------------------------------ reference API
    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None):
        return self._step("ddpm", model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                    eta=0.0):
        return self._step("ddim", model, x, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta)

    def _loop(self, kind, model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs, device,
              progress, eta, yield_start):
        if device is None:
            device = next(model.parameters()).device
        assert isinstance(shape, (tuple, list))
        img = noise if noise is not None else torch.randn(*shape, device=device)
        indices = list(range(self.num_timesteps))[::-1]
        if progress:
            from tqdm.auto import tqdm
            indices = tqdm(indices)
        if yield_start:
            yield {"sample": img}
        for i in indices:
            t = torch.tensor([i] * shape[0], device=device)
            with torch.no_grad():
                out = self._step(kind, model, img, t, clip_denoised, denoised_fn, cond_fn, model_kwargs, eta,
                                 want_u8=(i == 0), index=i)
            yield out
            img = out["sample"]

    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                  cond_fn=None, model_kwargs=None, device=None, progress=False):
        yield from self._loop("ddpm", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                              device, progress, 0.0, yield_start=False)

    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False):
        final = None
        for sample in self.p_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                                                     model_kwargs, device, progress):
            final = sample
        self.last_uint8_nhwc = final.get("uint8_nhwc")
        return final["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0):
        yield from self._loop("ddim", model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                              device, progress, eta, yield_start=True)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, return_all_images=False):
        final = None
        all_images = []
        for sample in self.ddim_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn,
                                                        model_kwargs, device, progress, eta):
            final = sample
            if return_all_images:
                all_images.append(final["sample"])
        self.last_uint8_nhwc = final.get("uint8_nhwc")
        if return_all_images:
            return all_images
        return final["sample"]


__all__ = ["SpacedDiffusion", "step_coefs", "ModelMeanType", "ModelVarType", "LossType"]
