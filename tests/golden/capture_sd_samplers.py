#!/usr/bin/env python3
"""Capture golden latents from the REFERENCE's own DDIMSampler / PLMSSampler ("Stable Diffusion"/ldm/models/diffusion/
ddim.py, plms.py) driven with searched timestep lists (``sampled_timestep``), classifier-free guidance and a toy
``apply_model`` (oracle/sd_sampler.py::toy_model).  Build container only (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_sd_samplers.py

The reference's ``register_buffer`` moves every table to "cuda" (ddim.py:17-21); there is no GPU in this container,
so the capture subclasses override it with a plain ``setattr`` (device placement only, no arithmetic); likewise the
one ``torch.Tensor(timesteps).to('cuda')`` in dpm_solver.py:1088/1091 is routed to the CPU for the DPM-Solver runs.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/examples/Stable Diffusion")
sys.path.insert(0, ROOT)

from ldm.models.diffusion.ddim import DDIMSampler  # noqa: E402
from ldm.models.diffusion.plms import PLMSSampler  # noqa: E402
from ldm.models.diffusion.dpm_solver import dpm_solver as _dpm  # noqa: E402
from ldm.models.diffusion.dpm_solver.sampler import DPMSolverSampler  # noqa: E402
from ldm.modules.diffusionmodules.util import make_beta_schedule  # noqa: E402
from oracle.sd_sampler import toy_model  # noqa: E402


class Model:  # the attributes of LatentDiffusion the samplers read (ddpm.py:117-137)
    def __init__(self):
        betas = make_beta_schedule("linear", 1000, linear_start=0.00085, linear_end=0.0120)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.num_timesteps = 1000
        self.device = torch.device("cpu")
        self.betas = torch.tensor(betas, dtype=torch.float32)
        self.alphas_cumprod = torch.tensor(ac, dtype=torch.float32)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]), dtype=torch.float32)

    def apply_model(self, x, t, c):
        return toy_model(x, t, c)


class CpuDDIM(DDIMSampler):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


class CpuPLMS(PLMSSampler):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


class CpuDPM(DPMSolverSampler):
    def register_buffer(self, name, attr):
        setattr(self, name, attr)


# dpm_solver.py:1088/1091 builds its time points with ``torch.Tensor(list).to('cuda')``; no GPU here, so the capture
# routes that one call to the CPU (device placement only)
def _tensor_on_cpu(data):
    t = torch.tensor(data, dtype=torch.float32)

    class _T:
        def to(self, *_a, **_k):
            return t
    return _T()


if __name__ == "__main__":
    g = torch.Generator().manual_seed(7)
    b, shape = 3, (4, 8, 8)
    x_T = torch.randn(b, *shape, generator=g)
    c = torch.randn(b, 5, 16, generator=g)
    uc = torch.randn(b, 5, 16, generator=g)
    m = Model()
    out = dict(x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(), alphas_cumprod=m.alphas_cumprod.numpy(), betas=m.betas.numpy())
    cands = {"k4": [153, 424, 926, 690], "k6": [94, 834, 217, 944, 574, 354], "k1": [500]}
    for tag, cand in cands.items():
        out[f"cand_{tag}"] = np.array(cand)
        for name, cls in (("ddim", CpuDDIM), ("plms", CpuPLMS)):
            for gtag, (scale, ucond) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
                st = np.array(sorted(cand)) if name == "plms" else np.array(cand)  # search_ea.py sorts before PLMS
                s, _ = cls(m).sample(S=len(cand), batch_size=b, shape=list(shape), conditioning=c, verbose=False, eta=0.0,
                                     x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=ucond,
                                     sampled_timestep=st)
                out[f"{name}_{tag}_{gtag}"] = s.numpy()
    # DPM-Solver++(2M): integer candidates (indices into the 1000-point time grid) and continuous-time candidates
    class _TorchShim:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def Tensor(data):
            return _tensor_on_cpu(data)
    _dpm.torch = _TorchShim()
    dpm_cands = {"i4": [999, 750, 500, 250, 0], "i6": [980, 901, 640, 433, 210, 77, 3],
                 "f4": [1.0, 0.7502, 0.5005, 0.2508, 0.001], "i2": [900, 450, 10]}
    for tag, cand in dpm_cands.items():
        out[f"dpmcand_{tag}"] = np.array(cand, dtype=np.float64)
        for gtag, (scale, ucond) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
            s, _ = CpuDPM(m).sample(S=len(cand) - 1, batch_size=b, shape=list(shape), conditioning=c, verbose=False,
                                    x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=ucond,
                                    sampled_timestep=cand)
            out[f"dpm_{tag}_{gtag}"] = s.numpy()
    _dpm.torch = torch
    # the uniform schedule the samplers fall back to without a searched list
    for S in (4, 10, 50):
        d = CpuDDIM(m)
        d.make_schedule(ddim_num_steps=S, ddim_eta=0.0, verbose=False)
        out[f"uniform_{S}"] = np.asarray(d.ddim_timesteps)
    s, _ = CpuDDIM(m).sample(S=4, batch_size=b, shape=list(shape), conditioning=c, verbose=False, eta=0.0, x_T=x_T)
    out["ddim_uniform4_plain"] = s.numpy()
    path = os.path.join(HERE, "sd_samplers.npz")
    np.savez_compressed(path, **out)
    print(sorted(out), f"{os.path.getsize(path) / 1024:.1f} KiB")
