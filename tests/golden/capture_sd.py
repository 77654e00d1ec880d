#!/usr/bin/env python3
"""Capture golden vectors of the Stable-Diffusion latent UNet by importing the REFERENCE's own modules
("Stable Diffusion"/ldm/modules/diffusionmodules/openaimodel.py, ldm/modules/attention.py).

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.  Only inputs and
expected outputs are stored -- weights are regenerated on both sides from ``oracle/fill.py``.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_sd.py

``omegaconf`` is absent from this image; openaimodel.py imports ``omegaconf.listconfig.ListConfig`` for one
``type(context_dim) == ListConfig`` check (openaimodel.py:476-478).  A stand-in class is registered *in this capture
process only* (an int ``context_dim`` never matches it).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/examples/Stable Diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

_oc, _lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")
_lc.ListConfig = type("ListConfig", (list,), {})
_oc.listconfig = _lc
sys.modules.setdefault("omegaconf", _oc)
sys.modules.setdefault("omegaconf.listconfig", _lc)

from oracle.fill import fill_array  # noqa: E402
from ldm.modules.diffusionmodules.openaimodel import UNetModel  # noqa: E402

torch.set_num_threads(8)

CONFIGS = {
    # two levels, heads of 32 and 64 channels, a Downsample and an Upsample
    "sd_unet_tiny": dict(cfg=dict(in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[1, 2],
                                  num_res_blocks=1, channel_mult=[1, 2], num_heads=2, transformer_depth=1,
                                  context_dim=96, legacy=False),
                         n=2, hw=16, s=7, t=[10, 500]),
    # one level of the real v1 width: 320 channels, 8 heads of 40 channels, 77 x 768 context
    "sd_unet_w320": dict(cfg=dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[1],
                                  num_res_blocks=1, channel_mult=[1], num_heads=8, transformer_depth=1,
                                  context_dim=768, legacy=False),
                         n=1, hw=16, s=77, t=[981]),
}


def rnd(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


if __name__ == "__main__":
    for name, c in CONFIGS.items():
        net = UNetModel(image_size=32, use_spatial_transformer=True, use_checkpoint=False, **c["cfg"]).eval()
        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(torch.from_numpy(fill_array(k, tuple(v.shape))))
        x = rnd((c["n"], 4, c["hw"], c["hw"]), 1)
        ctx = rnd((c["n"], c["s"], c["cfg"]["context_dim"]), 2)
        t = torch.tensor(c["t"], dtype=torch.int64)
        with torch.no_grad():
            out = net(x, t, ctx)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, x=x.numpy(), t=t.numpy(), context=ctx.numpy(), out=out.numpy(),
                            cfg=np.array(repr(c["cfg"])))
        print(name, "params", sum(p.numel() for p in net.parameters()), "out rms", float(out.pow(2).mean().sqrt()),
              f"{os.path.getsize(path) / 1024:.1f} KiB")
