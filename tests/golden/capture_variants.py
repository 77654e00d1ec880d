#!/usr/bin/env python3
"""Golden vectors for the reference UNet variants no launch script uses but the factories offer (round 3): ``use_scale_shift_norm=False``
(unet.py:251-254: ``out_layers(h + emb_out)``) and ``resblock_updown=False`` (unet.py:78-141: conv ``Downsample`` / ``Upsample``), captured by
importing the REFERENCE's own modules, like capture_golden.py (same fill rule, same stand-ins: none needed here).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_variants.py     ->  tests/golden/unet_m32_variants.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/examples/guided_diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle.fill import fill_array  # noqa: E402
from guided_diffusion.script_util import create_model_and_diffusion, model_and_diffusion_defaults  # noqa: E402

torch.set_num_threads(8)


def cfg(scale_shift, updown, dynamic):
    d = model_and_diffusion_defaults()
    d.update(image_size=32, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="16,8",
             num_head_channels=32, class_cond=True, learn_sigma=True, resblock_updown=updown,
             use_scale_shift_norm=scale_shift, use_new_attention_order=True, use_dynamic_unet=dynamic, noise_schedule="cosine")
    return d


def main():
    g = torch.Generator().manual_seed(41)
    x = torch.randn(2, 3, 32, 32, generator=g)
    t, y = torch.tensor([37, 901]), torch.tensor([3, 977])
    out = {"x": x.numpy(), "t": t.numpy(), "y": y.numpy()}
    for tag, ss, ud, dyn in (("noss", False, True, True), ("convres", True, False, False), ("defaults", False, False, True)):
        m, _ = create_model_and_diffusion(**cfg(ss, ud, dyn))
        with torch.no_grad():
            for k, v in m.state_dict().items():
                v.copy_(torch.from_numpy(fill_array(k, tuple(v.shape))))
        m.eval()
        with torch.no_grad():
            out[f"out_{tag}"] = m(x, t, y).numpy()
            if dyn:
                sk = [1, 4, m.layer_num - 2]
                out[f"skip_{tag}"] = np.array(sk, dtype=np.int64)
                out[f"out_{tag}_skip"] = m(x, t, y, skip_layer=sk).numpy()
                out[f"layer_num_{tag}"] = np.array(m.layer_num)
        out[f"nparams_{tag}"] = np.array(sum(p.numel() for p in m.parameters()))
    path = os.path.join(HERE, "unet_m32_variants.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB", sorted(out))


if __name__ == "__main__":
    main()
