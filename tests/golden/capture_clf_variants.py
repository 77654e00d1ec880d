#!/usr/bin/env python3
"""Golden vectors for the reference CLASSIFIER variants no launch script uses but ``create_classifier`` offers: ``classifier_pool`` =
"adaptive" | "spatial" | "spatial_v2" (unet.py:826-856, 880-896), ``classifier_use_scale_shift_norm=False`` (unet.py:251-254) and
``classifier_resblock_updown=False`` (conv ``Downsample``, unet.py:115-140) -- logits and the guidance gradient of the search script's
``cond_fn`` (search_imagenet64_classifier_guidance.py:319-326: autograd through log_softmax), captured by importing the REFERENCE's own
modules like capture_variants.py (same fill rule).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_clf_variants.py     ->  tests/golden/clf_variants.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/examples/guided_diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from oracle.fill import fill_array  # noqa: E402
from guided_diffusion.script_util import create_classifier  # noqa: E402

torch.set_num_threads(8)

# tag -> (pool, use_scale_shift_norm, resblock_updown)
CASES = {"adaptive_noss_convres": ("adaptive", False, False), "spatial": ("spatial", True, True),
         "spatialv2_noss": ("spatial_v2", False, True), "attention_convres": ("attention", True, False)}


def main():
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 3, 64, 64, generator=g)
    t, y = torch.tensor([41, 873]), torch.tensor([7, 912])
    out = {"x": x.numpy(), "t": t.numpy(), "y": y.numpy()}
    for tag, (pool, ss, ud) in CASES.items():
        m = create_classifier(image_size=64, classifier_use_fp16=False, classifier_width=64, classifier_depth=1,
                              classifier_attention_resolutions="32,16,8", classifier_use_scale_shift_norm=ss,
                              classifier_resblock_updown=ud, classifier_pool=pool)
        with torch.no_grad():
            for k, v in m.state_dict().items():
                v.copy_(torch.from_numpy(fill_array(k, tuple(v.shape))))
        m.eval()
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            logits = m(x_in, t)
            sel = F.log_softmax(logits, dim=-1)[range(len(logits)), y.view(-1)]
            grad = torch.autograd.grad(sel.sum(), x_in)[0]
        out[f"logits_{tag}"] = logits.detach().numpy()
        out[f"grad_{tag}"] = grad.numpy()
        out[f"nparams_{tag}"] = np.array(sum(p.numel() for p in m.parameters()))
        print(tag, "logits", float(logits.abs().max()), "grad", float(grad.abs().max()), int(out[f"nparams_{tag}"]))
    path = os.path.join(HERE, "clf_variants.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
