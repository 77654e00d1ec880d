#!/usr/bin/env python3
"""Full-size parity of the two late-round conv modes against the reference-captured fixtures (tests/golden/full_*.npz): the
ADM-G-64 UNet with the up-ResBlock convs as one launch / four phase launches, and the 64x64 classifier's guidance gradient
with the GroupNorm-backward sums taken by a separate pass / in the backward conv's epilogue."""
import sys, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_fullsize as T
g = T.golden("full_adm64")
model, diffusion = T.adm64()
x, t, y = (torch.from_numpy(g[k]).to(T.DEV) for k in ("x", "t", "y"))
for ph in (False, True):
    model.upconv_phases = ph
    out = model(x, t, y)
    print("ADM-64 UNet, up-conv phases", ph, "rel vs reference fp32:", T.rel(out, g["out"]))
gc = T.golden("full_clf64"); c64 = T.clf(64, 4)
xc, tc, yc = (torch.from_numpy(gc[k]).to(T.DEV) for k in ("x", "t", "y"))
for fu in (False, True):
    c64.fuse_gn_bwd = fu
    grad, logits = c64.log_prob_grad(xc, tc, yc, 1.0, return_logits=True)
    print("64x64 classifier gradient, GN-backward epilogue fusion", fu, "rel vs reference autograd:", T.rel(grad, gc["grad"]))
