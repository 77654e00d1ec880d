#!/usr/bin/env python3
"""Variant 8 (4 waves, one per SIMD, 512 registers) against the production tiling on the 3x3 shapes of tools/conv_bench.py: the K
order per output element is the same (chunks, taps, one MFMA chain), so the outputs and the fused statistics must be BITWISE equal."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
CASES = [  # n, hw, c0, c1, cout, prologue, res, in_up / res_up
    (3, 64, 192, 0, 192, 2, True, False), (2, 64, 192, 192, 192, 2, False, False), (3, 32, 384, 0, 384, 2, True, False),
    (5, 16, 576, 0, 576, 1, True, False), (2, 32, 128, 0, 128, 0, False, False), (2, 16, 64, 32, 96, 2, True, False),
    (2, 32, 192, 0, 192, 2, True, True), (1, 16, 32, 0, 200, 0, True, False),
]
ok = True
for n, hw, c0, c1, cout, pro, res, up in CASES:
    g = torch.Generator(device=DEV).manual_seed(hw + c0 + cout)
    hs = hw // 2 if up else hw
    x0 = torch.randn(n, hs, hs, c0, device=DEV, generator=g).to(torch.bfloat16)
    x1 = torch.randn(n, hw, hw, c1, device=DEV, generator=g).to(torch.bfloat16) if c1 else None
    cin = c0 + c1
    w = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) * (cin * 9) ** -0.5
    wp = ops.pack_conv_weight(w)
    b = torch.randn(cout, device=DEV, generator=g) * 0.1
    aff = (1 + 0.1 * torch.randn(n, cin, device=DEV, generator=g), 0.1 * torch.randn(n, cin, device=DEV, generator=g)) if pro else None
    r = torch.randn(n, hs, hs, cout, device=DEV, generator=g).to(torch.bfloat16) if res else None
    outs = []
    for v in (0, int(os.environ.get("CHECK_VARIANT", "8"))):
        o = ops.conv(x0, wp, b, cout, 9, x1=x1, aff=aff, silu=(pro == 2), res=r, variant=v, want_stats=True, in_up=up, res_up=up and res)
        torch.cuda.synchronize()
        outs.append((o, getattr(o, "_adm_stats", (None,))[0]))
    eq = torch.equal(outs[0][0], outs[1][0])
    eqs = outs[0][1] is None or torch.equal(outs[0][1], outs[1][1])
    d = (outs[0][0].float() - outs[1][0].float()).abs().max().item()
    print(f"n={n} hw={hw} cin={c0}|{c1} cout={cout} pro={pro} res={res} up={up}: out equal {eq} (max diff {d:.3g}), stats equal {eqs}, finite {bool(torch.isfinite(outs[1][0].float()).all())}")
    ok &= eq and eqs
print("ALL EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
