#!/usr/bin/env python3
"""conv3x3(nearest-upsample-2x(act(GN(x)))) on the up-ResBlock shapes of ADM-G-64 / SD at bench batch sizes: the one-launch
virtual-upsample path (9 taps per output pixel) against four 2x2-tap phase launches (adm_conv_args.up_phase)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
for name, n, hs, c, pro in [("adm64 up 32->64 @384", 256, 32, 384, True), ("adm64 up 16->32 @576", 256, 16, 576, True),
                            ("sd up 32->64 @640 (6 latents)", 6, 32, 640, False), ("sd up 16->32 @1280 (6 latents)", 6, 16, 1280, False),
                            ("lsun256 up 128->256 @256 (b16)", 16, 128, 256, True)]:
    x = torch.randn(n, hs, hs, c, device=DEV).to(torch.bfloat16)
    w = torch.randn(c, c, 3, 3, device=DEV) * (c * 9) ** -0.5
    wp, wu = ops.pack_conv_weight(w), ops.pack_conv_weight_up(w)
    b = torch.zeros(c, device=DEV)
    aff = (1 + 0.1 * torch.randn(n, c, device=DEV), 0.1 * torch.randn(n, c, device=DEV)) if pro else None
    out = torch.empty(n, 2 * hs, 2 * hs, c, dtype=torch.bfloat16, device=DEV)
    res = {}
    for tag, kw in (("one launch, 9 taps", {}), ("four phases, 4 taps", {"w_up": wu})):
        f = lambda: ops.conv(x, wp, b, c, 9, aff=aff, silu=True, in_up=True, out=out, want_stats=True, **kw)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:
            f()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        res[tag] = e0.elapsed_time(e1) * 50
    fl = 2.0 * n * 4 * hs * hs * c * c * 9
    a, bq = res["one launch, 9 taps"], res["four phases, 4 taps"]
    print(f"{name:34s} one launch {a:8.1f} us ({fl / a / 1e6:6.0f} TFLOP/s)   four phases {bq:8.1f} us ({fl / bq / 1e6:6.0f} algorithmic TFLOP/s)   x{a / bq:.2f}")
