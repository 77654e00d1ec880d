#!/usr/bin/env python3
"""Micro-benchmark of adm_conv on the layer shapes that dominate ADM-G-64 at batch 256 (MI355X).
Prints TFLOP/s (algorithmic 2*M*Cout*Cin*taps) per shape; used to iterate on csrc/adm_conv.hip."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [
    # name, n, hw, c0, c1, cout, taps, prologue, res
    ("unet 192->192 @64 3x3 gn+res", 256, 64, 192, 0, 192, 9, 2, True),
    ("unet 384|192->192 @64 3x3 cat", 256, 64, 192, 192, 192, 9, 2, False),
    ("unet 384->384 @32 3x3", 256, 32, 384, 0, 384, 9, 2, True),
    ("unet 576->576 @16 3x3", 256, 16, 576, 0, 576, 9, 2, True),
    ("unet 768->768 @8 3x3", 256, 8, 768, 0, 768, 9, 2, True),
    ("unet 1536->768 @8 3x3 cat", 256, 8, 768, 768, 768, 9, 2, False),
    ("unet qkv 384->1152 @32 1x1", 256, 32, 384, 0, 1152, 1, 1, False),
    ("unet proj 384->384 @32 1x1", 256, 32, 384, 0, 384, 1, 0, True),
    ("clf 128->128 @64 3x3", 256, 64, 128, 0, 128, 9, 2, True),
    ("clf 256->256 @32 3x3", 256, 32, 256, 0, 256, 9, 2, True),
    ("clf 512->512 @8 3x3", 256, 8, 512, 0, 512, 9, 2, True),
    ("clf bwd 128->128 @64 raw", 256, 64, 128, 0, 128, 9, 0, False),
]

# SHAPESET=wide: the 512- / 1024-channel levels of ADM-G-128 (256 x (1,1,2,3,4)) and LSUN-256 (256 x (1,1,2,2,4,4)): which tile width
# (VARIANT=5: 192, pads 512 -> 576 and 1024 -> 1152; VARIANT=6: 128, no padding) serves them better?
WIDE = [
    ("adm 512->512 @64 3x3 gn+res", 32, 64, 512, 0, 512, 9, 2, True),
    ("adm 512->512 @32 3x3 gn+res", 64, 32, 512, 0, 512, 9, 2, True),
    ("adm 768|512->512 @32 3x3 cat", 64, 32, 768, 512, 512, 9, 2, False),
    ("adm 1024->1024 @16 3x3 gn+res", 64, 16, 1024, 0, 1024, 9, 2, True),
    ("adm 1024->1024 @8 3x3 gn+res", 64, 8, 1024, 0, 1024, 9, 2, True),
    ("adm 256->256 @128 3x3 gn+res", 32, 128, 256, 0, 256, 9, 2, True),
    ("clf 512->512 @8 3x3", 256, 8, 512, 0, 512, 9, 2, True),
    ("clf bwd 512->512 @8 raw", 256, 8, 512, 0, 512, 9, 0, False),
]
if os.environ.get("SHAPESET") == "wide":
    SHAPES = WIDE


def main():
    reps = int(os.environ.get("REPS", "5"))
    variant = int(os.environ.get("VARIANT", "0"))
    only = os.environ.get("ONLY")
    force_raw = os.environ.get("RAW") == "1"
    # DATA=zeros | small (|x| < 2^-6: few mantissa / exponent bits toggle) | default N(0, 1): the chip runs these kernels at its power cap
    # (profiles/r04/power_trace_guided.log), so the operand VALUES change the clock it grants and with it the time of the same instructions
    data = os.environ.get("DATA", "")
    gen = (lambda *sh: torch.zeros(*sh, device=DEV)) if data == "zeros" else (
        (lambda *sh: torch.randn(*sh, device=DEV) * 2.0 ** -8) if data == "small" else (lambda *sh: torch.randn(*sh, device=DEV)))
    tot_f = tot_t = 0.0
    for name, n, hw, c0, c1, cout, taps, prologue, res in SHAPES:
        if only and only not in name:
            continue
        if force_raw:
            prologue = 0
        cin = c0 + c1
        k = 3 if taps == 9 else 1
        x0 = gen(n, hw, hw, c0).to(torch.bfloat16)
        x1 = gen(n, hw, hw, c1).to(torch.bfloat16) if c1 else None
        w = gen(cout, cin, k, k) * (cin * taps) ** -0.5
        wp = ops.pack_conv_weight(w)
        wp32 = ops.pack_conv_weight32(w) if (variant == 7 and taps == 9 and hw >= 16) else None
        if variant == 7 and wp32 is None:
            continue
        b = torch.randn(cout, device=DEV) * 0.1
        aff = (1 + 0.1 * torch.randn(n, cin, device=DEV), 0.1 * torch.randn(n, cin, device=DEV)) if prologue else None
        r = gen(n, hw, hw, cout).to(torch.bfloat16) if res else None
        out = torch.empty(n, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.03:   # warm up by time: the clocks ramp for some milliseconds after an idle spell
            ops.conv(x0, wp, b, cout, taps, x1=x1, aff=aff, silu=(prologue == 2), res=r, variant=variant, out=out, w_packed32=wp32)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            ops.conv(x0, wp, b, cout, taps, x1=x1, aff=aff, silu=(prologue == 2), res=r, variant=variant, out=out, w_packed32=wp32)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * n * hw * hw * cout * cin * taps
        tot_f += fl
        tot_t += ms
        print(f"{name:36s} {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s  (variant {variant})")
    print(f"{'TOTAL':36s} {tot_t * 1e3:9.1f} us  {tot_f / tot_t / 1e9:8.1f} TFLOP/s  (variant {variant}, data {data or 'N(0,1)'})")


if __name__ == "__main__":
    main()
