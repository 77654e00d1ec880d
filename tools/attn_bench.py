#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels on the ADM-G-64 / classifier shapes at batch 256."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
SHAPES = [("unet 32x32 T=1024 H=6", 256, 1024, 6, 64), ("unet 16x16 T=256 H=9", 256, 256, 9, 64),
          ("unet 8x8 T=64 H=12", 256, 64, 12, 64), ("clf 32x32 T=1024 H=4", 256, 1024, 4, 64),
          ("clf 16x16 T=256 H=6", 256, 256, 6, 64), ("clf 8x8 T=64 H=8", 256, 64, 8, 64)]


def timeit(fn, reps=20):
    # warm up by TIME, not by count: the first launches after an idle spell run while the clocks are still ramping
    # (the first shape of this list read 15-20 % low with 2 warm-up launches + 5 timed ones: 578 us vs 482 us for
    # the very same launch later in the run, profiles/r02/attn_bench_sweep.log)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, n, t, h, d in SHAPES:
    qkv = torch.randn(n, t, 3 * h * d, device=DEV).to(torch.bfloat16)
    dout = torch.randn(n, t, h * d, device=DEV).to(torch.bfloat16)
    out, lse = ops.attention(qkv, h, True, want_lse=True)
    fwd = timeit(lambda: ops.attention(qkv, h, True, want_lse=True))
    bwd = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, h, True))
    fl = 4.0 * t * t * d * h * n
    print(f"{name:26s} fwd {fwd * 1e3:8.1f} us {fl / fwd / 1e9:7.1f} TFLOP/s   bwd {bwd * 1e3:8.1f} us {3.5 * fl / bwd / 1e9:7.1f} TFLOP/s")

if os.environ.get("SWEEP"):   # forward only: the same FLOPs through different head counts / batch sizes (row stride, block count)
    for n, t, h, d in [(256, 1024, 6, 64), (384, 1024, 4, 64), (512, 1024, 3, 64), (768, 1024, 2, 64), (192, 1024, 8, 64),
                       (128, 1024, 12, 64), (170, 1024, 6, 64), (128, 1024, 6, 64), (64, 1024, 6, 64), (256, 1024, 4, 64),
                       (128, 1024, 4, 64)]:
        qkv = torch.randn(n, t, 3 * h * d, device=DEV).to(torch.bfloat16)
        fwd = timeit(lambda: ops.attention(qkv, h, True, want_lse=True), reps=20)
        fl = 4.0 * t * t * d * h * n
        print(f"sweep N={n:4d} T={t} H={h:2d} D={d}  blocks {8 * n * h:6d}  fwd {fwd * 1e3:8.1f} us {fl / fwd / 1e9:7.1f} TFLOP/s")
        del qkv
