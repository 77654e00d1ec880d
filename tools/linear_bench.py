#!/usr/bin/env python3
"""Micro-benchmark of adm_linear_f32 on the batched emb_layers shapes (n = 256 rows, k = 768)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
for name, n, k, o in [("unet film", 256, 768, 33792), ("clf film", 256, 512, 12288), ("time_embed 2", 256, 768, 768)]:
    x, w, b = torch.randn(n, k, device=DEV), torch.randn(o, k, device=DEV) * k ** -0.5, torch.randn(o, device=DEV)
    out = torch.empty(n, o, device=DEV)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.03:
        ops.linear_f32(x, w, b, silu_in=True, out=out)
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear_f32(x, w, b, silu_in=True, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:14s} [{n} x {k}] -> {o}: {us:8.1f} us  {2.0 * n * k * o / us / 1e6:6.1f} TFLOP/s  weights {o * k * 4 / us / 1e3:7.1f} GB/s")
