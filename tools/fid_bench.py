#!/usr/bin/env python3
"""Time adm_fid_accumulate (f64-MFMA symmetric-half Gram) at the sizes of the search: one candidate's 5000 x 2048
Inception activations in one call, and the per-batch calls (256 x 2048).  Prints TFLOP/s on the algorithmic
2 * n * d^2 (the kernel itself computes a little over half of it) and GB/s on the S2 read-modify-write."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd.fid import ActivationAccumulator  # noqa: E402

DEV = "cuda:0"
for n, d in ((5000, 2048), (256, 2048), (100, 2048)):
    acts = torch.randn(n, d, device=DEV)
    acc = ActivationAccumulator(d, DEV)
    for _ in range(2):
        acc.add(acts)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    reps = 5
    e0.record()
    for _ in range(reps):
        acc.add(acts)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"fid_accumulate n={n} d={d}: {ms * 1e3:9.1f} us  {2.0 * n * d * d / ms / 1e9:7.2f} TFLOP/s (algorithmic 2nd^2, f64)  "
          f"{2 * d * d * 8 / ms / 1e6:7.1f} GB/s (S2 read+write)")
