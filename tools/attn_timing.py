#!/usr/bin/env python3
"""Per-phase shader cycles of the attention forward tile loop (debug build: `make -C autodiffusion_amd/csrc timing`, run with
ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so).  Wave 0 of every block accumulates s_memtime deltas per phase."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
NAMES = ["loads-issue", "S mfma (+K reads)", "V tr reads", "max (+move)", "mul/exp/cvt", "PV mfma", "lds stores", "barrier"]
lib = _lib.load()
fn = lib.adm_attn_timing_read
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for name, n, t, h, d in [("T=1024 H=6", 256, 1024, 6, 64), ("T=256 H=9", 256, 256, 9, 64), ("T=64 H=12", 256, 64, 12, 64)]:
    qkv = torch.randn(n, t, 3 * h * d, device=DEV).to(torch.bfloat16)
    for _ in range(3):
        ops.attention(qkv, h, True)
    torch.cuda.synchronize()
    buf = np.zeros((65536, 8), dtype=np.uint64)
    fn(buf.ctypes.data, 65536)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.attention(qkv, h, True)
    e1.record()
    torch.cuda.synchronize()
    fn(buf.ctypes.data, 65536)
    b = buf[buf.sum(axis=1) > 0].astype(np.float64)
    ntiles = (t + 63) // 64
    per = b.mean(axis=0) / ntiles
    print(f"{name}: kernel {e0.elapsed_time(e1) * 1e3:.1f} us, {len(b)} blocks, cycles per tile (wave 0): "
          + "  ".join(f"{nm} {v:.0f}" for nm, v in zip(NAMES, per)) + f"  | total {per.sum():.0f}")
