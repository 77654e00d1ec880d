#!/usr/bin/env python3
"""One shape of the GroupNorm(+SiLU) backward chain, many times: run under `rocprofv3 --kernel-trace --stats` to split the
chain's time into its three kernels (SHAPE=n,hw,c; default 256,64,128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

n, hw, c = (int(v) for v in os.environ.get("SHAPE", "256,64,128").split(","))
DEV = "cuda:0"
x = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
dy = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
aff = (1 + 0.1 * torch.randn(n, c, device=DEV), 0.1 * torch.randn(n, c, device=DEV))
stats = torch.stack([torch.zeros(n, 32, device=DEV), torch.ones(n, 32, device=DEV)], dim=-1).contiguous()
for _ in range(50):
    ops.gn_bwd(x, dy, aff, stats, os.environ.get("SILU", "1") == "1")
torch.cuda.synchronize()
