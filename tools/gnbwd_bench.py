#!/usr/bin/env python3
"""GroupNorm(+SiLU) backward on the classifier's largest activation (MI355X): whole batch vs image chunks that fit
the 256 MB Infinity Cache (does the apply pass re-read x / dy from the cache?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"


def run(x, dy, aff, stats, chunk):
    n = x.shape[0]
    outs = []
    for i in range(0, n, chunk):
        sl = slice(i, i + chunk)
        outs.append(ops.gn_bwd(x[sl], dy[sl], (aff[0][sl], aff[1][sl]), stats[sl], True))
    return outs


def main():
    for n, hw, c in [(256, 64, 128), (256, 32, 256), (256, 64, 192)]:
        x = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
        dy = torch.randn(n, hw, hw, c, device=DEV).to(torch.bfloat16)
        aff = (1 + 0.1 * torch.randn(n, c, device=DEV), 0.1 * torch.randn(n, c, device=DEV))
        stats = torch.stack([torch.zeros(n, 32, device=DEV), torch.ones(n, 32, device=DEV)], dim=-1).contiguous()
        for chunk in (256, 128, 64, 32):
            for _ in range(2):
                run(x, dy, aff, stats, chunk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                run(x, dy, aff, stats, chunk)
            e1.record()
            torch.cuda.synchronize()
            print(f"n={n} {hw}x{hw}x{c} chunk={chunk:4d}: {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")


if __name__ == "__main__":
    main()
