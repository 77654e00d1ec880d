#!/usr/bin/env python3
"""Throughput of the HIP Inception-v3 pool3 extractor (autodiffusion_amd/inception.py) on uint8 batches as the sampler
leaves them: images/s, model TFLOP/s (11.46 GFLOP per image = 2 x 5.73 GMAC at 299 x 299), and the largest layers'
per-launch rates (HIP events around single adm_conv2d launches)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402
from autodiffusion_amd.inception import CONVS, InceptionV3  # noqa: E402

DEV = "cuda:0"
GFLOP_PER_IMAGE = 11.46


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.05:
        fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    for dt in (torch.float16, torch.bfloat16):
        m = InceptionV3(dtype=dt).to(DEV)
        m.weights_loaded = True   # random weights on purpose: a throughput run
        for size, n, chunk in ((64, 100, 100), (256, 100, 100), (64, 256, 256), (64, 640, 100), (64, 640, 200), (64, 640, 320)):
            m.CHUNK = chunk
            u8 = torch.randint(0, 256, (n, size, size, 3), dtype=torch.uint8, device=DEV)
            ms = timeit(lambda: m.features(u8), reps=5)
            print(f"{str(dt):16s} {n:4d} x {size}x{size} uint8 (passes of {chunk}) -> pool3: {ms:8.2f} ms  {n / ms * 1e3:8.0f} images/s  "
                  f"{GFLOP_PER_IMAGE * n / ms:7.1f} TFLOP/s")
    # single layers at batch 100 (fp16)
    n = 100
    for name, cin, cout, kh, kw, stride, pad, hw in [("Conv2d_2b_3x3", 32, 64, 3, 3, 1, (1, 1), 147), ("Conv2d_4a_3x3", 80, 192, 3, 3, 1, (0, 0), 73),
                                                     ("Mixed_5x 5x5", 48, 64, 5, 5, 1, (2, 2), 35), ("Mixed_6a 3x3 s2", 288, 384, 3, 3, 2, (0, 0), 35),
                                                     ("Mixed_6x 1x7", 160, 160, 1, 7, 1, (0, 3), 17), ("Mixed_6x 1x1", 768, 192, 1, 1, 1, (0, 0), 17),
                                                     ("Mixed_7x 3x3", 448, 384, 3, 3, 1, (1, 1), 8), ("Mixed_7c 1x1", 2048, 320, 1, 1, 1, (0, 0), 8)]:
        cp = (cin + 31) // 32 * 32
        x = torch.randn(n, hw, hw, cp, device=DEV).to(torch.float16)
        w = ops.pack_conv2d_weight(torch.randn(cout, cin, kh, kw, device=DEV) * (cin * kh * kw) ** -0.5, None, torch.float16)
        b = torch.zeros(cout, device=DEV)
        oh = (hw + 2 * pad[0] - kh) // stride + 1
        out = torch.empty(n, oh, (hw + 2 * pad[1] - kw) // stride + 1, cout, dtype=torch.float16, device=DEV)
        ms = timeit(lambda: ops.conv2d(x, w, b, kh, kw, stride, pad, True, out=out), reps=20)
        fl = 2.0 * out.numel() * cin * kh * kw
        print(f"  {name:18s} {cin:4d}->{cout:4d} {kh}x{kw} s{stride} @{hw:3d}: {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
