"""Micro-benchmark of the 1x1 convs (resident-tile kernel / staged kernel) on the shapes that matter (GPU; used while tuning).
SHAPESET=sd: SD v1's 1280-wide transformer projections at a 6-latent half batch (16x16 and 8x8 maps); VARIANT forces adm_conv's tiling
variant (6 = the staged kernel's 128-wide tiles); GEGLU=1 times the fused GEGLU projection where the shape carries one."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402

DEV = "cuda:0"
ADM = [("clf proj 256->256 @32", 256, 32, 256, 256, 0, True, False), ("clf qkv 256->768 @32", 256, 32, 256, 768, 1, False, False),
       ("unet proj 384->384 @32", 256, 32, 384, 384, 0, True, False), ("unet qkv 384->1152 @32", 256, 32, 384, 1152, 1, False, False)]
SD = [("sd o1/o2/q2 1280->1280 +res @16", 6, 16, 1280, 1280, 0, True, False), ("sd proj_in 1280->1280 gn @16", 6, 16, 1280, 1280, 1, False, False),
      ("sd qkv 1280->3840 @16", 6, 16, 1280, 3840, 0, False, False), ("sd ff1 1280->10240 @16", 6, 16, 1280, 10240, 0, False, False),
      ("sd ff1 1280->10240 geglu @16", 6, 16, 1280, 10240, 0, False, True), ("sd o1 1280->1280 +res @8", 6, 8, 1280, 1280, 0, True, False),
      ("sd ff1 1280->10240 geglu @8", 6, 8, 1280, 10240, 0, False, True)]
variant = int(os.environ.get("VARIANT", "0"))
for name, n, hw, cin, cout, pro, res, gg in (SD if os.environ.get("SHAPESET") == "sd" else ADM):
    x = torch.randn(n, hw, hw, cin, device=DEV).to(torch.bfloat16)
    w = torch.randn(cout, cin, device=DEV) * cin ** -0.5
    b = torch.randn(cout, device=DEV) * 0.1
    if gg:
        if variant:
            continue
        w, b = ops.geglu_interleave(w, b)
    wp = ops.pack_conv_weight(w[:, :, None, None])
    aff = (1 + 0.1 * torch.randn(n, cin, device=DEV), 0.1 * torch.randn(n, cin, device=DEV)) if pro else None
    r = torch.randn(n, hw, hw, cout, device=DEV).to(torch.bfloat16) if res else None
    out = None if gg else torch.empty(n, hw, hw, cout, dtype=torch.bfloat16, device=DEV)
    f = lambda: ops.conv(x, wp, b, cout, 1, aff=aff, silu=False, res=r, out=out, variant=variant, geglu=gg)  # noqa: E731
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
    us = e0.elapsed_time(e1) * 50
    byt = (x.numel() + n * hw * hw * cout * (2 if res else 1)) * 2
    print(f"{name:34s} {us:8.1f} us {2.0 * n * hw * hw * cin * cout / us / 1e6:7.1f} TFLOP/s {byt / us / 1e6:6.2f} TB/s  (variant {variant})")
