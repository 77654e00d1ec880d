#!/usr/bin/env python3
"""Per-block phase timing of conv_kernel (debug build: `make -C autodiffusion_amd/csrc timing`, then run with
ADM_HIP_LIB=autodiffusion_amd/libadm_hip_timing.so).  Every block's wave 0 stamps s_memtime at kernel entry,
after the prologue (first halo chunk parked), after the K loop and after the epilogue; this prints the mean
of each phase in microseconds (100 MHz constant clock) and how many blocks ran per CU."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import _lib, ops  # noqa: E402

DEV = "cuda:0"
VARIANT = int(os.environ.get("VARIANT", "0"))
SHAPES = [
    ("192->192 @64 3x3 gn+res", 256, 64, 192, 192, 9, 2, True),
    ("384->384 @32 3x3 gn+res", 256, 32, 384, 384, 9, 2, True),
    ("384->384 @32 3x3 raw", 256, 32, 384, 384, 9, 0, False),
    ("768->768 @8 3x3 gn+res", 256, 8, 768, 768, 9, 2, True),
    ("qkv 384->1152 @32 1x1", 256, 32, 384, 1152, 1, 1, False),
    ("proj 384->384 @32 1x1 res", 256, 32, 384, 384, 1, 0, True),
]
# FOLD=1: the out_layers convs of ResBlocks with a skip_connection, with the 1x1 folded in (adm_conv_args.fold0) -- name, n, hw, cin, cout, skip channels
FOLD_SHAPES = [
    ("192->192 @64 + skip 576", 256, 64, 192, 192, 576),
    ("192->192 @64 + skip 384", 256, 64, 192, 192, 384),
    ("384->384 @32 + skip 768", 256, 32, 384, 384, 768),
    ("384->384 @32 + skip 192", 256, 32, 384, 384, 192),
    ("576->576 @16 + skip 1152", 256, 16, 576, 576, 1152),
]


def main():
    lib = _lib.load("f16" if os.environ.get("ADM_TIMING_F16") else "bf16")
    fn = lib.adm_conv_timing_read
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    shapes = SHAPES
    if os.environ.get("FOLD"):
        shapes = [(nm, n, hw, cin, cout, 9, 2, fc) for nm, n, hw, cin, cout, fc in FOLD_SHAPES]
    for name, n, hw, cin, cout, taps, prologue, res in shapes:
        k = 3 if taps == 9 else 1
        DT = torch.float16 if os.environ.get("ADM_TIMING_F16") else torch.bfloat16   # with ADM_HIP_LIB_F16 = the f16 timing build
        x0 = torch.randn(n, hw, hw, cin, device=DEV).to(DT)
        w = torch.randn(cout, cin, k, k, device=DEV) * (cin * taps) ** -0.5
        wp = ops.pack_conv_weight(w, DT)
        b = torch.randn(cout, device=DEV) * 0.1
        aff = (1 + 0.1 * torch.randn(n, cin, device=DEV), 0.1 * torch.randn(n, cin, device=DEV)) if prologue else None
        r = torch.randn(n, hw, hw, cout, device=DEV).to(DT) if res else None
        out = torch.empty(n, hw, hw, cout, dtype=DT, device=DEV)
        kwf = {}
        if os.environ.get("FOLD"):   # `res` carries the skip path's channel count
            fc, r = res, None
            xs = torch.randn(n, hw, hw, fc, device=DEV).to(DT)
            w1 = ops.pack_conv_weight(torch.randn(cout, fc, 1, 1, device=DEV) * fc ** -0.5, DT)
            if os.environ["FOLD"] == "1":
                wp, kwf = ops.fold_weights(wp, w1), dict(fold=(xs, None))
            else:                    # FOLD=0: the same layer as two launches; the stamps are those of the 3x3 conv with the residual
                r = ops.conv(xs, w1, b, cout, 1)
        try:
            ops.conv(x0, wp, b, cout, taps, aff=aff, silu=(prologue == 2), res=r, out=out, variant=VARIANT, **kwf)
        except _lib.AdmError as e:   # an explicit variant that does not take this shape (e.g. VARIANT=8: 3x3 on maps >= 16x16 only)
            print(f"{name:28s} skipped: {str(e).split(': ', 2)[-1][:90]}")
            continue
        for _ in range(2):
            ops.conv(x0, wp, b, cout, taps, aff=aff, silu=(prologue == 2), res=r, out=out, variant=VARIANT, **kwf)
        torch.cuda.synchronize()
        scratch = np.zeros((16384, 16), dtype=np.uint64)
        fn(scratch.ctypes.data, 16384)  # clears the device buffer
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        for _ in range(int(os.environ.get("PRE", "0"))):  # sustained load before the stamped launch
            ops.conv(x0, wp, b, cout, taps, aff=aff, silu=(prologue == 2), res=r, out=out, variant=VARIANT, **kwf)
        e0.record()
        ops.conv(x0, wp, b, cout, taps, aff=aff, silu=(prologue == 2), res=r, out=out, variant=VARIANT, **kwf)
        e1.record()
        torch.cuda.synchronize()
        buf = np.zeros((16384, 16), dtype=np.uint64)
        rc = fn(buf.ctypes.data, 16384)
        assert rc == 0, rc
        buf = buf[buf[:, 6] > 0]
        if not len(buf):   # served by a kernel without stamps (the resident-tile 1x1 kernel)
            print(f"{name:28s} no stamps (not conv_kernel)")
            continue
        t = buf[:, :7].astype(np.int64)
        kern_us = e0.elapsed_time(e1) * 1e3
        tick = 100.0  # s_memrealtime: 100 MHz
        span = (t[:, 6].max() - t[:, 0].min()) / tick
        d = np.diff(t, axis=1) / tick
        hw_id, xcc = buf[:, 14].astype(np.int64), buf[:, 15].astype(np.int64) & 0xF
        cu = (xcc << 8) | (((hw_id >> 13) & 7) << 5) | (((hw_id >> 12) & 1) << 4) | ((hw_id >> 8) & 0xF)
        ncu = len(np.unique(cu))
        clk = ((buf[:, 9].astype(np.int64) - buf[:, 8].astype(np.int64)) / np.maximum(t[:, 2] - t[:, 1], 1)) * 100.0  # MHz
        if os.environ.get("XCD_REPORT"):   # per-XCD balance of the static tile partition: when does each XCD's last tile end?
            t0 = t[:, 0].min()
            ends, busy, kclk = [], [], []
            for xc in range(8):
                m = xcc == xc
                if not m.any():
                    continue
                ends.append((t[m, 6].max() - t0) / tick)
                ncu_x = len(np.unique(cu[m]))
                busy.append((t[m, 6] - t[m, 0]).sum() / tick / max(ncu_x, 1))
                kclk.append(np.median(clk[m]))
            per_cu_end = np.array([(t[cu == c, 6].max() - t0) / tick for c in np.unique(cu)])
            print(f"    per XCD: last tile ends at {' '.join(f'{e:7.1f}' for e in ends)} us | busy per CU {' '.join(f'{b:7.1f}' for b in busy)} us | "
                  f"k-loop clock {' '.join(f'{k:5.0f}' for k in kclk)} MHz")
            print(f"    per CU: end of last tile min {per_cu_end.min():.1f} / median {np.median(per_cu_end):.1f} / max {per_cu_end.max():.1f} us "
                  f"(spread {100 * (per_cu_end.max() - per_cu_end.min()) / per_cu_end.max():.1f} % of the kernel)")
        names = ["wait", "k-loop", "switch+stage", "park-next", "sweep", "stats"]
        sub = buf[:, [2, 10, 11, 12, 13, 3]].astype(np.int64)
        subd = np.diff(sub, axis=1) / tick
        tot = (t[:, 6] - t[:, 0]) / tick
        print(f"{name:28s} kernel {kern_us:8.1f} us | {len(buf)} tiles on {ncu} CUs | span {span:8.1f} us | k-loop clock {np.median(clk):6.0f} MHz | block {tot.mean():7.2f} us "
              f"(sum/CU {tot.sum() / ncu:8.1f}) | "
              + "  ".join(f"{nm} {d[:, i].mean():6.2f}" for i, nm in enumerate(names))
              + " | switch: " + "  ".join(f"{nm} {subd[:, i].mean():5.2f}" for i, nm in enumerate(["bases", "stage", "setup+dma+addr", "loads", "barrier"])))


if __name__ == "__main__":
    main()
