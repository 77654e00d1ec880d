"""Tensor-level wrappers over the C ABI (include/adm_hip.h).

PyTorch is used here only as the owner of device memory and of the HIP
stream: every function takes torch tensors, checks dtype / contiguity, and
passes raw device pointers to libadm_hip.so.  No torch compute op runs on
the hot path; if the shared library is missing these functions raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import AdmError, ConvArgs, StepCoefs, check

BF16 = torch.bfloat16
F16 = torch.float16   # the reference's torso type (use_fp16=True): libadm_hip_f16.so, same kernels built for IEEE half


def _L(t):
    """The library built for tensor t's 16-bit element type (bf16: libadm_hip.so, fp16: libadm_hip_f16.so)."""
    if t.dtype == BF16:
        return _lib.load()
    if t.dtype == F16:
        return _lib.load("f16")
    raise AdmError(f"expected a bfloat16 or float16 activation tensor, got {t.dtype}")
GN_EPS = 1e-5

# conv epilogues accumulate the GroupNorm partial sums of their output (consumed by gn_affine)
USE_FUSED_STATS = True

# bench.py sets this to a list to time the dominant kernel with HIP events on the launch stream:
# every conv launch appends (start_event, end_event, algorithmic_flops, (variant, taps, big_map, prologue)).
CONV_PROFILE = None       # bench.py: list receiving (event0, event1, flops, key, shape) per conv launch
CONV_PROFILE_KEY = None   # ... restricted to launches with this (variant, taps, big map, prologue) key


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor], dtype=None, name="tensor") -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise AdmError(f"{name}: expected a device tensor (the HIP path has no CPU fallback)")
    if not t.is_contiguous():
        raise AdmError(f"{name}: must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise AdmError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # kernels launch in the CURRENT device's context on its current stream (_stream): a pointer of another GPU
        # there is a memory fault or a silent cross-device access
        raise AdmError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}; "
                       "call torch.cuda.set_device(tensor.device) (dist_util.setup_dist does) before using the HIP path")
    return t.data_ptr()


# ------------------------------------------------------------------ sampler
def sampler_step(kind: str, x, model_out, coefs: StepCoefs, grad=None, noise=None,
                 want_xstart=False, want_u8=False):
    """One ddim / ddpm update.  Returns (x_prev, pred_xstart | None, uint8 NHWC | None)."""
    n, c, h, w = x.shape
    lib = _lib.load()
    x_prev = torch.empty_like(x)
    x0 = torch.empty_like(x) if want_xstart else None
    u8 = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device) if want_u8 else None
    exp_c = 2 * c if coefs.learned_range else c
    if tuple(model_out.shape) != (n, exp_c, h, w):
        raise AdmError(f"model_out shape {tuple(model_out.shape)} != {(n, exp_c, h, w)}")
    fn = lib.adm_ddim_step if kind == "ddim" else lib.adm_ddpm_step
    check(fn(_ptr(x, torch.float32, "x"), _ptr(model_out, torch.float32, "model_out"),
             _ptr(grad, torch.float32, "grad"), _ptr(noise, torch.float32, "noise"),
             _ptr(x_prev), _ptr(x0), _ptr(u8), n, c, h, w, C.byref(coefs), _stream()),
          f"adm_{kind}_step")
    return x_prev, x0, u8


def pack_u8_nhwc(x):
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    check(_lib.load().adm_pack_u8_nhwc(_ptr(x, torch.float32, "x"), _ptr(out), n, c, h, w, _stream()),
          "adm_pack_u8_nhwc")
    return out


# ------------------------------------------------------------------ embeddings
def timestep_embedding(t, dim: int, max_period: float = 10000.0):
    t = t.to(torch.float32).contiguous()
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    check(_lib.load().adm_timestep_embedding(_ptr(t), _ptr(out), t.shape[0], dim, max_period, _stream()),
          "adm_timestep_embedding")
    return out


def linear_f32(x, w, b=None, silu_in=False, table=None, idx=None, out=None):
    n, k = x.shape
    o = w.shape[0]
    if w.shape[1] != k:
        raise AdmError(f"linear: weight {tuple(w.shape)} vs input {tuple(x.shape)}")
    if out is None:
        out = torch.empty((n, o), dtype=torch.float32, device=x.device)
    check(_lib.load().adm_linear_f32(_ptr(x, torch.float32, "x"), _ptr(w, torch.float32, "w"),
                                     _ptr(b, torch.float32, "b"), _ptr(table, torch.float32, "table"),
                                     _ptr(idx, torch.int64, "idx"), _ptr(out), n, k, o, int(silu_in), _stream()),
          "adm_linear_f32")
    return out


# ------------------------------------------------------------------ stem / norm / resample
def stem_conv3x3(x_nchw, w, b, dtype=BF16):
    n, cin, h, wd = x_nchw.shape
    cout = w.shape[0]
    out = torch.empty((n, h, wd, cout), dtype=dtype, device=x_nchw.device)
    check(_L(out).adm_stem_conv3x3(_ptr(x_nchw, torch.float32, "x"), _ptr(w, torch.float32, "w"),
                                       _ptr(b, torch.float32, "b"), _ptr(out), n, cin, h, wd, cout, _stream()),
          "adm_stem_conv3x3")
    return out


def nchw_to_nhwc_pad(x_nchw, cpad: int = 32, dtype=BF16):
    """fp32 NCHW image -> bf16 NHWC with channels zero-padded to cpad (input of the MFMA stem conv)."""
    n, c, h, w = x_nchw.shape
    out = torch.empty((n, h, w, cpad), dtype=dtype, device=x_nchw.device)
    check(_L(out).adm_nchw_to_nhwc_pad(_ptr(x_nchw, torch.float32, "x"), _ptr(out), n, c, h, w, cpad, _stream()),
          "adm_nchw_to_nhwc_pad")
    return out


GN_BWD_SLAB_PIXELS = 256   # pixels per block of the backward partial pass (64 / 128 / 256 / 512 measured: 256 is 3-4 % ahead)


def gn_slabs(hw: int) -> int:
    return max(1, min(64, hw // 64))


def gn_affine(x0, gamma, beta, x1=None, film=None, film_stride=0, partial=None, want_stats=False, eps=None, add=None):
    """GroupNorm32 statistics of the virtual concat (x0 | x1) -> per-(image, channel) affine (a, b).

    film: fp32 view whose row n, columns [0:C) = scale and [C:2C) = shift (row stride film_stride).
    want_stats: also return fp32 [N, 32, 2] (mean, rstd) for the backward-data pass.
    eps: GroupNorm epsilon (default GroupNorm32's 1e-5; the SpatialTransformer's Normalize uses 1e-6).
    add: fp32 [N, C]: the affine is that of GroupNorm(x0 + add[:, :, None, None]) applied to the stored x0.
    """
    eps = GN_EPS if eps is None else float(eps)
    n, h, w, c0 = x0.shape
    c1 = 0 if x1 is None else x1.shape[3]
    c, hw = c0 + c1, h * w
    lib = _L(x0)
    a = torch.empty((n, c), dtype=torch.float32, device=x0.device)
    b = torch.empty((n, c), dtype=torch.float32, device=x0.device)
    film_ptr = None
    if film is not None:
        if film.dtype != torch.float32 or not film.is_cuda:
            raise AdmError("film must be a float32 device tensor")
        film_ptr = film.data_ptr()
    stats = torch.empty((n, 32, 2), dtype=torch.float32, device=x0.device) if want_stats else None
    fused0 = getattr(x0, "_adm_stats", None)
    fused1 = getattr(x1, "_adm_stats", None) if x1 is not None else None
    if add is not None:
        if x1 is not None or film is not None:
            raise AdmError("gn_affine(add=...) takes one source and no FiLM")
        if add.dtype != torch.float32 or add.shape != (n, c) or add.stride(1) != 1:
            raise AdmError("add must be float32 [N, C] with unit channel stride")
        if USE_FUSED_STATS and fused0 is not None:
            part, slabs = fused0
        else:
            slabs = gn_slabs(hw)
            part = torch.empty((n, slabs, c, 2), dtype=torch.float32, device=x0.device)
            check(lib.adm_gn_partial(_ptr(x0, x0.dtype, "x0"), c0, None, 0, _ptr(part), n, hw, slabs, _stream()), "adm_gn_partial")
        check(lib.adm_gn_finalize_add(_ptr(part), _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"),
                                      add.data_ptr(), add.stride(0), _ptr(a), _ptr(b), _ptr(stats), n, c, hw, slabs, eps, _stream()),
              "adm_gn_finalize_add")
        return (a, b, stats) if want_stats else (a, b)
    if USE_FUSED_STATS and fused0 is not None and (x1 is None or fused1 is not None):
        # the producing conv already accumulated sum / sum-of-squares of this tensor in its epilogue
        p1, s1 = (fused1 if fused1 is not None else (None, 0))
        check(lib.adm_gn_finalize2(_ptr(fused0[0]), c0, fused0[1], _ptr(p1), c1, s1, _ptr(gamma, torch.float32, "gamma"),
                                   _ptr(beta, torch.float32, "beta"), film_ptr, film_stride, _ptr(a), _ptr(b),
                                   _ptr(stats), n, hw, eps, _stream()), "adm_gn_finalize2")
    else:
        slabs = gn_slabs(hw)
        if partial is None:
            partial = torch.empty((n, slabs, c, 2), dtype=torch.float32, device=x0.device)
        check(lib.adm_gn_partial(_ptr(x0, x0.dtype, "x0"), c0, _ptr(x1, x0.dtype, "x1"), c1, _ptr(partial), n, hw, slabs,
                                 _stream()), "adm_gn_partial")
        check(lib.adm_gn_finalize(_ptr(partial), _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"),
                                  film_ptr, film_stride, _ptr(a), _ptr(b), _ptr(stats), n, c, hw, slabs, eps,
                                  _stream()), "adm_gn_finalize")
    if want_stats:
        return a, b, stats
    return a, b


def resample(x, mode: str, aff=None):
    """mode 'down' = AvgPool2d(2), 'up' = nearest x2, 'stride2' = every second pixel, 'zero2' = zero-insert x2 (out[2y][2x] = x[y][x]:
    what a stride-2 conv's backward-data conv reads); aff=(a, b) applies SiLU(a*x+b) first."""
    n, h, w, c = x.shape
    m = {"down": 1, "up": 2, "stride2": 3, "zero2": 4}[mode]
    oh, ow = (h * 2, w * 2) if m in (2, 4) else (h // 2, w // 2)
    out = torch.empty((n, oh, ow, c), dtype=x.dtype, device=x.device)
    a, b = aff if aff is not None else (None, None)
    check(_L(x).adm_resample(_ptr(x, x.dtype, "x"), _ptr(a, torch.float32), _ptr(b, torch.float32), _ptr(out),
                                   n, h, w, c, m, _stream()), "adm_resample")
    return out


# ------------------------------------------------------------------ conv / GEMM
def pack_conv_weight(w, dtype=BF16):
    """fp32 [cout, cin, kh, kw] or [cout, cin, 1] or [cout, cin] -> packed bf16 image (1-D tensor)."""
    cout, cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    lib = _lib.load("f16" if dtype == F16 else "bf16")
    elems = lib.adm_packed_weight_elems(cout, cin, taps)
    if elems < 0:
        raise AdmError(f"pack_conv_weight: unsupported weight shape {tuple(w.shape)} (cin % 32 == 0, taps 1|9)")
    w32 = w.detach().to(torch.float32).contiguous()
    out = torch.empty((elems,), dtype=dtype, device=w.device)
    check(lib.adm_pack_conv_weight(_ptr(w32), _ptr(out), cout, cin, taps, _stream()), "adm_pack_conv_weight")
    return out


FUSE_GN_BWD = os.environ.get("ADM_FUSE_GN_BWD", "1") != "0"   # GroupNorm-backward partial sums in the producing backward conv's epilogue
UPCONV_PHASES = os.environ.get("ADM_UPCONV_PHASES", "1") != "0"   # conv3x3(upsample2x(x)) as four 2x2-tap phase convs (4/9 of the MACs)


def up_phase_weights(w):
    """fp32 [4, cout, cin, 3, 3]: for phase ph = 2 py + px the weights of the 3x3 conv on the HALF-resolution source x that gives
    the output pixels (2y + py, 2x + px) of conv3x3(nearest-upsample-2x(x), w).  The nine taps collapse onto a 2x2 window: window
    rows {y-1, y, y+1} carry {w0, w1 + w2, 0} for py = 0 and {0, w0 + w1, w2} for py = 1, likewise for columns (5 zero taps).
    Pure tensor algebra (testable without a GPU)."""
    w32 = w.detach().to(torch.float32)
    rows = {0: [[0], [1, 2], []], 1: [[], [0, 1], [2]]}   # window row r <- original taps, by phase
    out = torch.zeros((4,) + tuple(w32.shape), dtype=torch.float32, device=w32.device)
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        for r in range(3):
            for c in range(3):
                for ky in rows[py][r]:
                    for kx in rows[px][c]:
                        out[ph, :, :, r, c] += w32[:, :, ky, kx]
    return out


def pack_conv_weight_up(w, dtype=BF16):
    """The four phase weights of conv3x3(nearest-upsample-2x(x), w) (adm_conv_args.up_phase; up_phase_weights), summed in fp32
    and packed like any 3x3 weight: a [4, elems] tensor."""
    wp = up_phase_weights(w)
    return torch.stack([pack_conv_weight(wp[ph], dtype) for ph in range(4)], 0).contiguous()


def pack_conv_weight32(w, dtype=BF16):
    """Same weight in the 32x32x16 fragment order (enables adm_conv's variant 7)."""
    cout, cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    lib = _lib.load("f16" if dtype == F16 else "bf16")
    elems = lib.adm_packed_weight32_elems(cout, cin, taps)
    if elems < 0:
        raise AdmError(f"pack_conv_weight32: unsupported weight shape {tuple(w.shape)}")
    w32 = w.detach().to(torch.float32).contiguous()
    out = torch.empty((elems,), dtype=dtype, device=w.device)
    check(lib.adm_pack_conv_weight32(_ptr(w32), _ptr(out), cout, cin, taps, _stream()), "adm_pack_conv_weight32")
    return out


def splitk_for(h: int, w: int, cin: int) -> int:
    """Split-K factor of a 3x3 conv by SHAPE only (never by batch: the fp32 summation order, hence the result, must not
    depend on how many images ride along): 4 on 8x8 maps, 2 on 16x16 maps, where the K loop divides into even runs of at
    least 4 chunks.  For small-batch callers (the SD latent UNet at n_samples 6): a launch there has a handful of tiles
    with 40-80-chunk K loops on 256 CUs."""
    chunks = cin // 32
    want = 4 if h * w <= 64 else (2 if h * w <= 256 else 1)   # measured on SD v1 at 6-latent half batches: 4 / 2 / none at 8x8 / 16x16 / 32x32
    while want > 1 and (chunks % want or (chunks // want) % 2 or chunks // want < 4):
        want //= 2
    return max(1, want)


SPLITK_1X1_MIN_CHUNKS = int(os.environ.get("ADM_SPLITK_1X1_MIN_CHUNKS", "32"))


def splitk_1x1_for(h: int, w: int, cin: int, cout: int) -> int:
    """Split-K factor of a 1x1 conv by SHAPE only (as splitk_for): the wide projections of the 16x16 / 8x8 levels of a small-batch
    model (SD v1: 1280 -> 1280 attention / feed-forward outputs, 5120 -> 1280 after the GEGLU) have 30-60 output tiles with
    40-160-chunk K loops on 256 CUs.  Runs of an even number of >= 8 chunks; outputs wider than 1280 have tiles enough."""
    chunks = cin // 32
    if h * w > 256 or chunks < SPLITK_1X1_MIN_CHUNKS or cout > 1280:
        return 1
    want = 8
    while want > 1 and (chunks % want or (chunks // want) % 2 or chunks // want < 8):
        want //= 2
    return max(1, want)


def conv(x0, w_packed, bias, cout: int, taps: int, x1=None, aff=None, silu=True, res=None,
         out_f32_nchw=False, variant=0, out=None, w_packed32=None, want_stats=False, in_up=False, res_up=False, ksplit=1,
         w_up=None, gnb=None, geglu=False, fold=None, out_scale=None):
    """Fused [GN(+FiLM) affine (+SiLU)] -> conv (3x3 pad 1 | 1x1) -> +bias (+res).

    x0 (| x1): 16-bit NHWC (bf16, or fp16 for an fp16-torso model: the library is picked by x0's dtype).  Returns the same
    type NHWC [n,h,w,cout] or fp32 NCHW [n,cout,h,w].  ksplit > 1: split-K schedule for small batches (see splitk_for).
    in_up / res_up: x0 / res are at half resolution and are read through a virtual nearest-neighbour 2x upsample
    (the output is [n, 2h, 2w, cout]): ResBlock(up=True) without materialising the upsampled tensors.  With w_up
    (pack_conv_weight_up) an in_up conv runs as four phase launches of 2x2 live taps each (4/9 of the MACs).
    geglu: w_packed / bias are an interleaved (value, gate) projection (geglu_interleave); returns [n, h, w, cout // 2] =
    value * gelu(gate) -- the Stable-Diffusion GEGLU without the [.., cout] tensor (1x1 resident-tile kernel only).
    fold=(xs0, xs1 | None): the ResBlock's skip_connection inside this (out_layers) conv: w_packed = fold_weights(3x3, 1x1), bias = the
    sum of both biases; extra one-tap K-steps over the block input (xs0 | xs1) replace the 1x1 launch and the residual operand.
    out_scale: fp32 NCHW output only -- (acc + bias) * out_scale in the fp32 epilogue.
    """
    n, h, w, c0 = x0.shape
    if (in_up and w_up is not None and UPCONV_PHASES and taps == 9 and x1 is None and res is None and not res_up
            and not out_f32_nchw and ksplit <= 1 and h >= 16 and w >= 16 and (h * w) % 256 == 0 and variant == 0):
        return _conv_up_phases(x0, w_up, bias, cout, aff, silu, out, want_stats)
    if in_up:
        h, w = 2 * h, 2 * w
    c1 = 0 if x1 is None else x1.shape[3]
    dev = x0.device
    if geglu and (taps != 1 or res is not None or aff is not None or want_stats or out_f32_nchw or gnb is not None or ksplit > 1):
        raise AdmError("conv(geglu=True): a raw 1x1 projection without residual / statistics")
    if out is None:
        out = (torch.empty((n, cout, h, w), dtype=torch.float32, device=dev) if out_f32_nchw
               else torch.empty((n, h, w, cout // 2 if geglu else cout), dtype=x0.dtype, device=dev))
    a = ConvArgs()
    a.geglu = int(bool(geglu))
    if out_scale is not None:
        if not out_f32_nchw:
            raise AdmError("conv(out_scale=...): the fp32 NCHW epilogue only")
        a.out_scale = float(out_scale)
    lib = _L(x0)
    a.in0, a.in1 = _ptr(x0, x0.dtype, "x0"), _ptr(x1, x0.dtype, "x1")
    a.w_packed, a.bias = _ptr(w_packed, x0.dtype, "w_packed"), _ptr(bias, torch.float32, "bias")
    if gnb is not None:
        # backward-data conv whose epilogue is the first half of the GroupNorm + SiLU backward of the layer in front:
        # gnb = (x, (a, b)) of that layer; the result is dz = conv(...) * SiLU'(a x + b) with out._adm_stats = (sum dz, sum dz x)
        if aff is not None or res is not None or taps != 9 or out_f32_nchw or in_up or res_up or ksplit > 1:
            raise AdmError("conv(gnb=...): a raw 3x3 backward-data conv with bf16 output")
        gx, (ga, gb_) = gnb
        if tuple(gx.shape) != (n, h, w, cout):
            raise AdmError(f"conv(gnb=...): x {tuple(gx.shape)} must have the output's shape {(n, h, w, cout)}")
        a.aff_a, a.aff_b = _ptr(ga, torch.float32, "gnb a"), _ptr(gb_, torch.float32, "gnb b")
        a.prologue = 3
        res, want_stats = gx, True
    elif aff is not None:
        a.aff_a, a.aff_b = _ptr(aff[0], torch.float32, "aff_a"), _ptr(aff[1], torch.float32, "aff_b")
        a.prologue = 2 if silu else 1
    else:
        a.prologue = 0
    a.res = _ptr(res, x0.dtype, "res")
    a.out = _ptr(out)
    a.n, a.h, a.w, a.c0, a.c1, a.cout = n, h, w, c0, c1, cout
    a.w_packed32 = _ptr(w_packed32, x0.dtype, "w_packed32")
    a.taps, a.out_mode, a.variant = taps, int(out_f32_nchw), variant
    a.in_up, a.res_up = int(in_up), int(res_up)
    if fold is not None:
        f0, f1 = fold
        if tuple(f0.shape[:3]) != (n, h, w) or (f1 is not None and tuple(f1.shape[:3]) != (n, h, w)):
            raise AdmError("conv(fold=...): the folded skip input must have the output's map")
        a.fold0, a.fold1 = _ptr(f0, x0.dtype, "fold0"), _ptr(f1, x0.dtype, "fold1")
        a.fc0, a.fc1 = f0.shape[3], 0 if f1 is None else f1.shape[3]
    ws = None
    if ksplit > 1:
        if out_f32_nchw or res_up:
            raise AdmError("conv(ksplit > 1): 16-bit NHWC output only, no res_up")
        ws = torch.empty((ksplit, n * h * w, cout), dtype=torch.float32, device=dev)
        a.ksplit, a.ws = int(ksplit), ws.data_ptr()
    if in_up or res_up:
        w_packed32 = None  # the 32x32x16 kernel does not take the virtual upsample
    if variant == 0:
        variant = lib.adm_conv_pick_variant(C.byref(a))  # the library's own rule (incl. the resident-tile 1x1 kernel)
        if variant == 5 and taps == 9 and w_packed32 is not None and h >= 16 and w >= 16 and not out_f32_nchw and ksplit <= 1 and fold is None:
            variant = 7  # 3x3 on >= 16x16 maps, Cout a multiple of 192: the 32x32x16 MFMA kernel
        a.variant = variant
    fused = None
    if want_stats and (USE_FUSED_STATS or gnb is not None):
        slabs = lib.adm_conv_stat_slabs(C.byref(a))
        if slabs > 0:
            fused = (torch.empty((n, slabs, cout, 2), dtype=torch.float32, device=dev), slabs)
            a.out_stats = fused[0].data_ptr()
            out._adm_stats = fused
        elif gnb is not None:
            raise AdmError("conv(gnb=...): the map does not offer fused statistics (needs >= 16x16, pixels % 256 == 0)")
    pkey = (variant, taps, h * w > 64, 4 if fold is not None else a.prologue)   # prologue 4: the folded launches are another kernel symbol (PROX = 4)
    if CONV_PROFILE is not None and (CONV_PROFILE_KEY is None or CONV_PROFILE_KEY == pkey):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.adm_conv(C.byref(a), _stream()), "adm_conv")
        e1.record()
        CONV_PROFILE.append((e0, e1, 2.0 * n * h * w * cout * ((c0 + c1) * taps + a.fc0 + a.fc1), pkey,
                             (n, h, w, c0 + c1, cout)))
        return out
    check(lib.adm_conv(C.byref(a), _stream()), "adm_conv")
    return out


def _conv_up_phases(x0, w_up, bias, cout, aff, silu, out, want_stats):
    """conv3x3(upsample2x(x0)) as the four 2x2-tap phase convs of adm_conv_args.up_phase, all in one launch (up_phase = 5):
    x0 is the half-resolution source [n, h, w, c0], the result [n, 2h, 2w, cout]."""
    n, h, w, c0 = x0.shape
    dev = x0.device
    lib = _L(x0)
    if out is None:
        out = torch.empty((n, 2 * h, 2 * w, cout), dtype=x0.dtype, device=dev)
    a = ConvArgs()
    a.in0, a.bias = _ptr(x0, x0.dtype, "x0"), _ptr(bias, torch.float32, "bias")
    if aff is not None:
        a.aff_a, a.aff_b = _ptr(aff[0], torch.float32, "aff_a"), _ptr(aff[1], torch.float32, "aff_b")
        a.prologue = 2 if silu else 1
    a.out = _ptr(out)
    a.n, a.h, a.w, a.c0, a.c1, a.cout = n, h, w, c0, 0, cout
    a.taps, a.out_mode, a.up_phase = 9, 0, 5   # 5: the four phases in one launch (phase = part of the tile index)
    if not w_up.is_contiguous() or w_up.dim() != 2 or w_up.shape[0] != 4:
        raise AdmError("conv(w_up=...): the [4, elems] tensor of pack_conv_weight_up expected")
    a.w_packed = _ptr(w_up, x0.dtype, "w_up")
    variant = a.variant = lib.adm_conv_pick_variant(C.byref(a))
    if want_stats and USE_FUSED_STATS:
        slabs = lib.adm_conv_stat_slabs(C.byref(a))
        if slabs > 0:
            fused = (torch.empty((n, slabs, cout, 2), dtype=torch.float32, device=dev), slabs)
            a.out_stats = fused[0].data_ptr()
            out._adm_stats = fused
    prof = CONV_PROFILE is not None and (CONV_PROFILE_KEY is None or CONV_PROFILE_KEY == (variant, 4, True, a.prologue))
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib.adm_conv(C.byref(a), _stream()), "adm_conv")
    if prof:
        e1.record()
        # algorithmic work of the layer as the reference states it: a 9-tap conv on the upsampled map
        CONV_PROFILE.append((e0, e1, 2.0 * n * 4 * h * w * cout * c0 * 9, (variant, 4, True, a.prologue), (n, 2 * h, 2 * w, c0, cout)))
    return out


# ------------------------------------------------------------------ attention
def attention(qkv, heads: int, new_order: bool, want_lse: bool = False):
    """qkv bf16 [N, T, 3*H*D] -> bf16 [N, T, H*D] (and the fp32 [N, H, T] log-sum-exp if want_lse)."""
    n, t, c3 = qkv.shape
    c = c3 // 3
    d = c // heads
    out = torch.empty((n, t, c), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((n, heads, t), dtype=torch.float32, device=qkv.device) if want_lse else None
    check(_L(qkv).adm_attention_lse(_ptr(qkv, qkv.dtype, "qkv"), _ptr(out), _ptr(lse), n, t, heads, d,
                                        int(new_order), _stream()), "adm_attention")
    return (out, lse) if want_lse else out


def attention_cross(q, kv, heads: int, d: int, tk: int, scale: float, q_cols: int = None):
    """q bf16 [N, Tq, >= H*D] (head h at columns h*D), kv bf16 [N, rows >= tk, >= 2*H*D] (K heads, then V heads)
    -> bf16 [N, Tq, H*D] = softmax(q k^T * scale) v over the first tk rows of kv.  kv may alias q's storage
    (self-attention over a fused [q | k | v] projection: pass kv = qkv[:, :, H*D:])."""
    n, tq, _ = q.shape
    if q.stride(2) != 1 or kv.stride(2) != 1 or q.stride(0) != tq * q.stride(1) or kv.stride(0) != kv.shape[1] * kv.stride(1):
        raise AdmError("attention_cross: q / kv must be row-major with dense image pitch")
    out = torch.empty((n, tq, heads * d), dtype=q.dtype, device=q.device)
    for t_ in (q, kv):
        if t_.dtype != q.dtype or not t_.is_cuda:
            raise AdmError("attention_cross: q and kv must be device tensors of one 16-bit type")
    check(_L(q).adm_attention_cross(q.data_ptr(), q.stride(1), kv.data_ptr(), kv.stride(1), kv.shape[1],
                                          _ptr(out), n, tq, tk, heads, d, float(scale), _stream()), "adm_attention_cross")
    return out


# ------------------------------------------------------------------ Stable-Diffusion token kernels
def layernorm(x, gamma, beta, eps: float = 1e-5):
    """bf16 [..., C] -> bf16, LayerNorm over the last dimension (fp32 statistics)."""
    c = x.shape[-1]
    out = torch.empty_like(x)
    check(_L(x).adm_layernorm(_ptr(x, x.dtype, "x"), _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"),
                                    _ptr(out), x.numel() // c, c, float(eps), _stream()), "adm_layernorm")
    return out


def geglu_interleave(w, b):
    """GEGLU projection [2*I, C] / [2*I] (rows: values, then gates -- `proj(x).chunk(2, dim=-1)`) -> the row order conv(geglu=True)
    expects: row 2m = value m, row 2m + 1 = gate m.  Pure tensor algebra (testable without a GPU)."""
    inner = w.shape[0] // 2
    wi = torch.stack([w[:inner], w[inner:]], dim=1).reshape(w.shape)
    bi = torch.stack([b[:inner], b[inner:]], dim=1).reshape(b.shape)
    return wi.contiguous(), bi.contiguous()


_GEGLU_OK = {}


def geglu_fusable(x, cout: int) -> bool:
    """Does adm_conv take conv(x, ..., cout, 1, geglu=True)?  The GEGLU epilogue lives on the resident-tile 1x1 kernel only, and
    whether that kernel takes a shape is the library's decision (LDS budget of the pixel tile, its A/B switches): asked once per
    (map, channels, dtype) through adm_conv_pick_variant -- by shape, never by batch."""
    n, h, w, c = x.shape
    key = (h, w, c, cout, x.dtype)
    ok = _GEGLU_OK.get(key)
    if ok is None:
        a = ConvArgs()
        a.n, a.h, a.w, a.c0, a.c1, a.cout, a.taps, a.geglu = 1, h, w, c, 0, cout, 1, 1
        ok = _GEGLU_OK[key] = _L(x).adm_conv_pick_variant(C.byref(a)) == 10
    return ok


def geglu(u):
    """bf16 [..., 2*I] -> bf16 [..., I]: u[..., :I] * gelu(u[..., I:])."""
    inner = u.shape[-1] // 2
    out = torch.empty(u.shape[:-1] + (inner,), dtype=u.dtype, device=u.device)
    check(_L(u).adm_geglu(_ptr(u, u.dtype, "u"), _ptr(out), u.numel() // (2 * inner), inner, _stream()), "adm_geglu")
    return out


# ------------------------------------------------------------------ backward-data (classifier guidance)
def attention_bwd(qkv, out, dout, lse, heads: int, new_order: bool):
    n, t, c3 = qkv.shape
    d = c3 // 3 // heads
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((n, heads, t), dtype=torch.float32, device=qkv.device)
    check(_L(qkv).adm_attention_bwd(_ptr(qkv, qkv.dtype, "qkv"), _ptr(out, qkv.dtype, "out"), _ptr(dout, qkv.dtype, "dout"),
                                        _ptr(lse, torch.float32, "lse"), _ptr(delta), _ptr(dqkv), n, t, heads, d,
                                        int(new_order), _stream()), "adm_attention_bwd")
    return dqkv


def gn_bwd(x, dy, aff, stats, silu: bool, dy_half=False, add=None, add_half=False, partial=None, norm_add=None):
    """Backward of y = act(a*x+b) (GroupNorm(+FiLM)(+SiLU)): returns dx bf16 NHWC (+ add).

    norm_add: fp32 [N, C] e of a layer that normalised x + e[:, :, None, None] (gn_affine(add=e, want_stats=True)).

    partial: the producing backward-data conv already wrote dz = dy * SiLU'(a x + b) and its (sum dz, sum dz x) slabs
    (conv(..., gnb=(x, aff)): adm_conv_args.prologue == 3) -- `dy` is then that dz and the partial pass is skipped."""
    n, h, w, c = x.shape
    hw = h * w
    lib = _L(x)
    k1 = torch.empty((n, c), dtype=torch.float32, device=x.device)
    k0 = torch.empty((n, c), dtype=torch.float32, device=x.device)
    a, b = aff
    if partial is None:
        slabs = max(1, hw // GN_BWD_SLAB_PIXELS)
        partial = torch.empty((n, slabs, c, 2), dtype=torch.float32, device=x.device)
        check(lib.adm_gn_bwd_partial(_ptr(x, x.dtype, "x"), _ptr(dy, x.dtype, "dy"), _ptr(a, torch.float32), _ptr(b, torch.float32),
                                     _ptr(partial), n, h, w, c, slabs, int(silu), int(dy_half), _stream()),
              "adm_gn_bwd_partial")
    else:
        if dy_half or tuple(partial.shape[0:1] + partial.shape[2:]) != (n, c, 2):
            raise AdmError("gn_bwd(partial=...): same-resolution dz and [n, slabs, c, 2] sums expected")
        slabs = partial.shape[1]
        silu = False   # the SiLU derivative is already in dz
    if norm_add is not None and (norm_add.dtype != torch.float32 or norm_add.shape != (n, c) or norm_add.stride(1) != 1):
        raise AdmError("norm_add must be float32 [N, C] with unit channel stride")
    check(lib.adm_gn_bwd_finalize(_ptr(partial), _ptr(a), _ptr(stats, torch.float32, "stats"),
                                  None if norm_add is None else norm_add.data_ptr(), 0 if norm_add is None else norm_add.stride(0),
                                  _ptr(k1), _ptr(k0), n, c, hw, slabs, _stream()), "adm_gn_bwd_finalize")
    out = torch.empty_like(x)
    check(lib.adm_gn_bwd_apply(_ptr(x), _ptr(dy, x.dtype, "dy"), _ptr(a), _ptr(b), _ptr(k1), _ptr(k0), _ptr(add, x.dtype, "add"),
                               _ptr(out), n, h, w, c, int(silu), int(dy_half), int(add_half), _stream()),
          "adm_gn_bwd_apply")
    return out


def grad_add(a, b, b_half=False):
    n, h, w, c = a.shape
    out = torch.empty_like(a)
    check(_L(a).adm_grad_add(_ptr(a, a.dtype, "a"), _ptr(b, a.dtype, "b"), _ptr(out), n, h, w, c, int(b_half),
                                   _stream()), "adm_grad_add")
    return out


def logsoftmax_grad(logits, y, scale: float):
    n, k = logits.shape
    dl = torch.empty_like(logits)
    check(_lib.load().adm_logsoftmax_grad(_ptr(logits, torch.float32, "logits"), _ptr(y, torch.int64, "y"), _ptr(dl),
                                          None, n, k, float(scale), _stream()), "adm_logsoftmax_grad")
    return dl


def pool_prep(h, aff, pos, tpad: int):
    n, hh, ww, c = h.shape
    tok = torch.empty((n, tpad, c), dtype=h.dtype, device=h.device)
    check(_L(h).adm_pool_prep(_ptr(h, h.dtype, "h"), _ptr(aff[0], torch.float32), _ptr(aff[1], torch.float32),
                                    _ptr(pos, torch.float32, "pos"), _ptr(tok), n, hh * ww, c, tpad, _stream()),
          "adm_pool_prep")
    return tok


def pool_attn_fwd(qkv, t: int, heads: int):
    n, tpad, c3 = qkv.shape
    c = c3 // 3
    a0 = torch.empty((n, c), dtype=torch.float32, device=qkv.device)
    wts = torch.empty((n, heads, tpad), dtype=torch.float32, device=qkv.device)
    check(_L(qkv).adm_pool_attn_fwd(_ptr(qkv, qkv.dtype, "qkv"), _ptr(a0), _ptr(wts), n, t, tpad, heads, c // heads,
                                        _stream()), "adm_pool_attn_fwd")
    return a0, wts


def pool_attn_bwd(qkv, wts, da0, t: int, heads: int):
    n, tpad, c3 = qkv.shape
    c = c3 // 3
    dqkv = torch.empty_like(qkv)
    check(_L(qkv).adm_pool_attn_bwd(_ptr(qkv, qkv.dtype, "qkv"), _ptr(wts, torch.float32), _ptr(da0, torch.float32),
                                        _ptr(dqkv), n, t, tpad, heads, c // heads, _stream()), "adm_pool_attn_bwd")
    return dqkv


def pool_prep_bwd(dtok, hh: int, ww: int):
    n, tpad, c = dtok.shape
    dact = torch.empty((n, hh, ww, c), dtype=dtok.dtype, device=dtok.device)
    check(_L(dtok).adm_pool_prep_bwd(_ptr(dtok, dtok.dtype, "dtok"), _ptr(dact), n, hh * ww, c, tpad, _stream()),
          "adm_pool_prep_bwd")
    return dact


# ------------------------------------------------------------------ the classifier's other heads (csrc/adm_clfhead.hip)
def channel_mean(h, aff=None, out=None, col: int = 0):
    """mean over the pixels of SiLU(a*h + b) (aff given) or of h itself -> fp32 [N, C], or into out[:, col:col + C]."""
    n, hh, ww, c = h.shape
    if out is None:
        out, col = torch.empty((n, c), dtype=torch.float32, device=h.device), 0
    if out.dtype != torch.float32 or out.stride(1) != 1 or col + c > out.shape[1]:
        raise AdmError("channel_mean: out must be float32 [N, >= col + C] with unit column stride")
    a, b = aff if aff is not None else (None, None)
    check(_L(h).adm_channel_mean(_ptr(h, h.dtype, "h"), _ptr(a, torch.float32), _ptr(b, torch.float32),
                                 out.data_ptr() + 4 * col, out.stride(0), n, hh * ww, c, _stream()), "adm_channel_mean")
    return out


def bcast_add(v, shape, dtype, scale: float, add=None, col: int = 0):
    """16-bit NHWC [n, h, w, c] = (add or 0) + v[:, col:col + c, None, None] * scale (the backward of a pixel mean)."""
    n, hh, ww, c = shape
    if v.dtype != torch.float32 or v.stride(1) != 1 or v.shape[0] != n or col + c > v.shape[1]:
        raise AdmError("bcast_add: v must be float32 [N, >= col + C] with unit column stride")
    out = torch.empty(shape, dtype=dtype, device=v.device)
    check(_L(out).adm_bcast_add(v.data_ptr() + 4 * col, v.stride(0), float(scale), _ptr(add, dtype, "add"), _ptr(out), n, hh * ww, c,
                                _stream()), "adm_bcast_add")
    return out


def vec_act(x, mode: str, dy=None):
    """fp32 vectors: act(x), or dy * act'(x) with dy; mode 'silu' | 'relu'."""
    out = torch.empty_like(x)
    check(_lib.load().adm_vec_act(_ptr(x, torch.float32, "x"), _ptr(dy, torch.float32, "dy"), _ptr(out), x.numel(),
                                  {"silu": 1, "relu": 2}[mode], _stream()), "adm_vec_act")
    return out


def vec_gn(x, gamma, beta, eps: float = GN_EPS):
    """GroupNorm32(32, C) of fp32 [N, C] rows -> (y, stats [N, 32, 2])."""
    n, c = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((n, 32, 2), dtype=torch.float32, device=x.device)
    check(_lib.load().adm_vec_gn(_ptr(x, torch.float32, "x"), _ptr(gamma, torch.float32, "gamma"), _ptr(beta, torch.float32, "beta"),
                                 _ptr(y), _ptr(stats), n, c, float(eps), _stream()), "adm_vec_gn")
    return y, stats


def vec_gn_bwd(x, gamma, stats, dz):
    n, c = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().adm_vec_gn_bwd(_ptr(x, torch.float32, "x"), _ptr(gamma, torch.float32, "gamma"), _ptr(stats, torch.float32, "stats"),
                                     _ptr(dz, torch.float32, "dz"), _ptr(dx), n, c, _stream()), "adm_vec_gn_bwd")
    return dx


def fold_ok(h: int, w: int) -> bool:
    """Maps on which adm_conv takes fold0 (the 8-wave tiles: 8x8, or >= 16x16 in whole 256-pixel tiles)."""
    return (h == 8 and w == 8) or (h >= 16 and w >= 16 and (h * w) % 256 == 0)


def fold_weights(w3_packed, w1_packed):
    """Packed weights of conv(fold=...): the skip_connection's 1x1 K-steps follow the 3x3 conv's (same Cout tiling)."""
    return torch.cat([w3_packed.reshape(-1), w1_packed.reshape(-1)]).contiguous()


def pack_conv_weight_bwd(w, dtype=BF16):
    """Backward-data image of a conv weight [cout, cin, ...]: conv with cin' = cout, cout' = cin, flipped taps."""
    cout, cin = w.shape[0], w.shape[1]
    taps = 1
    for s in w.shape[2:]:
        taps *= s
    lib = _lib.load("f16" if dtype == F16 else "bf16")
    elems = lib.adm_packed_weight_elems(cin, cout, taps)
    if elems < 0:
        raise AdmError(f"pack_conv_weight_bwd: unsupported weight shape {tuple(w.shape)} (cout % 32 == 0, taps 1|9)")
    w32 = w.detach().to(torch.float32).contiguous()
    out = torch.empty((elems,), dtype=dtype, device=w.device)
    check(lib.adm_pack_conv_weight_bwd(_ptr(w32), _ptr(out), cout, cin, taps, _stream()), "adm_pack_conv_weight_bwd")
    return out


# ------------------------------------------------------------------ Inception layers (csrc/adm_convg.hip)
def _nhwc_view(t: torch.Tensor, name: str):
    """(data pointer, channel stride) of a 4-D NHWC tensor or of a CHANNEL SLICE of one (t[..., a:b]): pixels dense,
    channels contiguous."""
    if t.dim() != 4 or not t.is_cuda:
        raise AdmError(f"{name}: expected a 4-D NHWC device tensor")
    n, h, w, c = t.shape
    cs = t.stride(2)
    if t.stride(3) != 1 or t.stride(1) != w * cs or t.stride(0) != h * w * cs or cs < c:
        raise AdmError(f"{name}: not an NHWC tensor / channel slice (strides {t.stride()})")
    if t.device.index != torch.cuda.current_device():
        raise AdmError(f"{name}: tensor lives on {t.device} but the current device is cuda:{torch.cuda.current_device()}")
    return t.data_ptr(), cs


def pack_conv2d_weight(w, scale=None, dtype=F16):
    """fp32 [cout, cin, kh, kw] (x scale[cout]: a folded BatchNorm) -> the [cout][kh*kw][cin_pad] layout of adm_conv2d."""
    cout, cin, kh, kw = w.shape
    cin_pad = (cin + 31) // 32 * 32
    lib = _lib.load("f16" if dtype == F16 else "bf16")
    w32 = w.detach().to(torch.float32).contiguous()
    s32 = None if scale is None else scale.detach().to(torch.float32).contiguous()
    out = torch.empty((cout, kh * kw, cin_pad), dtype=dtype, device=w.device)
    check(lib.adm_pack_conv2d_weight(_ptr(w32), _ptr(s32), _ptr(out), cout, cin, kh, kw, cin_pad, _stream()),
          "adm_pack_conv2d_weight")
    return out


def conv2d(x, w_packed, bias, kh: int, kw: int, stride: int = 1, pad=(0, 0), relu: bool = True, out=None):
    """General NHWC convolution on the matrix cores.  x: [N,H,W,Cs] (or a channel slice) whose first cin_pad =
    w_packed.shape[2] channels are read; out: [N,OH,OW,cout] tensor or channel slice of a concatenated tensor."""
    n, h, w, _ = x.shape
    cout, taps, cin_pad = w_packed.shape
    if taps != kh * kw:
        raise AdmError(f"conv2d: packed weight has {taps} taps, kernel is {kh}x{kw}")
    if x.shape[3] < cin_pad:
        raise AdmError(f"conv2d: the input holds {x.shape[3]} channels, the packed weight reads {cin_pad} (pad the tensor with zeros)")
    oh, ow = (h + 2 * pad[0] - kh) // stride + 1, (w + 2 * pad[1] - kw) // stride + 1
    if out is None:
        out = torch.empty((n, oh, ow, cout), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (n, oh, ow, cout) or out.dtype != x.dtype or w_packed.dtype != x.dtype:
        raise AdmError(f"conv2d: output {tuple(out.shape)} / dtypes do not match ({n}, {oh}, {ow}, {cout}) {x.dtype}")
    a = _lib.Conv2dArgs()
    a.in_, a.in_stride = _nhwc_view(x, "x")
    a.out, a.out_stride = _nhwc_view(out, "out")
    a.w, a.bias = _ptr(w_packed, x.dtype, "w_packed"), _ptr(bias, torch.float32, "bias")
    a.n, a.h, a.w_in, a.cin_pad, a.cout = n, h, w, cin_pad, cout
    a.kh, a.kw, a.stride, a.pad_h, a.pad_w, a.relu = kh, kw, stride, pad[0], pad[1], int(relu)
    check(_L(x).adm_conv2d(C.byref(a), _stream()), "adm_conv2d")
    return out


def pool2d(x, k: int, stride: int, pad: int, mode: str, out=None):
    """mode "max" (F.max_pool2d) or "avg" (F.avg_pool2d(count_include_pad=False)) over NHWC; out may be a channel slice."""
    n, h, w, c = x.shape
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((n, oh, ow, c), dtype=x.dtype, device=x.device)
    if tuple(out.shape) != (n, oh, ow, c) or out.dtype != x.dtype:
        raise AdmError(f"pool2d: output {tuple(out.shape)} does not match ({n}, {oh}, {ow}, {c})")
    pi, si = _nhwc_view(x, "x")
    po, so = _nhwc_view(out, "out")
    check(_L(x).adm_pool2d(pi, po, n, h, w, c, si, so, k, stride, pad, {"max": 0, "avg": 1}[mode], _stream()), "adm_pool2d")
    return out


def global_avgpool_f32(x):
    n, h, w, c = x.shape
    out = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(_L(x).adm_global_avgpool_f32(_ptr(x, x.dtype, "x"), _ptr(out), n, h * w, c, _stream()), "adm_global_avgpool_f32")
    return out


def resize_bilinear(images, oh: int, ow: int, cpad: int, layout: str, half_pixel: bool, scale: float, shift: float, dtype=F16):
    """3-channel images -> [N, oh, ow, cpad] 16-bit NHWC (channels 3.. zero), value * scale + shift.
    layout "u8_nhwc" | "f32_nchw" | "f32_nhwc"."""
    kind = {"u8_nhwc": 0, "f32_nchw": 1, "f32_nhwc": 2}[layout]
    want = torch.uint8 if kind == 0 else torch.float32
    if images.dim() != 4 or images.shape[1 if kind == 1 else 3] != 3:
        raise AdmError(f"resize_bilinear: expected 3-channel {layout} images, got {tuple(images.shape)}")
    n = images.shape[0]
    h, w = (images.shape[2], images.shape[3]) if kind == 1 else (images.shape[1], images.shape[2])
    out = torch.empty((n, oh, ow, cpad), dtype=dtype, device=images.device)
    lib = _lib.load("f16" if dtype == F16 else "bf16")
    check(lib.adm_resize_bilinear(_ptr(images, want, "images"), _ptr(out), n, h, w, oh, ow, cpad, kind, int(half_pixel),
                                  float(scale), float(shift), _stream()), "adm_resize_bilinear")
    return out
