"""ctypes binding of libadm_hip.so (see include/adm_hip.h).

The HIP extension IS the product: if the shared library is missing or an entry
point is absent this module raises -- there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ADM_HIP_LIB") or os.path.join(_HERE, "libadm_hip.so")  # env override: A/B builds
# the same kernels built with IEEE half as the 16-bit element type (csrc/adm_common.h, -DADM_ACT_F16): the reference's own
# torso precision (use_fp16=True); selected per tensor dtype by ops.py, per model by `torso="fp16"` / ADM_TORSO=fp16
LIB_PATH_F16 = os.environ.get("ADM_HIP_LIB_F16") or os.path.join(_HERE, "libadm_hip_f16.so")
ABI_VERSION = 9


class AdmError(RuntimeError):
    pass


class StepCoefs(C.Structure):
    """struct adm_step_coefs (include/adm_hip.h)."""
    _fields_ = [
        ("sqrt_recip_ac", C.c_float), ("sqrt_recipm1_ac", C.c_float), ("ac", C.c_float),
        ("ac_prev", C.c_float), ("coef1", C.c_float), ("coef2", C.c_float),
        ("log_var_lo", C.c_float), ("log_var_hi", C.c_float), ("fixed_var", C.c_float),
        ("eta", C.c_float), ("nonzero", C.c_int32), ("learned_range", C.c_int32),
        ("predict_xstart", C.c_int32), ("clip_denoised", C.c_int32),
    ]


class Conv2dArgs(C.Structure):
    """struct adm_conv2d_args (include/adm_hip.h)."""
    _fields_ = [("in_", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p)] + [
        (k, C.c_int32) for k in ("n", "h", "w_in", "cin_pad", "in_stride", "cout", "out_stride", "kh", "kw", "stride",
                                 "pad_h", "pad_w", "relu")]


class ConvArgs(C.Structure):
    """struct adm_conv_args (include/adm_hip.h)."""
    _fields_ = [
        ("in0", C.c_void_p), ("in1", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
        ("aff_a", C.c_void_p), ("aff_b", C.c_void_p), ("res", C.c_void_p), ("out", C.c_void_p),
        ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c0", C.c_int32), ("c1", C.c_int32),
        ("cout", C.c_int32), ("taps", C.c_int32), ("prologue", C.c_int32), ("out_mode", C.c_int32),
        ("variant", C.c_int32), ("out_stats", C.c_void_p), ("w_packed32", C.c_void_p),
        ("in_up", C.c_int32), ("res_up", C.c_int32), ("ksplit", C.c_int32), ("ws", C.c_void_p),
        ("up_phase", C.c_int32), ("geglu", C.c_int32),
        ("fold0", C.c_void_p), ("fold1", C.c_void_p), ("fc0", C.c_int32), ("fc1", C.c_int32),
        ("out_scale", C.c_float),
    ]


class SdStepCoefs(C.Structure):
    """struct adm_sd_step_coefs (include/adm_hip.h)."""
    _fields_ = [("cfg_scale", C.c_float), ("w", C.c_float * 4), ("sqrt_one_minus_at", C.c_float), ("sqrt_at", C.c_float),
                ("sqrt_a_prev", C.c_float), ("dir_coef", C.c_float), ("sigma", C.c_float)]


_P, _I, _F = C.c_void_p, C.c_int, C.c_float

# name -> (restype, argtypes); every symbol declared in include/adm_hip.h
SIGNATURES = {
    "adm_abi_version": (_I, []),
    "adm_last_error": (C.c_char_p, []),
    "adm_stream_create_cumask": (_I, [_P, _I, C.POINTER(C.c_void_p)]),
    "adm_stream_set_cus": (_I, [_P, _I]),
    "adm_stream_destroy": (_I, [_P]),
    "adm_stream_probe": (_I, [_P, _I, _P]),
    "adm_ddim_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(StepCoefs), _P]),
    "adm_ddpm_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, C.POINTER(StepCoefs), _P]),
    "adm_pack_u8_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "adm_timestep_embedding": (_I, [_P, _P, _I, _I, _F, _P]),
    "adm_linear_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "adm_stem_conv3x3": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_nchw_to_nhwc_pad": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_gn_partial": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "adm_gn_finalize": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "adm_resample": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_conv": (_I, [C.POINTER(ConvArgs), _P]),
    "adm_conv_stat_slabs": (_I, [C.POINTER(ConvArgs)]),
    "adm_conv_pick_variant": (_I, [C.POINTER(ConvArgs)]),
    "adm_gn_finalize2": (_I, [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _F, _P]),
    "adm_packed_weight_elems": (C.c_int64, [_I, _I, _I]),
    "adm_pack_conv_weight": (_I, [_P, _P, _I, _I, _I, _P]),
    "adm_packed_weight32_elems": (C.c_int64, [_I, _I, _I]),
    "adm_pack_conv_weight32": (_I, [_P, _P, _I, _I, _I, _P]),
    "adm_attention": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_attention_lse": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_attention_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_gn_bwd_partial": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "adm_gn_bwd_finalize": (_I, [_P, _P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _P]),
    "adm_gn_bwd_apply": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "adm_grad_add": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_logsoftmax_grad": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "adm_pool_prep": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "adm_pool_attn_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_pool_attn_bwd": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_pool_prep_bwd": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "adm_pack_conv_weight_bwd": (_I, [_P, _P, _I, _I, _I, _P]),
    "adm_attention_cross": (_I, [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _P]),
    "adm_layernorm": (_I, [_P, _P, _P, _P, C.c_int64, _I, _F, _P]),
    "adm_geglu": (_I, [_P, _P, C.c_int64, _I, _P]),
    "adm_gn_finalize_add": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _F, _P]),
    "adm_sd_step": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(SdStepCoefs), _P]),
    "adm_dpm_step": (_I, [_P, _P, _P, _P, _P, _P, C.c_int64, _F, _F, _F, _F, _F, _F, _P]),
    "adm_fid_accumulate": (_I, [_P, _P, _P, _I, _I, _P]),
    "adm_conv2d": (_I, [C.POINTER(Conv2dArgs), _P]),
    "adm_pack_conv2d_weight": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "adm_pool2d": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "adm_global_avgpool_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "adm_resize_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _P]),
    "adm_channel_mean": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "adm_bcast_add": (_I, [_P, _I, _F, _P, _P, _I, _I, _I, _P]),
    "adm_vec_act": (_I, [_P, _P, _P, C.c_int64, _I, _P]),
    "adm_vec_gn": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "adm_vec_gn_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
}

_libs = {}


def load(kind: str = "bf16") -> C.CDLL:
    """Load (once) and type the shared library of the given element type; raise AdmError if it is unusable."""
    if kind in _libs:
        return _libs[kind]
    LIB_PATH = {"bf16": globals()["LIB_PATH"], "f16": LIB_PATH_F16}[kind]
    # PyTorch-ROCm ships its own libamdhip64.so.7; import it FIRST so that the dynamic loader binds
    # our library to the same HIP runtime instance (shared streams / device pointers).  Loading ours
    # first would bring in /opt/rocm's copy and leave torch without a usable device.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AdmError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C autodiffusion_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU/PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise AdmError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    v = lib.adm_abi_version()
    if v != ABI_VERSION:
        raise AdmError(f"{os.path.basename(LIB_PATH)} ABI {v} != expected {ABI_VERSION}; rebuild")
    _libs[kind] = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().adm_last_error().decode("utf-8", "replace")
        raise AdmError(f"{what} failed (status {status}): {msg}")
