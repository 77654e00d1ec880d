"""Shared test helpers: golden loading, tiny model configs, filled parameters."""
import os

import numpy as np

from autodiffusion_amd.arch import build_unet_plan
from oracle.fill import fill_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def plan_m32(dynamic=True, legacy=False, class_cond=True):
    """Mirrors cfg_m32 in tests/golden/capture_golden.py."""
    return build_unet_plan(
        image_size=32, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
        attention_resolutions=(2, 4), channel_mult=(1, 2, 2),
        num_classes=1000 if class_cond else None,
        num_heads=2 if legacy else 4, num_head_channels=-1 if legacy else 32,
        use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=not legacy,
        dynamic=dynamic)


def plan_m64(dynamic=False):
    """Mirrors cfg_m64."""
    return build_unet_plan(
        image_size=64, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
        attention_resolutions=(4,), channel_mult=(1, 2, 2), num_classes=1000,
        num_heads=4, num_head_channels=32, use_scale_shift_norm=True, resblock_updown=True,
        use_new_attention_order=True, dynamic=dynamic)


def plan_c64():
    """Mirrors cfg_c64 (create_classifier: width 64, depth 1, attention pool)."""
    return build_unet_plan(
        image_size=64, in_channels=3, model_channels=64, out_channels=1000, num_res_blocks=1,
        attention_resolutions=(2, 4, 8), channel_mult=(1, 2, 3, 4), num_classes=None,
        num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True,
        encoder_only=True, pool="attention")


def filled(plan, prefix=""):
    shapes = {prefix + k: v for k, v in plan.param_shapes().items()}
    sd = fill_state_dict(shapes)
    return {k[len(prefix):]: v for k, v in sd.items()}


def denoised_fn_fixture(x):
    """The `denoised_fn` of tests/golden/capture_denoised.py (a fixed, smooth, non-linear map of the predicted x_0)."""
    import torch
    return 0.8 * x + 0.25 * torch.tanh(3.0 * x)
