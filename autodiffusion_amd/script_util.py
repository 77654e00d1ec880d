"""Factories with the reference's signatures (drop-in lower boundary).

Mirrors reference guided_diffusion/script_util.py:
``model_and_diffusion_defaults`` (:43-66), ``classifier_defaults`` (:27-40),
``create_model_and_diffusion`` (:75-130), ``create_model`` (:133-211),
``create_classifier`` (:257-295), ``create_gaussian_diffusion`` (:415-453) and the
argparse helpers (:456-481).  The returned model is the HIP engine
(``autodiffusion_amd.unet``), the diffusion is ``autodiffusion_amd.sampler.SpacedDiffusion``.
Super-resolution factories are out of scope (no search script uses them).
"""
from __future__ import annotations

import argparse

from . import schedule as gd
from .arch import build_unet_plan
from .sampler import SpacedDiffusion
from .schedule import space_timesteps

NUM_CLASSES = 1000


def diffusion_defaults():
    return dict(
        learn_sigma=False, diffusion_steps=1000, noise_schedule="linear", timestep_respacing="",
        use_kl=False, predict_xstart=False, rescale_timesteps=False, rescale_learned_sigmas=False,
    )


def classifier_defaults():
    return dict(
        image_size=64, classifier_use_fp16=False, classifier_width=128, classifier_depth=2,
        classifier_attention_resolutions="32,16,8", classifier_use_scale_shift_norm=True,
        classifier_resblock_updown=True, classifier_pool="attention",
    )


def model_and_diffusion_defaults():
    res = dict(
        image_size=64, num_channels=128, num_res_blocks=2, num_heads=4, num_heads_upsample=-1,
        num_head_channels=-1, attention_resolutions="16,8", channel_mult="", dropout=0.0,
        class_cond=False, use_checkpoint=False, use_scale_shift_norm=True, resblock_updown=False,
        use_fp16=False, use_new_attention_order=False, use_dynamic_unet=False,
    )
    res.update(diffusion_defaults())
    return res


def classifier_and_diffusion_defaults():
    res = classifier_defaults()
    res.update(diffusion_defaults())
    return res


def _default_channel_mult(image_size):
    table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4),
             64: (1, 2, 3, 4), 32: (1, 2, 2, 2)}
    if image_size not in table:
        raise ValueError(f"unsupported image size: {image_size}")
    return table[image_size]


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False,
                 class_cond=False, use_checkpoint=False, attention_resolutions="16", num_heads=1,
                 num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0,
                 resblock_updown=False, use_fp16=False, use_new_attention_order=False,
                 use_dynamic_unet=False):
    from .unet import UNetModel
    if channel_mult == "":
        channel_mult = _default_channel_mult(image_size)
    else:
        channel_mult = tuple(int(m) for m in channel_mult.split(","))
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    plan = build_unet_plan(
        image_size=image_size, in_channels=3, model_channels=num_channels,
        out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
        attention_resolutions=attention_ds, channel_mult=channel_mult,
        num_classes=(NUM_CLASSES if class_cond else None), num_heads=num_heads,
        num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, resblock_updown=resblock_updown,
        use_new_attention_order=use_new_attention_order, dynamic=use_dynamic_unet)
    # dropout / use_checkpoint only matter for training; accepted and ignored at inference
    if not use_fp16:
        from .unet import warn_compute_dtype
        warn_compute_dtype("create_model", "use_fp16")
    return UNetModel(plan, use_fp16=use_fp16)


def create_gaussian_diffusion(*, steps=1000, learn_sigma=False, sigma_small=False, noise_schedule="linear",
                              use_kl=False, predict_xstart=False, rescale_timesteps=False,
                              rescale_learned_sigmas=False, timestep_respacing=""):
    betas = gd.get_named_beta_schedule(noise_schedule, steps)
    if use_kl:
        loss_type = gd.LossType.RESCALED_KL
    elif rescale_learned_sigmas:
        loss_type = gd.LossType.RESCALED_MSE
    else:
        loss_type = gd.LossType.MSE
    if not timestep_respacing:
        timestep_respacing = [steps]
    if learn_sigma:
        var_type = gd.ModelVarType.LEARNED_RANGE
    else:
        var_type = gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
        model_mean_type=(gd.ModelMeanType.START_X if predict_xstart else gd.ModelMeanType.EPSILON),
        model_var_type=var_type, loss_type=loss_type, rescale_timesteps=rescale_timesteps)


def create_model_and_diffusion(image_size, class_cond, learn_sigma, num_channels, num_res_blocks,
                               channel_mult, num_heads, num_head_channels, num_heads_upsample,
                               attention_resolutions, dropout, diffusion_steps, noise_schedule,
                               timestep_respacing, use_kl, predict_xstart, rescale_timesteps,
                               rescale_learned_sigmas, use_checkpoint, use_scale_shift_norm,
                               resblock_updown, use_fp16, use_new_attention_order, use_dynamic_unet=False):
    model = create_model(
        image_size, num_channels, num_res_blocks, channel_mult=channel_mult, learn_sigma=learn_sigma,
        class_cond=class_cond, use_checkpoint=use_checkpoint, attention_resolutions=attention_resolutions,
        num_heads=num_heads, num_head_channels=num_head_channels, num_heads_upsample=num_heads_upsample,
        use_scale_shift_norm=use_scale_shift_norm, dropout=dropout, resblock_updown=resblock_updown,
        use_fp16=use_fp16, use_new_attention_order=use_new_attention_order,
        use_dynamic_unet=use_dynamic_unet)
    diffusion = create_gaussian_diffusion(
        steps=diffusion_steps, learn_sigma=learn_sigma, noise_schedule=noise_schedule, use_kl=use_kl,
        predict_xstart=predict_xstart, rescale_timesteps=rescale_timesteps,
        rescale_learned_sigmas=rescale_learned_sigmas, timestep_respacing=timestep_respacing)
    return model, diffusion


def create_classifier(image_size, classifier_use_fp16, classifier_width, classifier_depth,
                      classifier_attention_resolutions, classifier_use_scale_shift_norm,
                      classifier_resblock_updown, classifier_pool):
    from .classifier import EncoderUNetModel
    if image_size not in (64, 128, 256, 512):
        raise ValueError(f"unsupported image size: {image_size}")
    attention_ds = tuple(image_size // int(res) for res in classifier_attention_resolutions.split(","))
    plan = build_unet_plan(
        image_size=image_size, in_channels=3, model_channels=classifier_width, out_channels=1000,
        num_res_blocks=classifier_depth, attention_resolutions=attention_ds,
        channel_mult=_default_channel_mult(image_size), num_head_channels=64,
        use_scale_shift_norm=classifier_use_scale_shift_norm, resblock_updown=classifier_resblock_updown,
        encoder_only=True, pool=classifier_pool)
    if not classifier_use_fp16:  # the reference's default: every classifier runs fp32 (script_util.py:33)
        from .unet import warn_compute_dtype
        warn_compute_dtype("create_classifier", "classifier_use_fp16")
    return EncoderUNetModel(plan, use_fp16=classifier_use_fp16)


def add_dict_to_argparser(parser, default_dict):
    for k, v in default_dict.items():
        v_type = type(v)
        if v is None:
            v_type = str
        elif isinstance(v, bool):
            v_type = str2bool
        parser.add_argument(f"--{k}", default=v, type=v_type)


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("boolean value expected")
