"""CPU tests of the product's host logic: schedule tables, factories, argparse helpers, state dicts."""
import argparse
import copy

import numpy as np
import pytest
import torch

from autodiffusion_amd import schedule as S
from autodiffusion_amd.sampler import SpacedDiffusion, step_coefs
from autodiffusion_amd.script_util import (add_dict_to_argparser, args_to_dict, classifier_defaults,
                                           create_classifier, create_gaussian_diffusion,
                                           create_model_and_diffusion, model_and_diffusion_defaults, str2bool)
from helpers import golden

NAMES = ("betas",) + S.TABLES


def test_betas_and_space_timesteps_golden():
    g = golden("schedules")
    assert np.array_equal(S.get_named_beta_schedule("cosine", 1000), g["betas_cosine"])
    assert np.array_equal(S.get_named_beta_schedule("linear", 1000), g["betas_linear"])
    for key in ("ddim4", "ddim10", "4", "10,15", "25"):
        assert sorted(S.space_timesteps(1000, key)) == g["space_" + key.replace(",", "_")].tolist()
    with pytest.raises(ValueError):
        S.space_timesteps(1000, "ddim999")
    with pytest.raises(NotImplementedError):
        S.get_named_beta_schedule("sqrt", 10)


def test_spaced_constructor_tables_golden():
    g = golden("spaced_ddim4_cosine")
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="cosine", timestep_respacing="ddim4")
    assert d.timestep_map == g["timestep_map"].tolist() and d.original_num_steps == 1000
    for n in NAMES:
        assert np.array_equal(getattr(d, n), g[n]), n


@pytest.mark.parametrize("sched", ["cosine", "linear"])
def test_apply_candidate_is_reset_diffusion_bit_exact(sched):
    g = golden(f"reset_diffusion_{sched}")
    base = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule=sched)
    active = copy.deepcopy(base)
    for tag in ("k4", "k6", "k10", "k1"):
        S.apply_candidate(active, base, g[f"{tag}_cand"].tolist())
        assert active.timestep_map == g[f"{tag}_timestep_map"].tolist()
        assert active.num_timesteps == int(g[f"{tag}_num_timesteps"])
        assert active.use_timesteps == set(g[f"{tag}_cand"].tolist())
        for n in NAMES:
            assert np.array_equal(getattr(active, n), g[f"{tag}_{n}"]), (tag, n)
    assert base.num_timesteps == 1000  # the base process is never touched


def test_step_coefs_follow_external_table_mutation():
    d = create_gaussian_diffusion(steps=1000, learn_sigma=True, noise_schedule="linear", timestep_respacing="ddim4")
    c = d._coefs(2, True)
    assert c.learned_range == 1 and c.nonzero == 1 and c.clip_denoised == 1
    assert c.ac == np.float32(d.alphas_cumprod[2]) and c.log_var_hi == np.float32(np.log(d.betas)[2])
    d.alphas_cumprod = d.alphas_cumprod * 0.5  # the search scripts overwrite attributes in place
    assert d._coefs(2, True).ac == np.float32(d.alphas_cumprod[2])
    assert d._coefs(0, False).nonzero == 0
    d2 = create_gaussian_diffusion(steps=100, learn_sigma=False, noise_schedule="linear")
    c = d2._coefs(5, True)
    assert c.learned_range == 0 and c.fixed_var == np.float32(d2.betas[5])
    c0 = d2._coefs(0, True)
    assert c0.fixed_var == np.float32(d2.posterior_variance[1])  # FIXED_LARGE: first entry replaced


def test_single_step_respacing_fails_like_the_reference():
    with pytest.raises(IndexError):
        create_gaussian_diffusion(steps=1000, learn_sigma=True, timestep_respacing="1")


def test_factory_signatures_defaults_and_state_dict_layout():
    d = model_and_diffusion_defaults()
    assert list(d)[:3] == ["image_size", "num_channels", "num_res_blocks"] and d["use_dynamic_unet"] is False
    d.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2", attention_resolutions="32",
             num_head_channels=32, class_cond=True, learn_sigma=True, resblock_updown=True, use_dynamic_unet=True)
    model, diffusion = create_model_and_diffusion(**d)
    assert isinstance(diffusion, SpacedDiffusion) and model.num_classes == 1000 and model.layer_num == 14
    sd = model.state_dict()
    assert sd["input_blocks.0.0.weight"].shape == (32, 3, 3, 3)
    assert sd["label_emb.weight"].shape == (1000, 128) and sd["out.2.weight"].shape == (6, 32, 3, 3)
    assert float(sd["out.2.weight"].abs().max()) == 0.0  # zero_module
    sd2 = {k: torch.full_like(v, 0.5) for k, v in sd.items()}
    model.load_state_dict(sd2)
    assert float(model.state_dict()["time_embed.0.bias"][0]) == 0.5
    with pytest.raises(RuntimeError):
        model.load_state_dict({"nope": torch.zeros(1)})
    assert model.convert_to_fp16().dtype == torch.float16 and len(list(model.parameters())) == len(sd)
    with pytest.raises(ValueError):
        create_classifier(**{**classifier_defaults(), "image_size": 32})
    clf = create_classifier(**{**classifier_defaults(), "classifier_width": 64, "classifier_depth": 1})
    assert clf.state_dict()["out.2.positional_embedding"].shape == (256, 65)
    m2, _ = create_model_and_diffusion(**{**d, "resblock_updown": False})   # conv Downsample / Upsample: built since round 3 (tests/test_variants.py)
    assert any(k.endswith(".op.weight") for k in m2.state_dict())
    c2 = create_classifier(**{**classifier_defaults(), "classifier_width": 64, "classifier_depth": 1, "classifier_resblock_updown": False})
    assert any(k.endswith(".op.weight") for k in c2.state_dict())   # ... and for the classifier (late round 3: zero-insert + flipped conv backward)
    with pytest.raises(NotImplementedError):
        create_classifier(**{**classifier_defaults(), "classifier_width": 64, "classifier_depth": 1, "classifier_pool": "max"})


def test_hot_path_refuses_to_run_without_a_gpu():
    from autodiffusion_amd._lib import AdmError
    d = model_and_diffusion_defaults()
    d.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2", attention_resolutions="32",
             num_head_channels=32, resblock_updown=True)
    model, _ = create_model_and_diffusion(**d)
    with pytest.raises(AdmError):
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, dtype=torch.int64))


def test_argparse_helpers():
    p = argparse.ArgumentParser()
    add_dict_to_argparser(p, dict(a=1, flag=True, name=None, x=0.5))
    ns = p.parse_args(["--a", "3", "--flag", "no", "--name", "hi"])
    assert args_to_dict(ns, ["a", "flag", "name", "x"]) == dict(a=3, flag=False, name="hi", x=0.5)
    assert str2bool("Yes") is True and str2bool(False) is False
    with pytest.raises(argparse.ArgumentTypeError):
        str2bool("maybe")


def test_asking_for_the_fp32_network_warns_loudly(monkeypatch):
    """`use_fp16=False` (ADM-G-128, configs/128_guided_sample.sh:1) and `classifier_use_fp16=False` (the default of every
    reference classifier, script_util.py:33) cannot be honoured -- the HIP path computes in bf16 with fp32 accumulation --
    so the factories say so once, on the log and as a Python warning, instead of silently handing back a bf16 model."""
    import warnings
    from autodiffusion_amd import logger, unet
    from autodiffusion_amd.script_util import (classifier_defaults, create_classifier, create_model_and_diffusion,
                                               model_and_diffusion_defaults)
    lines = []
    monkeypatch.setattr(logger, "warn", lambda *a: lines.append(" ".join(map(str, a))))
    monkeypatch.setattr(unet, "_warned_precision", set())
    d = model_and_diffusion_defaults()
    d.update(image_size=32, num_channels=32, num_res_blocks=1, channel_mult="1,2", attention_resolutions="16",
             num_head_channels=32, resblock_updown=True, use_scale_shift_norm=True, use_fp16=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m, _ = create_model_and_diffusion(**d)
        create_model_and_diffusion(**d)  # once per process and flag
        cf = classifier_defaults()
        cf.update(classifier_width=64, classifier_depth=1)
        c = create_classifier(**cf)
    assert m.dtype == torch.float32 and m.compute_dtype == torch.bfloat16 and c.compute_dtype == torch.bfloat16
    msgs = [str(x.message) for x in w]
    assert len(msgs) == 2 and "use_fp16=False" in msgs[0] and "classifier_use_fp16=False" in msgs[1]
    assert len(lines) == 2 and all(ln.startswith("WARNING: ") and "bf16" in ln for ln in lines)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d["use_fp16"] = True
        create_model_and_diffusion(**d)
    assert not w


def test_upsample_conv_equals_four_phase_convs_on_the_half_resolution_source():
    """The algebra behind adm_conv_args.up_phase (ops.up_phase_weights): conv3x3(nearest-upsample-2x(x), w) restricted to the
    output pixels (2y + py, 2x + px) is a 3x3 conv of x itself with pre-summed weights of which 5 taps are zero (4/9 of the MACs)."""
    import torch
    import torch.nn.functional as F
    from autodiffusion_amd.ops import up_phase_weights
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64)
    full = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    wp = up_phase_weights(w).to(torch.float64)
    for ph in range(4):
        py, px = ph >> 1, ph & 1
        torch.testing.assert_close(F.conv2d(x, wp[ph], padding=1), full[:, :, py::2, px::2], rtol=1e-5, atol=1e-5)
        live = (wp[ph].abs().sum((0, 1)) > 0)
        assert int(live.sum()) == 4 and bool(live[py:py + 2, px:px + 2].all())      # the 2x2 window the kernel's tap mask names
