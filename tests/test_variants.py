"""The reference UNet variants no launch script uses but the factories offer -- ``use_scale_shift_norm=False`` (unet.py:251-254) and
``resblock_updown=False`` (conv ``Downsample`` / ``Upsample``, unet.py:78-141; the reference factory's DEFAULTS are both False) -- against
outputs captured from the reference itself (tests/golden/capture_variants.py): the CPU oracle here (pins the plan, the key layout and
the restatement), the HIP path in the `gpu` test."""
import numpy as np
import pytest
import torch

from helpers import filled, golden
from autodiffusion_amd.arch import ResampleSpec, build_unet_plan
from oracle import nets

CASES = {"noss": (False, True, True), "convres": (True, False, False), "defaults": (False, False, True)}


def plan_of(tag):
    ss, ud, dyn = CASES[tag]
    return build_unet_plan(image_size=32, in_channels=3, model_channels=32, out_channels=6, num_res_blocks=1,
                           attention_resolutions=(2, 4), channel_mult=(1, 2, 2), num_classes=1000, num_head_channels=32,
                           use_scale_shift_norm=ss, resblock_updown=ud, use_new_attention_order=True, dynamic=dyn)


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_matches_the_reference_on_the_variant(tag):
    g = golden("unet_m32_variants")
    plan = plan_of(tag)
    sd = filled(plan)
    assert sum(int(np.prod(v.shape)) for v in sd.values()) == int(g[f"nparams_{tag}"])      # same keys' worth of parameters
    if CASES[tag][2]:
        assert plan.layer_num == int(g[f"layer_num_{tag}"])      # Downsample / Upsample layers carry no layer id
    if not CASES[tag][1]:
        rs = [b for b in plan.all_blocks() if isinstance(b, ResampleSpec)]
        assert len(rs) == 4 and all(b.use_conv for b in rs)
    P = nets.params_from_numpy(sd)
    x, t, y = (torch.from_numpy(g[k]) for k in ("x", "t", "y"))
    out = nets.unet_forward(P, plan, x, t, y)
    torch.testing.assert_close(out, torch.from_numpy(g[f"out_{tag}"]), rtol=2e-4, atol=2e-5)
    if CASES[tag][2]:
        outs = nets.unet_forward(P, plan, x, t, y, skip_layer=g[f"skip_{tag}"].tolist())
        torch.testing.assert_close(outs, torch.from_numpy(g[f"out_{tag}_skip"]), rtol=2e-4, atol=2e-5)


def test_factory_builds_the_reference_default_flags():
    """model_and_diffusion_defaults() itself has use_scale_shift_norm=True but resblock_updown=False: the factory must build it."""
    from autodiffusion_amd.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    d = model_and_diffusion_defaults()
    d.update(image_size=64, num_channels=32, num_res_blocks=1, attention_resolutions="16")
    model, _ = create_model_and_diffusion(**d)
    keys = list(model.state_dict())
    assert any(k.endswith(".op.weight") for k in keys) and any(k.endswith(".conv.weight") for k in keys)
    # the classifier's backward network differentiates the FiLM / ResBlock-resample forms only
    from autodiffusion_amd.script_util import classifier_defaults, create_classifier
    for bad in ({"classifier_resblock_updown": False}, {"classifier_use_scale_shift_norm": False}):
        with pytest.raises(NotImplementedError):
            create_classifier(**{**classifier_defaults(), "classifier_width": 64, "classifier_depth": 1, **bad})


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_hip_unet_matches_the_reference_on_the_variant(tag):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd.unet import UNetModel
    g = golden("unet_m32_variants")
    plan = plan_of(tag)
    model = UNetModel(plan)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled(plan).items()})
    model.to("cuda:0").eval()
    x, t, y = (torch.from_numpy(g[k]).to("cuda:0") for k in ("x", "t", "y"))
    for torso, bound in (("bf16", 2e-2), ("fp16", 4e-3)):
        model.set_torso(torso)
        out = model(x, t, y).cpu()
        ref = torch.from_numpy(g[f"out_{tag}"])
        r = float((out - ref).norm() / ref.norm())
        print(f"variant {tag}, {torso}: rel {r:.3e}")
        assert torch.isfinite(out).all() and r < bound, (tag, torso, r)
        assert torch.equal(model(x[:1], t[:1], y[:1]).cpu(), out[:1])
        if CASES[tag][2]:
            outs = model(x, t, y, skip_layer=g[f"skip_{tag}"].tolist()).cpu()
            refs = torch.from_numpy(g[f"out_{tag}_skip"])
            rs = float((outs - refs).norm() / refs.norm())
            assert rs < bound, (tag, torso, rs)
    model.set_torso("bf16")
    model.upconv_phases = False      # the Upsample convs as one 9-tap launch
    r1 = float((model(x, t, y).cpu() - ref).norm() / ref.norm())
    assert r1 < 2e-2, r1
