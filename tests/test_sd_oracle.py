"""Pin the Stable-Diffusion latent-UNet oracle against golden vectors captured from the reference's own modules
(tests/golden/capture_sd.py -> sd_unet_*.npz)."""
import numpy as np
import pytest
import torch

from autodiffusion_amd.sd_arch import SD_V1, sd_unet_plan
from oracle import sd_nets
from oracle.fill import fill_state_dict

from helpers import golden


def sd_case(name):
    g = golden(name)
    cfg = eval(str(g["cfg"]), {"__builtins__": {}})  # a dict literal written by capture_sd.py
    plan = sd_unet_plan(**cfg)
    P = {k: torch.from_numpy(v) for k, v in fill_state_dict(plan.param_shapes()).items()}
    return g, plan, P


@pytest.mark.parametrize("name", ["sd_unet_tiny", "sd_unet_w320"])
def test_sd_unet_oracle_matches_reference(name):
    g, plan, P = sd_case(name)
    out = sd_nets.sd_unet_forward(P, plan, torch.from_numpy(g["x"]), torch.from_numpy(g["t"]),
                                  torch.from_numpy(g["context"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=2e-4, atol=2e-5)


def test_sd_v1_plan_matches_the_published_architecture():
    plan = sd_unet_plan(**SD_V1)
    shapes = plan.param_shapes()
    assert sum(int(np.prod(s)) for s in shapes.values()) == 859_520_964  # SD v1 UNet (859.5 M)
    assert len(plan.input_blocks) == 12 and len(plan.output_blocks) == 12
    heads = {(b.heads, b.d_head) for b in plan.all_blocks() if hasattr(b, "d_head")}
    assert heads == {(8, 40), (8, 80), (8, 160)}
    assert shapes["input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight"] == (320, 768)
    assert shapes["output_blocks.2.1.conv.weight"] == (1280, 1280, 3, 3)
    assert shapes["output_blocks.5.2.conv.weight"] == (1280, 1280, 3, 3)


# ------------------------------------------------------------------ latent samplers (DDIM / PLMS, searched timesteps)
def test_sd_sampler_oracle_matches_reference_samplers():
    from oracle import sd_sampler as S
    g = golden("sd_samplers")
    ac = S.alphas_cumprod_f32()
    np.testing.assert_array_equal(ac.numpy(), g["alphas_cumprod"])
    for n in (4, 10, 50):
        np.testing.assert_array_equal(S.uniform_timesteps(n), g[f"uniform_{n}"])
    x_T, c, uc = (torch.from_numpy(g[k]) for k in ("x_T", "c", "uc"))
    for tag in ("k4", "k6", "k1"):
        steps = sorted(g[f"cand_{tag}"].tolist())
        for gtag, (scale, u) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
            got = S.ddim_sample(S.toy_model, ac, x_T, c, steps, uc=u, scale=scale)
            np.testing.assert_allclose(got.numpy(), g[f"ddim_{tag}_{gtag}"], rtol=2e-5, atol=2e-5)
            got = S.plms_sample(S.toy_model, ac, x_T, c, steps, uc=u, scale=scale)
            np.testing.assert_allclose(got.numpy(), g[f"plms_{tag}_{gtag}"], rtol=2e-5, atol=2e-5)
    got = S.ddim_sample(S.toy_model, ac, x_T, c, S.uniform_timesteps(4))
    np.testing.assert_allclose(got.numpy(), g["ddim_uniform4_plain"], rtol=2e-5, atol=2e-5)


def test_dpm_solver_oracle_matches_reference_sampler():
    from oracle import sd_sampler as S
    g = golden("sd_samplers")
    ac = S.alphas_cumprod_f32()
    x_T, c, uc = (torch.from_numpy(g[k]) for k in ("x_T", "c", "uc"))
    assert S.dpm_time_points([999, 0])[0] == pytest.approx(0.999001, abs=1e-6) and S.dpm_time_points([999, 0])[1] == pytest.approx(0.001)
    for tag in ("i4", "i6", "f4", "i2"):
        cand = g[f"dpmcand_{tag}"].tolist()
        tp = S.dpm_time_points([int(v) for v in cand] if max(cand) > 1 else cand)
        for gtag, (scale, u) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
            got = S.dpm_sample(S.toy_model, ac, x_T, c, tp, uc=u, scale=scale)
            np.testing.assert_allclose(got.numpy(), g[f"dpm_{tag}_{gtag}"], rtol=1e-4, atol=1e-4, err_msg=f"{tag} {gtag}")
