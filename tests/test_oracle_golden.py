"""Pin the CPU oracle against golden vectors captured from the reference itself.

Fixtures: tests/golden/*.npz, produced by tests/golden/capture_golden.py by
importing the reference's guided_diffusion modules.
"""
import numpy as np
import pytest
import torch

from autodiffusion_amd.arch import AttnSpec, ResBlockSpec
from oracle import nets, sampler, schedule
from oracle.fill import fill_state_dict

from helpers import filled, golden, plan_c64, plan_m32, plan_m64

TOL = dict(rtol=2e-4, atol=2e-5)


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_space_timesteps_and_betas():
    g = golden("schedules")
    for key in ("ddim4", "ddim10", "4", "10,15", "25"):
        want = g["space_" + key.replace(",", "_")]
        assert sorted(schedule.spaced_steps(1000, key)) == want.tolist()
    assert np.array_equal(schedule.named_betas("cosine", 1000), g["betas_cosine"])
    assert np.array_equal(schedule.named_betas("linear", 1000), g["betas_linear"])
    with pytest.raises(ValueError):
        schedule.spaced_steps(1000, "ddim999")


def test_spaced_constructor_tables():
    g = golden("spaced_ddim4_cosine")
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", timestep_respacing="ddim4",
                                 learn_sigma=True)
    assert d.timestep_map == g["timestep_map"].tolist() == [0, 250, 500, 750]
    for n in schedule.TABLE_NAMES:
        assert np.array_equal(d.tables[n], g[n]), n


@pytest.mark.parametrize("sched", ["cosine", "linear"])
def test_reset_diffusion_tables_bit_exact(sched):
    g = golden(f"reset_diffusion_{sched}")
    d = schedule.OracleDiffusion(steps=1000, noise_schedule=sched, learn_sigma=True)
    for tag in ("k4", "k6", "k10", "k1"):
        d.reset(g[f"{tag}_cand"].tolist())
        assert d.timestep_map == g[f"{tag}_timestep_map"].tolist()
        assert d.num_timesteps == int(g[f"{tag}_num_timesteps"])
        for n in schedule.TABLE_NAMES:
            assert np.array_equal(d.tables[n], g[f"{tag}_{n}"]), (tag, n)
    # K == 1 quirk: raw variance, not its log
    d.reset([500])
    assert np.array_equal(d.tables["posterior_log_variance_clipped"], d.tables["posterior_variance"])


def test_timestep_embedding():
    g = golden("timestep_embedding")
    t = T(g["t"])
    for dim in (192, 32, 33):
        np.testing.assert_allclose(nets.sinusoid_embedding(t, dim).numpy(), g[f"dim{dim}"], rtol=1e-6, atol=1e-6)


def test_groupnorm32():
    g = golden("groupnorm32")
    sd = fill_state_dict({"gn.weight": (64,), "gn.bias": (64,)})
    y = nets.group_norm(T(g["x"]), T(sd["gn.weight"]), T(sd["gn.bias"]))
    np.testing.assert_allclose(y.numpy(), g["y"], **TOL)


@pytest.mark.parametrize("name", ["qkv_attention", "qkv_attention_d64"])
def test_qkv_attention_orders(name):
    g = golden(name)
    h = int(g["heads"])
    np.testing.assert_allclose(nets.qkv_attention(T(g["qkv"]), h, True).numpy(), g["new"], **TOL)
    np.testing.assert_allclose(nets.qkv_attention(T(g["qkv"]), h, False).numpy(), g["legacy"], **TOL)


@pytest.mark.parametrize("tag,kw", [("plain", {}), ("down", dict(down=True)), ("up", dict(up=True)),
                                    ("skipconv", dict(cout=96)), ("noss", dict(scale_shift=False))])
def test_resblock_variants(tag, kw):
    g = golden("resblock")
    spec = ResBlockSpec(prefix=f"rb_{tag}", cin=64, cout=kw.pop("cout", 64), emb_dim=128, **kw)
    P = nets.params_from_numpy(fill_state_dict(spec.param_shapes()))
    y = nets.resblock(P, spec, T(g[f"{tag}_x"]), T(g["emb"]))
    np.testing.assert_allclose(y.numpy(), g[f"{tag}_y"], **TOL)


@pytest.mark.parametrize("tag,heads,new", [("new", 2, True), ("legacy", 2, False)])
def test_attention_block(tag, heads, new):
    g = golden("attention_block")
    spec = AttnSpec(prefix=f"attn_{tag}", channels=64, num_heads=heads, new_order=new)
    P = nets.params_from_numpy(fill_state_dict(spec.param_shapes()))
    y = nets.attention_block(P, spec, T(g[f"{tag}_x"]))
    np.testing.assert_allclose(y.numpy(), g[f"{tag}_y"], **TOL)


def test_dynamic_unet_with_skip_lists():
    g = golden("unet_m32")
    plan = plan_m32(dynamic=True)
    assert plan.layer_num == int(g["layer_num"])
    P = nets.params_from_numpy(filled(plan))
    for tag in ("none", "a", "b", "all"):
        out = nets.unet_forward(P, plan, T(g["x"]), T(g["t"]), T(g["y"]), skip_layer=g[f"skip_{tag}"].tolist())
        np.testing.assert_allclose(out.numpy(), g[f"out_{tag}"], rtol=1e-3, atol=1e-4)


def test_unet_legacy_attention_order():
    g = golden("unet_m32_legacy")
    plan = plan_m32(dynamic=False, legacy=True)
    P = nets.params_from_numpy(filled(plan))
    out = nets.unet_forward(P, plan, T(g["x"]), T(g["t"]), T(g["y"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=1e-4)


def test_unet_m64():
    g = golden("unet_m64")
    plan = plan_m64()
    P = nets.params_from_numpy(filled(plan))
    out = nets.unet_forward(P, plan, T(g["x"]), T(g["t"]), T(g["y"]))
    np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-3, atol=1e-4)


def test_layer_count_adm64():
    from autodiffusion_amd.arch import build_unet_plan
    plan = build_unet_plan(64, 3, 192, 6, 3, (2, 4, 8), (1, 2, 3, 4), num_classes=1000,
                           num_head_channels=64, use_scale_shift_norm=True, resblock_updown=True,
                           use_new_attention_order=True, dynamic=True)
    assert plan.layer_num == 58
    n = sum(int(np.prod(s)) for s in plan.param_shapes().values())
    assert abs(n - 295.9e6) < 0.1e6


def test_classifier_logits_and_input_gradient():
    g = golden("classifier_c64")
    plan = plan_c64()
    P = nets.params_from_numpy(filled(plan))
    x, t, y = T(g["x"]), T(g["t"]), T(g["y"])
    logits = nets.unet_forward(P, plan, x, t)
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=1e-3, atol=1e-4)
    grad = nets.classifier_grad(P, plan, x, t, y, 1.0)
    np.testing.assert_allclose(grad.numpy(), g["grad"], rtol=2e-3, atol=1e-5)


def _guided_setup():
    plan, cplan = plan_m64(dynamic=True), plan_c64()
    P, CP = nets.params_from_numpy(filled(plan)), nets.params_from_numpy(filled(cplan))

    def model_fn(x, t, y=None, skip_layers=None, _map=None):
        sl = ()
        if skip_layers is not None:
            sl = skip_layers[_map.index(int(t[0]))]
        return nets.unet_forward(P, plan, x, t, y, skip_layer=sl)

    def cond_fn(x, t, y=None, **kw):
        return nets.classifier_grad(CP, cplan, x, t, y, 1.0)

    return model_fn, cond_fn


def test_single_steps_ddim_and_ddpm():
    g = golden("sampler_steps_m64")
    model_fn, cond_fn = _guided_setup()
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True).reset(g["cand"].tolist())
    x, y = T(g["x"]), T(g["y"])
    for idx in (2, 0):
        t = torch.full((2,), d.timestep_map[idx], dtype=torch.int64)
        mo = model_fn(x, t, y)
        if idx == 2:
            np.testing.assert_allclose(mo.numpy(), g["model_out_i2"], rtol=1e-3, atol=1e-4)
        grad = cond_fn(x, t, y=y)
        nz = T(g[f"noise_i{idx}"])
        for guided, gr in (("u", None), ("g", grad)):
            o = sampler.ddim_step(d, mo, x, idx, gr, nz)
            np.testing.assert_allclose(o["sample"].numpy(), g[f"ddim_i{idx}_{guided}_sample"], rtol=1e-3, atol=2e-4)
            np.testing.assert_allclose(o["pred_xstart"].numpy(), g[f"ddim_i{idx}_{guided}_x0"], rtol=1e-3, atol=2e-4)
            o = sampler.ddim_step(d, mo, x, idx, gr, nz, eta=0.7)
            np.testing.assert_allclose(o["sample"].numpy(), g[f"ddim_eta_i{idx}_{guided}_sample"], rtol=1e-3, atol=2e-4)
            o = sampler.ddpm_step(d, mo, x, idx, gr, nz)
            np.testing.assert_allclose(o["sample"].numpy(), g[f"ddpm_i{idx}_{guided}_sample"], rtol=1e-3, atol=2e-4)
            np.testing.assert_allclose(o["pred_xstart"].numpy(), g[f"ddpm_i{idx}_{guided}_x0"], rtol=1e-3, atol=2e-4)


def test_full_loops_guided_and_unguided():
    g = golden("sampler_loops_m64")
    model_fn, cond_fn = _guided_setup()
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True).reset(g["cand"].tolist())
    xT, y = T(g["x_T"]), T(g["y"])
    noises = [T(n) for n in g["noises"]]
    for name, ddim in (("ddim", True), ("ddpm", False)):
        for tag, cf in (("u", None), ("g", cond_fn)):
            s = sampler.sample_loop(d, model_fn, xT, use_ddim=ddim, cond_fn=cf, noises=noises,
                                    model_kwargs={"y": y})
            np.testing.assert_allclose(s.numpy(), g[f"{name}_{tag}_sample"], rtol=2e-3, atol=1e-3)
            u8 = sampler.pack_uint8_nhwc(s).numpy()
            assert u8.shape == (2, 64, 64, 3) and u8.dtype == np.uint8
            diff = np.abs(u8.astype(int) - g[f"{name}_{tag}_uint8"].astype(int))
            assert diff.max() <= 1 and (diff > 0).mean() < 0.01  # truncation boundary flips only
    skip_layers = [[int(v) for v in s.split(",") if v] for s in g["skip_layers"]]
    s = sampler.sample_loop(d, model_fn, xT, use_ddim=True, cond_fn=cond_fn, noises=noises,
                            model_kwargs={"y": y, "skip_layers": skip_layers, "_map": d.timestep_map})
    np.testing.assert_allclose(s.numpy(), g["ddim_g_skip_sample"], rtol=2e-3, atol=1e-3)


def test_unconditional_uniform_ddim4_loops():
    g = golden("sampler_loops_m32_uncond")
    plan = plan_m32(dynamic=False, class_cond=False)
    P = nets.params_from_numpy(filled(plan))
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True, timestep_respacing="ddim4")
    fn = lambda x, t: nets.unet_forward(P, plan, x, t)  # noqa: E731
    noises = [T(n) for n in g["noises"]]
    for name, ddim in (("ddim", True), ("ddpm", False)):
        s = sampler.sample_loop(d, fn, T(g["x_T"]), use_ddim=ddim, noises=noises)
        np.testing.assert_allclose(s.numpy(), g[f"{name}_sample"], rtol=2e-3, atol=1e-3)


def test_oracle_at_baseline_size_matches_the_reference_captures():
    """The oracle pinned at BASELINE size too: the 295.9 M-parameter ADM-G-64 UNet and the depth-4 64x64 classifier's
    guidance gradient against outputs of the reference's own modules (tests/golden/capture_fullsize.py)."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import adm64_flags
    from autodiffusion_amd.script_util import (classifier_defaults, create_classifier, create_model_and_diffusion)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = golden("full_adm64")
    plan = create_model_and_diffusion(**adm64_flags(class_cond=True, dynamic=False))[0].plan
    P = nets.params_from_numpy(filled(plan))
    with torch.no_grad():
        out = nets.unet_forward(P, plan, T(g["x"][:1]), T(g["t"][:1]), T(g["y"][:1]))
    r = float((out - T(g["out"][:1])).norm() / T(g["out"][:1]).norm())
    assert r < 1e-4, r
    del P
    gc = golden("full_clf64")
    cf = classifier_defaults()
    cf.update(image_size=64, classifier_depth=4)
    cplan = create_classifier(**cf).plan
    Pc = nets.params_from_numpy(filled(cplan))
    grad = nets.classifier_grad(Pc, cplan, T(gc["x"]), T(gc["t"]), T(gc["y"]), 1.0)
    r = float((grad - T(gc["grad"])).norm() / T(gc["grad"]).norm())
    assert r < 1e-3, r


def test_oracle_class_conditional_256_matches_the_reference_capture():
    """BASELINE configs[4] as written (search_lsun_cat.sh:1 + class_cond, 553.8 M parameters): the oracle's dynamic UNet with
    `label_emb` on the 6-level model against the reference's own output, with and without a layer-skip list (every second pixel)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import adm256_flags
    from autodiffusion_amd.script_util import create_model_and_diffusion
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    g = golden("full_adm256cc")
    flags = adm256_flags()
    flags["class_cond"] = True
    plan = create_model_and_diffusion(**flags)[0].plan
    assert plan.layer_num == int(g["layer_num"]) and sum(int(np.prod(s)) for s in plan.param_shapes().values()) == int(g["params"])
    P = nets.params_from_numpy(filled(plan))
    with torch.no_grad():
        for tag, skip in (("out", []), ("out_skip", g["skip"].tolist())):
            out = nets.unet_forward(P, plan, T(g["x"]), T(g["t"]), T(g["y"]), skip_layer=skip)[:, :, ::2, ::2]
            r = float((out - T(g[f"{tag}_sub"])).norm() / T(g[f"{tag}_sub"]).norm())
            assert r < 1e-4, (tag, r)


def test_denoised_fn_steps_and_loops_match_the_reference_capture():
    """`denoised_fn` acts on the predicted x_0 before the clip (gaussian_diffusion.py:293-298): single ddim (eta 0.3) / ddpm steps,
    guided and unguided, clipped and not, and both guided loops against tests/golden/capture_denoised.py's reference outputs."""
    from helpers import denoised_fn_fixture as dfn
    g = golden("sampler_denoised_m64")
    model_fn, cond_fn = _guided_setup()
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True).reset(g["cand"].tolist())
    x, y = T(g["x"]), T(g["y"])
    for idx in (2, 0):
        t = torch.full((1,), d.timestep_map[idx], dtype=torch.int64)
        mo = model_fn(x, t, y)
        grad = cond_fn(x, t, y=y)
        nz = T(g[f"noise_i{idx}"])
        for guided, gr in (("u", None), ("g", grad)):
            for clip in ((True, False) if (idx == 2 and gr is not None) else (True,)):
                tag = f"i{idx}_{guided}" + ("" if clip else "_noclip")
                o = sampler.ddim_step(d, mo, x, idx, gr, nz, eta=0.3, clip_denoised=clip, denoised_fn=dfn)
                np.testing.assert_allclose(o["sample"].numpy(), g[f"ddim_{tag}_sample"], rtol=1e-3, atol=2e-4)
                np.testing.assert_allclose(o["pred_xstart"].numpy(), g[f"ddim_{tag}_x0"], rtol=1e-3, atol=2e-4)
                o = sampler.ddpm_step(d, mo, x, idx, gr, nz, clip_denoised=clip, denoised_fn=dfn)
                np.testing.assert_allclose(o["sample"].numpy(), g[f"ddpm_{tag}_sample"], rtol=1e-3, atol=2e-4)
                np.testing.assert_allclose(o["pred_xstart"].numpy(), g[f"ddpm_{tag}_x0"], rtol=1e-3, atol=2e-4)
    noises = [T(n) for n in g["noises"]]
    for name, ddim in (("ddim", True), ("ddpm", False)):
        s = sampler.sample_loop(d, model_fn, x, use_ddim=ddim, cond_fn=cond_fn, noises=noises, model_kwargs={"y": y}, denoised_fn=dfn)
        np.testing.assert_allclose(s.numpy(), g[f"{name}_loop_sample"], rtol=2e-3, atol=1e-3)
