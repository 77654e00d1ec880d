"""Pin the Stable-Diffusion latent-UNet oracle against golden vectors captured from the reference's own modules
(tests/golden/capture_sd.py -> sd_unet_*.npz)."""
import ast

import numpy as np
import pytest
import torch

from autodiffusion_amd.sd_arch import SD_V1, sd_unet_plan
from oracle import sd_nets
from oracle.fill import fill_state_dict

from helpers import golden


def sd_case(name):
    g = golden(name)
    cfg = ast.literal_eval(str(g["cfg"]))  # a dict literal written by capture_sd.py
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


# ------------------------------------------------------------------ product host logic (no GPU, no compute calls)
def test_sd_sampler_host_tables_match_reference_goldens_and_oracle():
    from autodiffusion_amd import sd_sampler as P
    from oracle import sd_sampler as S
    g = golden("sd_samplers")
    ld = P.LatentDiffusion(None, device="cpu")
    np.testing.assert_array_equal(ld.alphas_cumprod.numpy(), g["alphas_cumprod"])
    np.testing.assert_array_equal(ld.betas.numpy(), g["betas"])
    assert ld.num_timesteps == 1000 and float(ld.alphas_cumprod_prev[0]) == 1.0
    for n in (4, 10, 50):
        np.testing.assert_array_equal(P.make_ddim_timesteps("uniform", n, 1000, verbose=False), g[f"uniform_{n}"])
    with pytest.raises(NotImplementedError):
        P.make_ddim_timesteps("log", 4, 1000)
    ac = S.alphas_cumprod_f32()
    for steps, eta in (([94, 217, 354, 574, 834, 944], 0.0), ([153, 424, 690, 926], 0.7), ([500], 0.3)):
        sig, a, ap = P.make_ddim_sampling_parameters(ac.numpy(), steps, eta)
        osig, oa, oap = S.sampling_parameters(ac, steps, eta)
        np.testing.assert_array_equal(a, oa.numpy())
        np.testing.assert_array_equal(ap, oap.numpy())
        np.testing.assert_allclose(sig, osig.numpy(), rtol=1e-6, atol=0)
    ns, ons = P.NoiseScheduleVP("discrete", alphas_cumprod=ac), S.DiscreteVP(ac)
    for t in (1.0, 0.7502, 0.5, 0.001, 0.0005, 0.00125):
        assert ns.marginal_log_mean_coeff(t) == pytest.approx(ons.log_mean(t), rel=1e-12)
        assert ns.marginal_lambda(t) == pytest.approx(ons.lam(t), rel=1e-12)
        assert ns.marginal_alpha(t) ** 2 + ns.marginal_std(t) ** 2 == pytest.approx(1.0, rel=1e-12)
    # knots reproduce the discrete schedule exactly: t_k = (k+1)/1000 <-> alphas_cumprod[k]
    assert ns.marginal_alpha(0.5) ** 2 == pytest.approx(float(ac[499]), rel=1e-6)
    with pytest.raises(ValueError):
        P.PLMSSampler(ld).make_schedule(4, ddim_eta=0.5)
    with pytest.raises(ValueError):
        P.make_beta_schedule("nope", 10)


def test_sd_samplers_fail_loudly_without_a_gpu():
    """No CPU fallback: with the tables on the CPU the samplers raise instead of computing."""
    from autodiffusion_amd import sd_sampler as P
    from autodiffusion_amd._lib import AdmError
    ld = P.LatentDiffusion(None, device="cpu")
    ld.apply_model = lambda x, t, c: x
    for cls in (P.DDIMSampler, P.PLMSSampler, P.DPMSolverSampler):
        with pytest.raises(AdmError):
            cls(ld).sample(S=4, batch_size=1, shape=[4, 8, 8], conditioning=torch.zeros(1, 2, 8), verbose=False,
                           x_T=torch.zeros(1, 4, 8, 8))


def test_sd_unet_head_padding_is_exact():
    """Zero-padding the 40-channel heads to 64 inside the projection weights leaves q.k and the output unchanged."""
    from autodiffusion_amd.sd_unet import _pad_heads_in, _pad_heads_out, _padded_head
    assert [_padded_head(d) for d in (32, 40, 64, 80, 160, 256)] == [32, 48, 64, 80, 160, 256]
    g = torch.Generator().manual_seed(0)
    heads, d, dp, cin, t = 8, 40, 48, 320, 5
    wq, wk, wv = (torch.randn(heads * d, cin, generator=g) * cin ** -0.5 for _ in range(3))
    wo = torch.randn(cin, heads * d, generator=g) * cin ** -0.5
    x = torch.randn(t, cin, generator=g)

    def attn(wq, wk, wv, wo, dd):
        q, k, v = (x @ w.T for w in (wq, wk, wv))
        q, k, v = (z.reshape(t, heads, dd).permute(1, 0, 2) for z in (q, k, v))
        o = torch.softmax(q @ k.transpose(1, 2) * d ** -0.5, dim=-1) @ v
        return o.permute(1, 0, 2).reshape(t, heads * dd) @ wo.T
    ref = attn(wq, wk, wv, wo, d)
    got = attn(_pad_heads_out(wq, heads, d, dp), _pad_heads_out(wk, heads, d, dp), _pad_heads_out(wv, heads, d, dp),
               _pad_heads_in(wo, heads, d, dp), dp)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


def test_sd_unet_module_surface_on_the_cpu():
    """Constructor bookkeeping and the nn.Module-like surface need no GPU; the forward fails loudly without one."""
    from autodiffusion_amd._lib import AdmError
    from autodiffusion_amd.sd_unet import UNetModel
    m = UNetModel(image_size=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[1, 2],
                  num_res_blocks=1, channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1,
                  context_dim=96, use_checkpoint=True, legacy=False)
    sd = m.state_dict()
    assert sum(v.numel() for v in sd.values()) == 4_237_956  # the reference module's count (capture_sd.py printed it)
    assert m.accepts_context_key and m.num_classes is None
    with pytest.raises(RuntimeError):
        m.load_state_dict({k: v for k, v in list(sd.items())[1:]})          # a missing key
    bad = dict(sd)
    bad["out.2.bias"] = torch.zeros(5)
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad)                                              # a shape mismatch
    m.load_state_dict(sd)
    with pytest.raises(AdmError):
        m(torch.zeros(1, 4, 16, 16), torch.zeros(1, dtype=torch.int64), torch.zeros(1, 3, 96))  # CPU tensors: no fallback
    with pytest.raises(NotImplementedError):
        UNetModel(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1],
                  num_heads=2, use_spatial_transformer=False, context_dim=None)


def test_geglu_interleave_is_the_chunk2_reordering():
    """ops.geglu_interleave: rows (values | gates) of the GEGLU projection -> (value m, gate m) pairs; value * gelu(gate) of the
    de-interleaved product equals the reference's `x, gate = proj(x).chunk(2, dim=-1)` form."""
    import torch.nn.functional as F
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(3)
    w, b, x = torch.randn(24, 5, generator=g), torch.randn(24, generator=g), torch.randn(7, 5, generator=g)
    wi, bi = ops.geglu_interleave(w, b)
    assert torch.equal(wi[0::2], w[:12]) and torch.equal(wi[1::2], w[12:]) and torch.equal(bi[0::2], b[:12]) and torch.equal(bi[1::2], b[12:])
    val, gate = F.linear(x, w, b).chunk(2, dim=-1)
    ui = F.linear(x, wi, bi)
    torch.testing.assert_close(ui[:, 0::2] * F.gelu(ui[:, 1::2]), val * F.gelu(gate), rtol=1e-6, atol=1e-6)
