"""GPU parity of the Stable-Diffusion latent UNet on the HIP path: the token kernels against
plain PyTorch fp32, the whole network against golden vectors captured from the reference's own modules
(tests/golden/capture_sd.py) and against the golden-pinned CPU oracle.

Tolerance: as for the ADM torso (test_hip_unet.py) -- bf16 activations / operands with fp32 accumulation, fp32
normalisation statistics and softmax: relative Frobenius error <= 2e-2, max |err| <= 6e-2 * max|ref|.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden
from test_sd_oracle import sd_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def check(got, ref, what, fro_tol=2e-2, max_tol=6e-2):
    got, ref = got.float().cpu().numpy(), np.asarray(ref, dtype=np.float32)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), what
    fro = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    mx = np.abs(got - ref).max() / np.abs(ref).max()
    print(f"{what}: fro {fro:.4g} max {mx:.4g}")
    assert fro <= fro_tol and mx <= max_tol, f"{what}: fro {fro:.4g} max {mx:.4g}"


def bf(x):
    return x.to(torch.bfloat16)


@pytest.mark.parametrize("rows,c", [(7, 64), (1024, 320), (300, 640), (130, 1280), (5, 2048)])
def test_layernorm(rows, c):
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(rows + c)
    x = bf(torch.randn(rows, c, generator=g) * 2 + 0.5).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV), (0.1 * torch.randn(c, generator=g)).to(DEV)
    ref = F.layer_norm(x.float(), (c,), gamma, beta, eps=1e-5)
    check(ops.layernorm(x, gamma, beta), ref.cpu(), f"layernorm {rows}x{c}", 5e-3, 2e-2)


@pytest.mark.parametrize("rows,inner", [(9, 256), (4096, 1280)])
def test_geglu(rows, inner):
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(rows)
    u = bf(torch.randn(rows, 2 * inner, generator=g) * 1.5).to(DEV)
    a, gate = u.float().chunk(2, dim=-1)
    check(ops.geglu(u), (a * F.gelu(gate)).cpu(), f"geglu {rows}x{inner}", 5e-3, 2e-2)


@pytest.mark.parametrize("n,hw,c,dtype", [(2, 32, 320, torch.bfloat16), (1, 64, 320, torch.float16), (3, 16, 640, torch.bfloat16),
                                          (1, 8, 384, torch.bfloat16), (2, 8, 64, torch.float16)])
def test_geglu_in_the_projection_epilogue(n, hw, c, dtype):
    """conv(geglu=True): the GEGLU feed-forward's `proj(x).chunk(2)` -> x * gelu(gate) (SD/ldm/modules/attention.py:37-44) formed in
    the 1x1 projection's epilogue from interleaved (value, gate) weight rows, against PyTorch fp32 on the same 16-bit operands; the
    fused form is closer to it than projection -> 16-bit tensor -> adm_geglu (the gate is not rounded in between)."""
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(hw + c)
    inner = 4 * c
    x = (torch.randn(n, hw, hw, c, generator=g)).to(dtype).to(DEV)
    w = torch.randn(2 * inner, c, generator=g) * c ** -0.5 * 1.5
    b = 0.1 * torch.randn(2 * inner, generator=g)
    wq = w.to(dtype).float()
    u = F.linear(x.float().cpu(), wq, b)
    val, gate = u.chunk(2, dim=-1)
    ref = val * F.gelu(gate)
    wi, bi = ops.geglu_interleave(w, b)
    got = ops.conv(x, ops.pack_conv_weight(wi.to(DEV), dtype), bi.to(DEV), 2 * inner, 1, geglu=True)
    assert got.shape == (n, hw, hw, inner) and got.dtype == dtype
    check(got, ref, f"fused geglu {n}x{hw}x{hw}x{c}", 4e-3, 2e-2)
    two = ops.geglu(ops.conv(x, ops.pack_conv_weight(w.to(DEV), dtype), b.to(DEV), 2 * inner, 1))
    check(two, ref, f"two-pass geglu {n}x{hw}x{hw}x{c}", 8e-3, 3e-2)
    e1, e2 = float((got.float().cpu() - ref).norm()), float((two.float().cpu() - ref).norm())
    assert e1 <= e2 * 1.05, (e1, e2)
    from autodiffusion_amd._lib import AdmError
    with pytest.raises(AdmError):   # not a shape the resident-tile kernel takes: refused, never silently unfused
        ops.conv(x[:, :, :, :32].contiguous(), ops.pack_conv_weight(wi[:, :32].contiguous().to(DEV), dtype), bi.to(DEV), 2 * inner, 1, geglu=True)


@pytest.mark.parametrize("n,tq,tk,rows,heads,d,true_d", [
    (2, 256, 77, 128, 8, 64, 40),     # cross-attention of the 320-channel level (40-wide heads padded to 64)
    (1, 1024, 77, 128, 8, 128, 80),
    (2, 1024, 77, 128, 8, 96, 80),    # 80-wide heads padded to 96
    (2, 512, 77, 128, 8, 48, 40),     # 40-wide heads padded to 48: one 32-deep + one 16-deep MFMA step
    (1, 1024, 1024, 1024, 8, 80, 80), # 80-wide heads as they are, long keys
    (3, 70, 130, 256, 2, 48, 40),     # ... ragged
    (1, 256, 77, 128, 8, 160, 160),   # 160-wide heads as they are
    (2, 64, 7, 128, 2, 32, 32),
    (1, 256, 77, 128, 8, 192, 160),
    (3, 100, 130, 256, 4, 64, 64),    # ragged on both sides, more than one key tile
    (1, 1, 1, 128, 1, 32, 32),        # a single query over a single key
    (2, 33, 64, 64, 2, 64, 64),       # exactly one key tile, kv_rows == tk
    (2, 65, 65, 128, 3, 96, 80),      # one key / one query past a tile boundary
    (1, 17, 200, 256, 2, 160, 160),   # wide kernel, several ragged key tiles
])
def test_attention_cross(n, tq, tk, rows, heads, d, true_d):
    from autodiffusion_amd import ops
    g = torch.Generator().manual_seed(tq + tk)
    q = bf(torch.randn(n, tq, heads * d, generator=g)).to(DEV)
    kv = bf(torch.randn(n, rows, 2 * heads * d, generator=g)).to(DEV)
    scale = true_d ** -0.5
    out = ops.attention_cross(q, kv, heads, d, tk, scale)
    qf = q.float().view(n, tq, heads, d).permute(0, 2, 1, 3)
    kf = kv.float()[:, :tk, :heads * d].reshape(n, tk, heads, d).permute(0, 2, 1, 3)
    vf = kv.float()[:, :tk, heads * d:].reshape(n, tk, heads, d).permute(0, 2, 1, 3)
    w = torch.softmax(qf @ kf.transpose(-1, -2) * scale, dim=-1)
    ref = (w @ vf).permute(0, 2, 1, 3).reshape(n, tq, heads * d)
    check(out, ref.cpu(), f"attention_cross tq={tq} tk={tk} d={d}", 1e-2, 3e-2)


def test_self_attention_through_the_fused_projection_alias():
    from autodiffusion_amd import ops
    n, t, heads, d = 2, 320, 4, 64
    g = torch.Generator().manual_seed(5)
    qkv = bf(torch.randn(n, t, 3 * heads * d, generator=g)).to(DEV)
    out = ops.attention_cross(qkv, qkv[:, :, heads * d:], heads, d, t, 40 ** -0.5)
    q, k, v = (z.reshape(n, t, heads, d).permute(0, 2, 1, 3) for z in qkv.float().chunk(3, dim=-1))
    ref = (torch.softmax(q @ k.transpose(-1, -2) * 40 ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(n, t, heads * d)
    check(out, ref.cpu(), "self-attention alias", 1e-2, 3e-2)


def test_gn_affine_with_channel_addend_and_stride2():
    from autodiffusion_amd import ops
    n, hw, c = 3, 16, 320
    g = torch.Generator().manual_seed(11)
    x = bf(torch.randn(n, hw, hw, c, generator=g)).to(DEV)
    e = torch.randn(n, 2 * c, generator=g).to(DEV)[:, c // 2:c // 2 + c]  # a strided row view, as the model passes
    gamma, beta = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV), (0.1 * torch.randn(c, generator=g)).to(DEV)
    a, b = ops.gn_affine(x, gamma, beta, add=e)
    got = a[:, None, None, :] * x.float() + b[:, None, None, :]
    ref = F.group_norm((x.float() + e[:, None, None, :]).permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-5).permute(0, 2, 3, 1)
    check(got, ref.cpu(), "gn(x + e)", 1e-4, 1e-3)
    a6, b6 = ops.gn_affine(x, gamma, beta, eps=1e-6)
    ref6 = F.group_norm(x.float().permute(0, 3, 1, 2), 32, gamma, beta, eps=1e-6).permute(0, 2, 3, 1)
    check(a6[:, None, None, :] * x.float() + b6[:, None, None, :], ref6.cpu(), "gn eps 1e-6", 1e-4, 1e-3)
    assert torch.equal(ops.resample(x, "stride2"), x[:, ::2, ::2].contiguous())


def _model(plan, P):
    from autodiffusion_amd.sd_unet import UNetModel
    m = UNetModel.__new__(UNetModel)
    from autodiffusion_amd.unet import HipModule
    HipModule.__init__(m, plan, False)
    m.num_classes = None
    m.load_state_dict(P)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name", ["sd_unet_tiny", "sd_unet_w320"])
def test_sd_unet_matches_reference_golden(name):
    g, plan, P = sd_case(name)
    m = _model(plan, P)
    x, t, ctx = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "context"))
    out = m(x, t, ctx)
    assert out.dtype == torch.float32 and out.shape == g["out"].shape
    check(out, g["out"], name)
    m.enable_graph()  # hipGraph replay: bit-identical, and a second call with new inputs replays the same graph
    assert torch.equal(m(x, t, ctx), out)
    x2 = x.flip(0).contiguous() if x.shape[0] > 1 else x * 0.5
    assert torch.equal(m(x2, t, ctx), m.enable_graph(False)(x2, t, ctx))
    # context_key: the k|v projections are computed once per key and reused (eager and graph mode), bit-identically
    for graph in (False, True):
        m.enable_graph(graph)
        assert torch.equal(m(x, t, ctx, context_key="a"), out) and torch.equal(m(x, t, ctx, context_key="a"), out)
        ctx2 = ctx * 0.5
        want2 = m(x, t, ctx2)
        assert not torch.equal(want2, out)
        assert torch.equal(m(x, t, ctx2, context_key="b"), want2) and torch.equal(m(x, t, ctx, context_key="c"), out)
    m.enable_graph(False)
    # Upsample convs as one 9-tap launch instead of the four 2x2-tap phase convs (the default): the same network up to the rounding
    # of the pre-summed taps
    assert m.upconv_phases
    one = m.enable_upconv_phases(False)(x, t, ctx)
    m.enable_upconv_phases(True)
    check(one, g["out"], name + " (one-launch upsample convs)")
    assert not torch.equal(one, out) or x.shape[-1] < 32      # (maps below 16x16 keep the one-launch path either way)
    if name == "sd_unet_tiny":  # ragged batch reproduces per-image results
        out3 = m(torch.cat([x, x[:1]]), torch.cat([t, t[:1]]), torch.cat([ctx, ctx[:1]]))
        assert torch.equal(out3[:2], out) and torch.equal(out3[2], out[0])


def test_sd_constructor_mirrors_the_reference_and_rejects_unbuilt_variants():
    from autodiffusion_amd.sd_unet import UNetModel
    m = UNetModel(image_size=32, in_channels=4, out_channels=4, model_channels=64, attention_resolutions=[1, 2],
                  num_res_blocks=1, channel_mult=[1, 2], num_heads=2, use_spatial_transformer=True, transformer_depth=1,
                  context_dim=96, use_checkpoint=True, legacy=False)
    sd = m.state_dict()
    assert "input_blocks.1.1.transformer_blocks.0.attn2.to_k.weight" in sd and "input_blocks.2.0.op.weight" in sd
    assert float(sd["out.2.weight"].abs().max()) == 0.0  # zero_module
    out = m.to(DEV)(torch.randn(1, 4, 16, 16, device=DEV), torch.tensor([3], device=DEV), torch.randn(1, 5, 96, device=DEV))
    assert float(out.abs().max()) == 0.0  # a freshly built model outputs exactly zero, like the reference's
    with pytest.raises(NotImplementedError):
        UNetModel(in_channels=4, model_channels=64, out_channels=4, num_res_blocks=1, attention_resolutions=[1],
                  num_heads=2, use_spatial_transformer=True, context_dim=96, use_scale_shift_norm=True)


# ------------------------------------------------------------------ latent samplers
class _ToyLatentModel:
    """The attributes of LatentDiffusion the samplers read; apply_model = the capture script's toy model, on the GPU."""

    def __init__(self):
        from autodiffusion_amd.sd_sampler import LatentDiffusion
        base = LatentDiffusion(None, device=DEV)
        self.num_timesteps, self.device = base.num_timesteps, base.device
        self.betas, self.alphas_cumprod, self.alphas_cumprod_prev = base.betas, base.alphas_cumprod, base.alphas_cumprod_prev
        self.calls = 0

    def apply_model(self, x, t, c):
        from oracle.sd_sampler import toy_model
        self.calls += 1
        return toy_model(x, t, c)


def test_latent_samplers_match_reference_goldens():
    from autodiffusion_amd.sd_sampler import DDIMSampler, PLMSSampler
    g = golden("sd_samplers")
    x_T, c, uc = (torch.from_numpy(g[k]).to(DEV) for k in ("x_T", "c", "uc"))
    m = _ToyLatentModel()
    np.testing.assert_array_equal(m.alphas_cumprod.cpu().numpy(), g["alphas_cumprod"])
    for tag in ("k4", "k6", "k1"):
        cand = g[f"cand_{tag}"]
        for name, cls in (("ddim", DDIMSampler), ("plms", PLMSSampler)):
            for gtag, (scale, u) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
                st = np.array(sorted(cand)) if name == "plms" else cand
                s = cls(m)
                assert s.ddpm_num_timesteps == 1000
                got, inter = s.sample(S=len(cand), batch_size=3, shape=[4, 8, 8], conditioning=c, verbose=False, eta=0.0,
                                      x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=u,
                                      sampled_timestep=st)
                np.testing.assert_allclose(got.cpu().numpy(), g[f"{name}_{tag}_{gtag}"], rtol=2e-5, atol=2e-5,
                                           err_msg=f"{name} {tag} {gtag}")
                assert len(inter["x_inter"]) >= 2
    got, _ = DDIMSampler(m).sample(S=4, batch_size=3, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T)
    np.testing.assert_allclose(got.cpu().numpy(), g["ddim_uniform4_plain"], rtol=2e-5, atol=2e-5)
    with pytest.raises(ValueError):
        PLMSSampler(m).sample(S=4, batch_size=3, shape=[4, 8, 8], conditioning=c, eta=0.5, x_T=x_T)
    with pytest.raises(NotImplementedError):
        DDIMSampler(m).sample(S=4, batch_size=3, shape=[4, 8, 8], conditioning=c, x_T=x_T, mask=torch.ones(1))


def test_ddim_step_with_noise_matches_the_oracle_formula():
    from autodiffusion_amd.sd_sampler import sd_step
    from oracle import sd_sampler as S
    g = torch.Generator().manual_seed(3)
    x, eu, ec, noise = (torch.randn(2, 4, 16, 16, generator=g) for _ in range(4))
    ac = S.alphas_cumprod_f32()
    sig, a, a_prev = S.sampling_parameters(ac, [100, 400, 900], eta=0.7)
    e = eu + 5.0 * (ec - eu)
    want_xp, want_x0 = S._update(x, e, a[1], a_prev[1], sig[1], noise)
    xp, x0, e_out = sd_step(x.to(DEV), torch.cat([eu, ec]).to(DEV), 2, 5.0, (1.0,), (), a[1].item(), a_prev[1].item(),
                            sig[1].item(), noise.to(DEV))
    np.testing.assert_allclose(e_out.cpu().numpy(), e.numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x0.cpu().numpy(), want_x0.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(xp.cpu().numpy(), want_xp.numpy(), rtol=2e-5, atol=2e-5)


def test_sd_search_path_end_to_end_tiny_unet_against_the_oracle():
    """sampler.sample(..., sampled_timestep=cand) over the HIP latent UNet with classifier-free guidance, against the
    oracle sampler over the oracle UNet (both golden-pinned); bf16 torso tolerance as above."""
    from autodiffusion_amd.sd_sampler import LatentDiffusion, PLMSSampler
    from oracle import sd_nets, sd_sampler as S
    g, plan, P = sd_case("sd_unet_tiny")
    unet = _model(plan, P)
    ld = LatentDiffusion(unet, device=DEV)
    x_T, ctx = torch.from_numpy(g["x"]), torch.from_numpy(g["context"])
    uc = ctx.flip(1).contiguous() * 0.5
    cand = [153, 424, 690, 926]
    got, _ = PLMSSampler(ld).sample(S=4, batch_size=2, shape=[4, 16, 16], conditioning=ctx.to(DEV), verbose=False,
                                    x_T=x_T.to(DEV), unconditional_guidance_scale=3.0,
                                    unconditional_conditioning=uc.to(DEV), sampled_timestep=np.array(cand))
    ref = S.plms_sample(lambda x, t, c: sd_nets.sd_unet_forward(P, plan, x, t, c), S.alphas_cumprod_f32(), x_T, ctx, cand,
                        uc=uc, scale=3.0)
    check(got, ref.numpy(), "PLMS 4-step, tiny latent UNet, cfg 3.0", 3e-2, 8e-2)


def test_dpm_solver_sampler_matches_reference_goldens():
    from autodiffusion_amd.sd_sampler import DPMSolverSampler
    g = golden("sd_samplers")
    x_T, c, uc = (torch.from_numpy(g[k]).to(DEV) for k in ("x_T", "c", "uc"))
    m = _ToyLatentModel()
    for tag in ("i4", "i6", "f4", "i2"):
        cand = g[f"dpmcand_{tag}"].tolist()
        cand = [int(v) for v in cand] if max(cand) > 1 else cand
        for gtag, (scale, u) in {"cfg": (7.5, uc), "plain": (1.0, None)}.items():
            m.calls = 0
            got, _ = DPMSolverSampler(m).sample(S=len(cand) - 1, batch_size=3, shape=[4, 8, 8], conditioning=c, verbose=False,
                                                x_T=x_T, unconditional_guidance_scale=scale, unconditional_conditioning=u,
                                                sampled_timestep=cand)
            assert m.calls == len(cand) - 1  # the final model value is never evaluated (dpm_solver.py:1116-1118)
            np.testing.assert_allclose(got.cpu().numpy(), g[f"dpm_{tag}_{gtag}"], rtol=1e-4, atol=1e-4, err_msg=f"{tag} {gtag}")
    # without a searched list: the uniform time grid, against the oracle
    from oracle import sd_sampler as S
    got, _ = DPMSolverSampler(m).sample(S=5, batch_size=3, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T)
    tp = [float(v) for v in np.linspace(1.0, 0.001, 6, dtype=np.float32)]
    ref = S.dpm_sample(S.toy_model, S.alphas_cumprod_f32(), x_T.cpu(), c.cpu(), tp)
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=1e-4, atol=1e-4)


def test_same_shape_evaluations_from_two_streams_use_disjoint_graphs():
    """Round 1 recorded a GPU memory-access fault when two captured evaluations were replayed concurrently on two streams:
    one graph entry (one hipGraphExec + one set of static buffers) was keyed by shape alone, so both streams replayed the
    same exec and raced its buffers.  Entries are keyed by the launching stream now: two same-shape batches evaluated
    from two streams own two graphs and equal the eager results bit for bit."""
    g, plan, P = sd_case("sd_unet_tiny")
    m = _model(plan, P)
    x, t, ctx = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "context"))
    xa, xb = x, (x * 0.5 + 0.1).contiguous()
    want_a, want_b = m(xa, t, ctx), m(xb, t, ctx)
    m.enable_graph()
    sa, sb = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    cur = torch.cuda.current_stream()
    outs = {}
    for rounds in range(3):  # first round captures (one graph per stream), the next two replay concurrently
        for name, s_, xx in (("a", sa, xa), ("b", sb, xb)):
            s_.wait_stream(cur)
            with torch.cuda.stream(s_):
                outs[name] = m(xx, t, ctx, context_key="k")
        cur.wait_stream(sa)
        cur.wait_stream(sb)
        torch.cuda.synchronize()
        assert torch.equal(outs["a"], want_a) and torch.equal(outs["b"], want_b), rounds
    assert len(m._packed.graphs) == 2


def test_split_stream_guidance_is_bit_identical():
    """Classifier-free guidance as two evaluations of N latents on two HIP streams (the default) against the reference's
    single batch of 2N (ddim.py:177-181), eager and with per-stream hipGraph replay, for the three samplers."""
    from autodiffusion_amd.sd_sampler import DDIMSampler, DPMSolverSampler, LatentDiffusion, PLMSSampler, _LatentSampler
    assert _LatentSampler.split_guidance is True
    g, plan, P = sd_case("sd_unet_tiny")
    unet = _model(plan, P)
    ld = LatentDiffusion(unet, device=DEV)
    x_T, ctx = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["context"]).to(DEV)
    uc = ctx.flip(1).contiguous() * 0.5
    for cls, cand in ((DDIMSampler, [153, 424, 690, 926]), (PLMSSampler, [153, 424, 690, 926]), (DPMSolverSampler, [999, 600, 300, 50, 1])):
        outs = []
        for split, graph in ((True, False), (False, False), (True, True), (False, True)):
            unet.enable_graph(graph)
            s = cls(ld)
            s.split_guidance = split
            out, _ = s.sample(S=len(cand) - (cls is DPMSolverSampler), batch_size=2, shape=[4, 16, 16], conditioning=ctx,
                              verbose=False, x_T=x_T, unconditional_guidance_scale=3.0, unconditional_conditioning=uc,
                              sampled_timestep=(cand if cls is DPMSolverSampler else np.array(cand)))
            torch.cuda.synchronize()
            outs.append(out.clone())
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), cls.__name__
    unet.enable_graph(False)
