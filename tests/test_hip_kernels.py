"""GPU parity tests, kernel by kernel, through the C ABI (ctypes -> libadm_hip.so).

Each HIP kernel is compared with the CPU oracle (or a plain PyTorch-CPU fp32
statement of the same op) on identical seeded inputs.  Tolerances:
  * fp32 elementwise kernels (sampler, embeddings): 1e-5 relative -- they compute
    in fp32 exactly as the reference does;
  * bf16 MFMA kernels: inputs / weights are rounded to bf16 on BOTH sides, the
    accumulation is fp32, so the remaining difference is summation order plus the
    final bf16 rounding of the output: |err| <= 1e-2 * max|ref| elementwise and
    <= 4e-3 relative in Frobenius norm.
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd import ops as _ops
    return _ops


DEV = "cuda:0"


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def nhwc_dev(x):  # fp32 NCHW cpu -> bf16 NHWC device
    return x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def nchw_cpu(y):  # bf16 NHWC device -> fp32 NCHW cpu
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def assert_close_bf16(got, ref, what=""):
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item()
    fro = ((got - ref).norm() / (ref.norm() + 1e-12)).item()
    assert err <= 1e-2 * scale and fro <= 4e-3, f"{what}: max err {err:.4g} (scale {scale:.4g}), fro {fro:.4g}"


# ------------------------------------------------------------------ sampler (K9)
def _coefs(d, i, eta=0.0, clip=True):
    from autodiffusion_amd.sampler import step_coefs
    return step_coefs(d.tables, i, learned_range=(d.var_type == "learned_range"),
                      fixed=("large" if d.var_type == "fixed_large" else "small"),
                      predict_xstart=d.predict_xstart, clip_denoised=clip, eta=eta)


@pytest.mark.parametrize("learn_sigma", [True, False])
def test_sampler_steps_match_oracle(ops, learn_sigma):
    from oracle import sampler as osm, schedule
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=learn_sigma).reset([153, 424, 926, 690])
    n, c, h, w = 3, 3, 16, 16
    x, g, nz = rnd((n, c, h, w), 1), rnd((n, c, h, w), 2, 0.3), rnd((n, c, h, w), 3)
    mo = rnd((n, 2 * c if learn_sigma else c, h, w), 4)
    for i in (3, 1, 0):
        for grad in (None, g):
            for eta in (0.0, 0.7):
                ref = osm.ddim_step(d, mo, x, i, grad, nz, eta)
                xp, x0, _ = ops.sampler_step("ddim", x.to(DEV), mo.to(DEV), _coefs(d, i, eta),
                                             None if grad is None else grad.to(DEV), nz.to(DEV), want_xstart=True)
                torch.testing.assert_close(xp.cpu(), ref["sample"], rtol=1e-5, atol=1e-5)
                torch.testing.assert_close(x0.cpu(), ref["pred_xstart"], rtol=1e-5, atol=1e-5)
            ref = osm.ddpm_step(d, mo, x, i, grad, nz)
            xp, x0, u8 = ops.sampler_step("ddpm", x.to(DEV), mo.to(DEV), _coefs(d, i),
                                          None if grad is None else grad.to(DEV), nz.to(DEV),
                                          want_xstart=True, want_u8=True)
            torch.testing.assert_close(xp.cpu(), ref["sample"], rtol=1e-5, atol=1e-5)
            torch.testing.assert_close(x0.cpu(), ref["pred_xstart"], rtol=1e-5, atol=1e-5)
            assert torch.equal(u8.cpu(), osm.pack_uint8_nhwc(xp.cpu()))


def test_pack_u8_bit_exact_and_odd_sizes(ops):
    from oracle import sampler as osm
    for shape in ((2, 3, 8, 8), (1, 3, 5, 7), (3, 1, 64, 64)):
        x = rnd(shape, 5, 1.2)
        x.view(-1)[:4] = torch.tensor([-1.0, 1.0, 0.99999, -3.0])
        assert torch.equal(ops.pack_u8_nhwc(x.to(DEV)).cpu(), osm.pack_uint8_nhwc(x))


def test_step_requires_noise_when_used(ops):
    from oracle import schedule
    from autodiffusion_amd._lib import AdmError
    d = schedule.OracleDiffusion(steps=1000, noise_schedule="linear", learn_sigma=True).reset([10, 500])
    x = rnd((1, 3, 8, 8), 1).to(DEV)
    mo = rnd((1, 6, 8, 8), 2).to(DEV)
    with pytest.raises(AdmError):
        ops.sampler_step("ddpm", x, mo, _coefs(d, 1), None, None)
    ops.sampler_step("ddpm", x, mo, _coefs(d, 0), None, None)  # i == 0 adds no noise


# ------------------------------------------------------------------ embeddings (K1)
def test_timestep_embedding(ops):
    from oracle import nets
    t = torch.tensor([0, 1, 250, 999, 37])
    for dim in (192, 32, 33, 128):
        got = ops.timestep_embedding(t.to(DEV), dim).cpu()
        torch.testing.assert_close(got, nets.sinusoid_embedding(t, dim), rtol=0, atol=2e-4)


def _linear_path(n, k):
    """Which of adm_linear_f32's three kernels serves n rows (csrc/adm_embed.hip): matrix-pipe (chosen by k alone, so that a
    row's bits never depend on the batch it rides in), GEMV-shaped, 64x64 tile."""
    if k % 16 == 0:
        return "mfma"
    return "small" if k % 4 == 0 else "tile"


@pytest.mark.parametrize("n,k,o", [(2, 128, 64), (5, 768, 1000), (256, 768, 1536), (3, 32, 128), (12, 1280, 21120), (48, 1280, 333),
                                   (17, 320, 1280), (1, 4, 1), (65, 128, 96), (7, 30, 50), (70, 48, 50), (130, 16, 33),
                                   (300, 64, 40), (129, 30, 20), (200, 1000, 512), (66, 20, 24)])
def test_linear_f32(ops, n, k, o):
    x, w, b = rnd((n, k), 1), rnd((o, k), 2, k ** -0.5), rnd((o,), 3, 0.1)
    tab, idx = rnd((10, o), 4), torch.randint(0, 10, (n,), generator=torch.Generator().manual_seed(5))
    got = ops.linear_f32(x.to(DEV), w.to(DEV), b.to(DEV), silu_in=True, table=tab.to(DEV), idx=idx.to(DEV)).cpu()
    ref = F.linear(F.silu(x), w, b) + tab[idx]
    torch.testing.assert_close(got, ref, rtol=1e-4, atol=1e-4)
    got = ops.linear_f32(x.to(DEV), w.to(DEV), None).cpu()
    torch.testing.assert_close(got, F.linear(x, w), rtol=1e-4, atol=1e-4)
    if n >= 4 and _linear_path(n, k) == _linear_path(n - n // 2, k):  # a row's result does not depend on where it sits in the batch (same kernel)
        assert torch.equal(ops.linear_f32(x[n // 2:].contiguous().to(DEV), w.to(DEV), None).cpu(), got[n // 2:])
    if k % 4 == 0 and n >= 2:   # ... nor on how many rows ride along: 2 rows alone == the first 2 of n (the kernel is chosen by k only)
        assert torch.equal(ops.linear_f32(x[:2].contiguous().to(DEV), w.to(DEV), None).cpu(), got[:2])


# ------------------------------------------------------------------ stem / GN / resample
def test_stem_conv(ops):
    x, w, b = rnd((3, 3, 32, 32), 1), rnd((64, 3, 3, 3), 2, 0.2), rnd((64,), 3, 0.1)
    got = nchw_cpu(ops.stem_conv3x3(x.to(DEV), w.to(DEV), b.to(DEV)))
    assert_close_bf16(got, F.conv2d(x, w, b, padding=1), "stem")


@pytest.mark.parametrize("c0,c1,hw", [(64, 0, 8), (192, 0, 64), (768, 576, 8), (32, 32, 16), (1536, 0, 8)])
def test_gn_affine_matches_group_norm(ops, c0, c1, hw):
    n = 3
    c = c0 + c1
    x = bf(rnd((n, c, hw, hw), 1, 1.5) + 0.3)
    gamma, beta = 1 + 0.2 * rnd((c,), 2), 0.1 * rnd((c,), 3)
    film = rnd((n, 2 * c + 5), 4, 0.3)  # wider row: exercises film_stride
    x0 = nhwc_dev(x[:, :c0])
    x1 = nhwc_dev(x[:, c0:]) if c1 else None
    filmd = film.to(DEV)
    a, b = ops.gn_affine(x0, gamma.to(DEV), beta.to(DEV), x1, film=filmd, film_stride=film.shape[1])
    got = a.cpu()[:, :, None, None] * x + b.cpu()[:, :, None, None]
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5) * (1 + film[:, :c, None, None]) + film[:, c:2 * c, None, None]
    torch.testing.assert_close(got, ref, rtol=2e-4, atol=2e-4)
    a, b = ops.gn_affine(x0, gamma.to(DEV), beta.to(DEV), x1)
    got = a.cpu()[:, :, None, None] * x + b.cpu()[:, :, None, None]
    torch.testing.assert_close(got, F.group_norm(x, 32, gamma, beta, eps=1e-5), rtol=2e-4, atol=2e-4)


def test_resample(ops):
    x = bf(rnd((2, 64, 16, 16), 1))
    a, b = 1 + 0.1 * rnd((2, 64), 2), 0.1 * rnd((2, 64), 3)
    act = F.silu(a[:, :, None, None] * x + b[:, :, None, None])
    xd = nhwc_dev(x)
    aff = (a.to(DEV), b.to(DEV))
    assert_close_bf16(nchw_cpu(ops.resample(xd, "down")), F.avg_pool2d(x, 2), "down raw")
    assert_close_bf16(nchw_cpu(ops.resample(xd, "up")), F.interpolate(x, scale_factor=2, mode="nearest"), "up raw")
    assert_close_bf16(nchw_cpu(ops.resample(xd, "down", aff)), F.avg_pool2d(act, 2), "down act")
    assert_close_bf16(nchw_cpu(ops.resample(xd, "up", aff)), F.interpolate(act, scale_factor=2, mode="nearest"), "up act")


# ------------------------------------------------------------------ conv (K2/K3/K4/K7/K8)
CONV_CASES = [
    # n, hw, c0, c1, cout, taps, prologue(0 raw,1 affine,2 affine+silu), res, f32_nchw, variant
    (2, 16, 32, 0, 64, 9, 2, False, False, 0),
    (3, 8, 64, 0, 96, 9, 2, True, False, 0),     # TI=4 with a ragged last tile (3 images)
    (1, 32, 64, 32, 128, 9, 2, True, False, 0),  # virtual concat
    (2, 64, 32, 0, 6, 9, 2, False, True, 0),     # output head, fp32 NCHW
    (2, 16, 192, 0, 192, 9, 0, False, False, 0),
    (2, 16, 64, 0, 192, 1, 1, False, False, 0),  # qkv projection (GN, no SiLU)
    (2, 8, 96, 0, 96, 1, 0, True, False, 0),     # proj_out + residual
    (5, 8, 64, 64, 64, 1, 0, False, False, 0),   # skip 1x1 on a concat
    (1, 16, 64, 0, 256, 9, 2, False, False, 5),   # Cout padded to two 192-wide blocks
    (1, 16, 64, 0, 64, 9, 2, False, False, 6),    # Cout below one 128-wide block
    (2, 32, 96, 0, 16, 9, 2, False, False, 3),
    (2, 16, 64, 0, 192, 9, 2, True, False, 5),
    (2, 16, 64, 0, 128, 9, 2, True, False, 6),
    (2, 16, 64, 0, 384, 1, 1, False, False, 0),   # resident-tile 1x1 kernel (auto: Cout >= 256), 128-pixel tiles
    (3, 8, 128, 0, 256, 1, 0, True, False, 0),    # ... 64-pixel tiles (8x8 maps), ragged second half of the Cout block
    (2, 32, 64, 64, 448, 1, 2, True, False, 10),  # ... virtual concat, SiLU prologue, two Cout blocks (384 + 64)
    (2, 8, 768, 0, 2304, 1, 1, False, False, 0),  # ... ADM-64's 8x8 qkv: 64-pixel tiles, 6 Cout blocks, wave-private epilogue
    (1, 16, 576, 0, 1728, 1, 1, False, False, 0),  # ... 16x16 qkv: K = 576 (64-pixel tiles by LDS size), ragged last Cout block
    (1, 16, 192, 0, 192, 1, 1, True, False, 10),  # ... narrow output: 2 x 4 wave layout, block epilogue (residual)
    (3, 16, 128, 64, 128, 1, 0, False, False, 10),  # ... narrow output, wave-private epilogue, concat, idle 4th column
    (6, 16, 1280, 0, 1280, 1, 0, True, False, 0),   # SD v1's 1280-wide projections at 16x16: staged kernel (K too deep for the resident tile), 128-pixel tiles
    (2, 16, 1024, 256, 320, 1, 1, False, False, 0),  # ... concat (K = 1280), affine prologue, ragged last Cout block
    (2, 16, 64, 0, 192, 9, 2, True, False, 7),    # 32x32x16 MFMA kernel
    (1, 32, 64, 32, 384, 9, 0, False, False, 7),  # ... raw prologue, virtual concat, two channel blocks
    (3, 64, 32, 0, 96, 9, 1, True, False, 7),     # ... Cout below one tile: padded columns
    (3, 8, 160, 0, 96, 1, 1, False, False, 0),    # 128-pixel 1x1 tiles (8x8 maps), 5 chunks: the no-exit loop runs 3 K-steps past the last (chunks % 4 == 1)
    (2, 8, 224, 0, 64, 1, 0, True, False, 0),     # ... 7 chunks (chunks % 4 == 3), raw prologue, residual
    (6, 16, 1312, 0, 320, 1, 0, False, False, 0),  # ... the deep staged 1x1 (16x16, K = 41 chunks: % 8 == 1 on the 8-deep weight ring)
    (3, 16, 1280, 0, 640, 1, 1, False, False, 10),  # resident-tile kernel on 32-PIXEL tiles (K >= 1280 at 16x16 / 8x8; explicit: variant 10): GroupNorm-affine prologue, ragged last Cout block, wave-private epilogue
    (5, 8, 1280, 0, 1280, 1, 0, True, False, 10),   # ... 8x8 maps (two tiles per image), residual: the block epilogue
    (2, 16, 1024, 256, 1408, 1, 2, False, False, 10),  # ... virtual concat (K = 1280), SiLU prologue, 4 Cout blocks (the last ragged)
    (2, 16, 1280, 0, 1920, 1, 0, False, False, 10),   # ... 5 Cout blocks
]


def test_conv1x1_pad_steps_read_zero_activations(ops):
    """The no-exit 1x1 loops run up to 3 (7) K-steps past the tile's last chunk on zero weight fragments; the activation side of
    those steps must be zeros too: an Inf in the LAST chunk (an overflowed fp16 torso) gives +-Inf at that pixel, as the exact
    loop does -- not 0 x Inf = NaN from re-reading it."""
    n, hw, cin, cout = 2, 8, 160, 96
    x = bf(rnd((n, cin, hw, hw), 1))
    x[1, cin - 3, 2, 5] = float("inf")          # last 32-channel chunk
    w = rnd((cout, cin, 1, 1), 2, cin ** -0.5)
    got = nchw_cpu(ops.conv(nhwc_dev(x), ops.pack_conv_weight(w.to(DEV)), torch.zeros(cout, device=DEV), cout, 1))
    hit = got[1, :, 2, 5]
    assert torch.isinf(hit).all() and not torch.isnan(got).any(), (hit, torch.isnan(got).sum())
    assert torch.equal(torch.sign(hit), torch.sign(bf(w)[:, cin - 3, 0, 0]))
    got[1, :, 2, 5] = 0
    assert torch.isfinite(got).all()


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fused(ops, case):
    n, hw, c0, c1, cout, taps, prologue, use_res, f32, variant = case
    cin = c0 + c1
    k = 3 if taps == 9 else 1
    x = bf(rnd((n, cin, hw, hw), 1))
    w = rnd((cout, cin, k, k), 2, (cin * taps) ** -0.5)
    b = rnd((cout,), 3, 0.1)
    a_, b_ = 1 + 0.2 * rnd((n, cin), 4), 0.2 * rnd((n, cin), 5)
    res = bf(rnd((n, cout, hw, hw), 6))
    h = x
    if prologue:
        h = a_[:, :, None, None] * x + b_[:, :, None, None]
        if prologue == 2:
            h = F.silu(h)
    ref = F.conv2d(bf(h), bf(w), b, padding=k // 2)
    if use_res:
        ref = ref + res
    wp = ops.pack_conv_weight(w.to(DEV))
    wp32 = ops.pack_conv_weight32(w.to(DEV)) if variant == 7 else None
    got = ops.conv(nhwc_dev(x[:, :c0]), wp, b.to(DEV), cout, taps, w_packed32=wp32,
                   x1=nhwc_dev(x[:, c0:]) if c1 else None,
                   aff=(a_.to(DEV), b_.to(DEV)) if prologue else None, silu=(prologue == 2),
                   res=nhwc_dev(res) if use_res else None, out_f32_nchw=f32, variant=variant)
    got = got.cpu() if f32 else nchw_cpu(got)
    assert_close_bf16(got, ref, f"conv {case}")


@pytest.mark.parametrize("n,hs,cin,cout,prologue,in_up,res_up", [(2, 8, 64, 192, 2, True, False), (3, 16, 32, 128, 2, False, True),
                                                              (1, 32, 64, 192, 1, True, True), (2, 8, 96, 64, 0, True, True)])
def test_conv_virtual_upsample(ops, n, hs, cin, cout, prologue, in_up, res_up):
    """ResBlock(up=True) (unet.py:236-249): conv(Upsample(act(GN(x)))) + Upsample(x) with the half-resolution tensors
    read through a virtual nearest-neighbour 2x upsample inside the conv."""
    h2 = 2 * hs
    x = bf(rnd((n, cin, hs if in_up else h2, hs if in_up else h2), 1))
    w = rnd((cout, cin, 3, 3), 2, (cin * 9) ** -0.5)
    b = rnd((cout,), 3, 0.1)
    a_, b_ = 1 + 0.2 * rnd((n, cin), 4), 0.2 * rnd((n, cin), 5)
    res = bf(rnd((n, cout, hs if res_up else h2, hs if res_up else h2), 6))
    h = x
    if prologue:
        h = a_[:, :, None, None] * x + b_[:, :, None, None]
        if prologue == 2:
            h = F.silu(h)
    h = bf(h)
    if in_up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
    ref = F.conv2d(h, bf(w), b, padding=1) + (F.interpolate(res, scale_factor=2, mode="nearest") if res_up else res)
    got = ops.conv(nhwc_dev(x), ops.pack_conv_weight(w.to(DEV)), b.to(DEV), cout, 9,
                   aff=(a_.to(DEV), b_.to(DEV)) if prologue else None, silu=(prologue == 2),
                   res=nhwc_dev(res), in_up=in_up, res_up=res_up, want_stats=True)
    assert got.shape == (n, h2, h2, cout)
    assert_close_bf16(nchw_cpu(got), ref, f"virtual upsample {n, hs, cin, cout, prologue, in_up, res_up}")


@pytest.mark.parametrize("n,hs,ws,cin,cout,prologue", [(3, 16, 16, 192, 192, 2), (2, 32, 32, 384, 384, 2), (2, 16, 32, 256, 256, 2),
                                                       (5, 16, 16, 64, 320, 0), (1, 32, 16, 128, 136, 1)])
def test_conv_upsample_as_four_phase_convs(ops, n, hs, ws, cin, cout, prologue):
    """conv3x3(nearest-upsample-2x(act(GN(x)))) as four 2x2-tap phase launches (adm_conv_args.up_phase, ops.pack_conv_weight_up):
    against the PyTorch-CPU reference of the op and against the one-launch virtual-upsample path, incl. the fused output
    statistics (per-image sums over all slabs) that the next GroupNorm consumes."""
    x = bf(rnd((n, cin, hs, ws), 1))
    w = rnd((cout, cin, 3, 3), 2, (cin * 9) ** -0.5)
    b = rnd((cout,), 3, 0.1)
    a_, b_ = 1 + 0.2 * rnd((n, cin), 4), 0.2 * rnd((n, cin), 5)
    h = x
    if prologue:
        h = a_[:, :, None, None] * x + b_[:, :, None, None]
        if prologue == 2:
            h = F.silu(h)
    ref = F.conv2d(F.interpolate(bf(h), scale_factor=2, mode="nearest"), bf(w), b, padding=1)
    aff = (a_.to(DEV), b_.to(DEV)) if prologue else None
    wd = w.to(DEV)
    one = ops.conv(nhwc_dev(x), ops.pack_conv_weight(wd), b.to(DEV), cout, 9, aff=aff, silu=(prologue == 2), in_up=True, want_stats=True)
    four = ops.conv(nhwc_dev(x), ops.pack_conv_weight(wd), b.to(DEV), cout, 9, aff=aff, silu=(prologue == 2), in_up=True, want_stats=True,
                    w_up=ops.pack_conv_weight_up(wd))
    assert four.shape == (n, 2 * hs, 2 * ws, cout) and ops.UPCONV_PHASES
    # pre-summed taps are rounded to bf16 once (w1 + w2) instead of twice: the same size of error as the reference's own rounding
    got = nchw_cpu(four)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-2 * scale and ((got - ref).norm() / ref.norm()).item() <= 6e-3
    assert ((got - nchw_cpu(one)).norm() / ref.norm()).item() <= 6e-3
    # fused statistics: 4 x as many slabs, each phase launch fills its quarter; totals = sums over the stored output
    st4, slabs4 = four._adm_stats
    assert slabs4 == 4 * (hs * ws // 256) and st4.shape == (n, slabs4, cout, 2)
    tot = st4.sum(1).cpu()
    f32o = four.float().cpu()
    torch.testing.assert_close(tot[..., 0], f32o.sum((1, 2)), rtol=2e-3, atol=2e-2)
    torch.testing.assert_close(tot[..., 1], (f32o * f32o).sum((1, 2)), rtol=2e-3, atol=2e-2)


def test_conv_random_shapes(ops):
    """Seeded sweep over shapes the fixed cases do not list: batch sizes that leave ragged tiles, virtual concat splits,
    Cout that pads the 192/128-wide blocks, 3x3 and 1x1 (staged and resident-tile kernels), every prologue, residual and
    fused statistics on or off -- each against the PyTorch-CPU fp32 reference of the same op."""
    rng = np.random.RandomState(1234)
    for it in range(24):
        taps = int(rng.choice([1, 9]))
        hw = int(rng.choice([8, 16, 32]))
        n = int(rng.randint(1, 6))
        c0 = 32 * int(rng.randint(1, 7))
        c1 = 32 * int(rng.randint(0, 4)) if rng.rand() < 0.4 else 0
        cout = 8 * int(rng.randint(1, 60))
        prologue = int(rng.randint(0, 3))
        use_res = bool(rng.rand() < 0.5)
        want_stats = bool(rng.rand() < 0.5) and hw >= 8
        cin = c0 + c1
        k = 3 if taps == 9 else 1
        x = bf(rnd((n, cin, hw, hw), 100 + it))
        w = rnd((cout, cin, k, k), 200 + it, (cin * taps) ** -0.5)
        b = rnd((cout,), 300 + it, 0.1)
        a_, b_ = 1 + 0.2 * rnd((n, cin), 400 + it), 0.2 * rnd((n, cin), 500 + it)
        res = bf(rnd((n, cout, hw, hw), 600 + it))
        h = x
        if prologue:
            h = a_[:, :, None, None] * x + b_[:, :, None, None]
            if prologue == 2:
                h = F.silu(h)
        ref = F.conv2d(bf(h), bf(w), b, padding=k // 2) + (res if use_res else 0)
        got = ops.conv(nhwc_dev(x[:, :c0]), ops.pack_conv_weight(w.to(DEV)), b.to(DEV), cout, taps,
                       x1=nhwc_dev(x[:, c0:]) if c1 else None,
                       aff=(a_.to(DEV), b_.to(DEV)) if prologue else None, silu=(prologue == 2),
                       res=nhwc_dev(res) if use_res else None, want_stats=want_stats)
        case = (it, n, hw, c0, c1, cout, taps, prologue, use_res, want_stats)
        assert_close_bf16(nchw_cpu(got), ref, f"conv random {case}")
        st = getattr(got, "_adm_stats", None)
        if st is not None:  # fused per-(image, channel) sums of the stored output
            part = st[0].sum(dim=1).cpu()
            yv = got.float().cpu()
            torch.testing.assert_close(part[..., 0], yv.sum(dim=(1, 2)), rtol=2e-3, atol=2e-2)
            torch.testing.assert_close(part[..., 1], (yv * yv).sum(dim=(1, 2)), rtol=2e-3, atol=2e-2)


def test_conv_rejects_bad_shapes(ops):
    from autodiffusion_amd._lib import AdmError
    x = torch.zeros((1, 4, 4, 32), dtype=torch.bfloat16, device=DEV)
    wp = ops.pack_conv_weight(torch.zeros((32, 32, 3, 3), device=DEV))
    with pytest.raises(AdmError):
        ops.conv(x, wp, torch.zeros(32, device=DEV), 32, 9)  # 4x4 map: below the halo tile
    with pytest.raises(AdmError):
        ops.pack_conv_weight(torch.zeros((32, 24, 3, 3), device=DEV))  # cin % 32 != 0


# ------------------------------------------------------------------ attention (K5)
@pytest.mark.parametrize("n,heads,d,t", [(2, 2, 32, 64), (2, 1, 64, 256), (1, 6, 64, 1024), (3, 2, 32, 16),
                                         (1, 2, 64, 80), (2, 4, 128, 256),
                                         (1, 2, 192, 100), (2, 4, 256, 64), (1, 4, 192, 256)])  # ADM-128's wide heads
@pytest.mark.parametrize("new_order", [True, False])
def test_attention_matches_oracle(ops, n, heads, d, t, new_order):
    from oracle import nets
    qkv = bf(rnd((n, 3 * heads * d, t), 7))
    ref = nets.qkv_attention(qkv, heads, new_order)  # [n, H*D, t]
    got = ops.attention(qkv.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV), heads, new_order)
    got = got.float().cpu().permute(0, 2, 1)
    assert_close_bf16(got, ref, f"attention {n, heads, d, t, new_order}")


def test_attention_softmax_is_stable_for_large_logits(ops):
    from oracle import nets
    qkv = bf(rnd((1, 3 * 64, 128), 8, 6.0))  # logits ~ +-100: needs the running-max rescale
    ref = nets.qkv_attention(qkv, 1, True)
    got = ops.attention(qkv.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV), 1, True)
    got = got.float().cpu().permute(0, 2, 1)
    assert torch.isfinite(got).all()
    assert_close_bf16(got, ref, "attention large logits")


@pytest.mark.parametrize("n,hw,cin,cout,taps,variant", [(3, 16, 64, 192, 9, 0), (2, 32, 32, 96, 9, 0), (5, 8, 64, 192, 9, 0),
                                                         (2, 64, 32, 64, 1, 0), (2, 16, 32, 128, 9, 6), (3, 8, 64, 128, 1, 0),
                                                         (2, 16, 64, 384, 1, 0), (3, 8, 128, 256, 1, 0), (2, 16, 64, 192, 1, 10),  # resident-tile 1x1
                                                         (3, 16, 1280, 320, 1, 0),     # deep 1x1 at 16x16: the staged kernel's 128-pixel tiles, four slabs per image
                                                         (3, 16, 1280, 320, 1, 10),    # ... resident 32-pixel tiles (explicit), eight slabs per image
                                                         (2, 8, 1280, 384, 1, 10)])    # ... 8x8 maps: two 32-pixel tiles per image
def test_conv_fused_output_statistics_feed_groupnorm(ops, n, hw, cin, cout, taps, variant):
    """The sums accumulated in the conv epilogue must give the same GroupNorm affine as a separate pass."""
    k = 3 if taps == 9 else 1
    x = nhwc_dev(bf(rnd((n, cin, hw, hw), 1)))
    wp = ops.pack_conv_weight((rnd((cout, cin, k, k), 2, (cin * taps) ** -0.5)).to(DEV))
    res = nhwc_dev(bf(rnd((n, cout, hw, hw), 3)))
    y = ops.conv(x, wp, (0.1 * rnd((cout,), 4)).to(DEV), cout, taps, res=res, variant=variant, want_stats=True)
    assert getattr(y, "_adm_stats", None) is not None
    gamma, beta = (1 + 0.2 * rnd((cout,), 5)).to(DEV), (0.1 * rnd((cout,), 6)).to(DEV)
    a1, b1 = ops.gn_affine(y, gamma, beta)                       # fused statistics
    ops.USE_FUSED_STATS = False
    try:
        a2, b2 = ops.gn_affine(y, gamma, beta)                   # separate adm_gn_partial pass
    finally:
        ops.USE_FUSED_STATS = True
    torch.testing.assert_close(a1, a2, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(b1, b2, rtol=2e-4, atol=2e-5)
    # virtual concat of two tensors that both carry fused statistics
    y2 = ops.conv(x, wp, torch.zeros(cout, device=DEV), cout, taps, variant=variant, want_stats=True)
    g2, be2 = torch.cat([gamma, gamma]), torch.cat([beta, beta])
    a1, b1 = ops.gn_affine(y, g2, be2, y2)
    ops.USE_FUSED_STATS = False
    try:
        a2, b2 = ops.gn_affine(y, g2, be2, y2)
    finally:
        ops.USE_FUSED_STATS = True
    torch.testing.assert_close(a1, a2, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(b1, b2, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("n,hw,c0,c1,cout,ks,pro,use_res", [
    (3, 8, 256, 0, 192, 4, 2, True),        # 8x8 level, 128-pixel tiles (two images per tile), ragged last tile
    (2, 16, 128, 128, 128, 2, 2, False),    # virtual concat: the split boundary is the boundary between the two sources
    (5, 8, 512, 0, 320, 4, 0, True),        # 320 output channels: the last 128-wide Cout block is ragged
    (1, 16, 64, 192, 256, 2, 1, True),      # the split falls inside the second source
])
def test_conv_split_k_matches_the_one_pass_conv(ops, n, hw, c0, c1, cout, ks, pro, use_res):
    """adm_conv with ksplit > 1 (K loop of every tile cut into runs that go out as separate tiles, fp32 partial sums,
    deterministic reduce with bias / residual / output statistics) against fp32 torch and against the one-pass kernel."""
    _split_k_case(ops, 9, n, hw, c0, c1, cout, ks, pro, use_res)


@pytest.mark.parametrize("n,hw,c0,c1,cout,ks,pro,use_res", [
    (6, 16, 1280, 0, 1280, 4, 0, True),     # SD v1: attention / feed-forward output projections at 16x16 (runs of 10 chunks)
    (3, 8, 1024, 0, 320, 8, 0, True),       # 8x8 level, 128-pixel tiles (two images per tile), ragged last tile and Cout block; runs of 4 chunks
    (2, 16, 512, 0, 192, 4, 1, False),      # GroupNorm prologue (the transformer's proj_in); the staged kernel serves cout < 256 without the split too
    (2, 16, 256, 256, 384, 2, 2, True),     # virtual concat, split at the source boundary, GN + SiLU prologue
    (1, 32, 128, 0, 256, 2, 0, False),      # a 32x32 map (4 tiles per image), runs of 2 chunks
])
def test_conv1x1_split_k_matches_the_one_pass_conv(ops, n, hw, c0, c1, cout, ks, pro, use_res):
    """The same for 1x1 convs (the staged kernel's 1x1 loop over chunks [cb, ce); without the split most of these shapes run on
    the resident-tile kernel): SD v1's 1280-wide projections at 6-latent half batches are 30-60 tiles of 40-160 chunks."""
    _split_k_case(ops, 1, n, hw, c0, c1, cout, ks, pro, use_res)


def _split_k_case(ops, taps, n, hw, c0, c1, cout, ks, pro, use_res):
    cin = c0 + c1
    kk = 3 if taps == 9 else 1
    x = bf(rnd((n, cin, hw, hw), 1))
    w = bf(rnd((cout, cin, kk, kk), 2, (cin * taps) ** -0.5))
    b = 0.1 * rnd((cout,), 3)
    res = bf(rnd((n, cout, hw, hw), 4)) if use_res else None
    a = 1 + 0.1 * rnd((n, cin), 5)
    sh = 0.1 * rnd((n, cin), 6)
    xin = x
    if pro:
        xin = a[:, :, None, None] * x + sh[:, :, None, None]
        if pro == 2:
            xin = F.silu(xin)
        xin = bf(xin)
    ref = F.conv2d(xin, w, b, padding=kk // 2)
    if use_res:
        ref = bf(ref) + res
    xd = nhwc_dev(x)
    x0, x1 = (xd[..., :c0].contiguous(), xd[..., c0:].contiguous()) if c1 else (xd, None)
    wp = ops.pack_conv_weight(w.to(DEV))
    kw = dict(x1=x1, aff=(a.to(DEV), sh.to(DEV)) if pro else None, silu=(pro == 2), res=nhwc_dev(res) if use_res else None,
              want_stats=True)
    y1 = ops.conv(x0, wp, b.to(DEV), cout, taps, **kw)
    yk = ops.conv(x0, wp, b.to(DEV), cout, taps, ksplit=ks, **kw)
    assert_close_bf16(nchw_cpu(yk), ref, f"split-K x{ks}")
    d = (nchw_cpu(yk) - nchw_cpu(y1)).abs().max().item()
    assert d <= 2e-2 * ref.abs().max().item(), d          # same values up to the last bf16 digit (fp32 order differs)
    assert torch.equal(ops.conv(x0, wp, b.to(DEV), cout, taps, ksplit=ks, **kw), yk)   # deterministic
    if n > 1:   # the result does not depend on the batch the image rides in
        kw1 = dict(kw, x1=None if x1 is None else x1[:1].contiguous(), aff=None if not pro else (a[:1].to(DEV), sh[:1].to(DEV)),
                   res=None if not use_res else nhwc_dev(res[:1]))
        assert torch.equal(ops.conv(x0[:1].contiguous(), wp, b.to(DEV), cout, taps, ksplit=ks, **kw1), yk[:1])
    # the reduce pass's output statistics feed GroupNorm like the one-pass epilogue's
    gamma, beta = (1 + 0.2 * rnd((cout,), 7)).to(DEV), (0.1 * rnd((cout,), 8)).to(DEV)
    a1, b1 = ops.gn_affine(yk, gamma, beta)
    ops.USE_FUSED_STATS = False
    try:
        a2, b2 = ops.gn_affine(yk, gamma, beta)
    finally:
        ops.USE_FUSED_STATS = True
    torch.testing.assert_close(a1, a2, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(b1, b2, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("n,hw,cin,cout,fc0,fc1", [
    (2, 16, 64, 192, 96, 0),        # one skip source, 3 one-tap K-steps
    (3, 16, 192, 192, 192, 384),    # decoder block: the skip path reads the concat (h | skip) = 18 K-steps, 3x the main path's 6 chunks
    (1, 32, 64, 128, 32, 0),        # a single skip K-step (the ring's clamps), 128-wide tile, 4 tiles per image
    (2, 16, 96, 384, 64, 32),       # two Cout blocks; an odd number of main chunks (the skip path starts in halo buffer 1)
    (5, 8, 128, 192, 96, 160),      # 8x8 maps: 128-pixel tiles of two images, ragged last tile
    (4, 8, 256, 256, 256, 0),       # 8x8, 128-wide tile
])
def test_conv_with_the_skip_connection_folded_in(ops, n, hw, cin, cout, fc0, fc1):
    """adm_conv_args.fold0: out_layers conv3x3(SiLU(GN-affine(h))) + skip_connection conv1x1(x) + both biases in ONE K loop (reference
    unet.py:216-222, 256) against fp32 torch and against the two-launch form (1x1 conv, then 3x3 conv with the residual operand)."""
    fc = fc0 + fc1
    h = bf(rnd((n, cin, hw, hw), 1))
    xs = bf(rnd((n, fc, hw, hw), 2))
    w3 = bf(rnd((cout, cin, 3, 3), 3, (cin * 9) ** -0.5))
    w1 = bf(rnd((cout, fc, 1, 1), 4, fc ** -0.5))
    b3, b1 = 0.1 * rnd((cout,), 5), 0.1 * rnd((cout,), 6)
    a, sh = 1 + 0.1 * rnd((n, cin), 7), 0.1 * rnd((n, cin), 8)
    act = bf(F.silu(a[:, :, None, None] * h + sh[:, :, None, None]))
    ref = F.conv2d(act, w3, b3, padding=1) + F.conv2d(xs, w1, b1)
    hd, xd = nhwc_dev(h), nhwc_dev(xs)
    x0, x1 = (xd[..., :fc0].contiguous(), xd[..., fc0:].contiguous()) if fc1 else (xd, None)
    w3p, w1p = ops.pack_conv_weight(w3.to(DEV)), ops.pack_conv_weight(w1.to(DEV))
    aff = (a.to(DEV), sh.to(DEV))
    got = ops.conv(hd, ops.fold_weights(w3p, w1p), (b3 + b1).to(DEV), cout, 9, aff=aff, silu=True, fold=(x0, x1), want_stats=True)
    assert_close_bf16(nchw_cpu(got), ref, "skip fold")
    res = ops.conv(x0, w1p, b1.to(DEV), cout, 1, x1=x1)
    two = ops.conv(hd, w3p, b3.to(DEV), cout, 9, aff=aff, silu=True, res=res, want_stats=True)
    d = (nchw_cpu(got) - nchw_cpu(two)).abs().max().item()
    assert d <= 2e-2 * ref.abs().max().item(), d
    # closer to fp32 than the two-launch form (which rounds the 1x1 result to 16 bits before the add)
    e1 = ((nchw_cpu(got) - ref).norm() / ref.norm()).item()
    e2 = ((nchw_cpu(two) - ref).norm() / ref.norm()).item()
    print(f"skip fold rel {e1:.3e}, two launches {e2:.3e}")
    assert e1 <= e2 * 1.05
    assert torch.equal(ops.conv(hd, ops.fold_weights(w3p, w1p), (b3 + b1).to(DEV), cout, 9, aff=aff, silu=True, fold=(x0, x1)), got)
    if n > 1:   # batch independence
        g1 = ops.conv(hd[:1].contiguous(), ops.fold_weights(w3p, w1p), (b3 + b1).to(DEV), cout, 9, aff=(a[:1].to(DEV), sh[:1].to(DEV)), silu=True,
                      fold=(x0[:1].contiguous(), None if x1 is None else x1[:1].contiguous()))
        assert torch.equal(g1, got[:1])
    gamma, beta = (1 + 0.2 * rnd((cout,), 9)).to(DEV), (0.1 * rnd((cout,), 10)).to(DEV)
    a1, b1_ = ops.gn_affine(got, gamma, beta)       # fused output statistics
    ops.USE_FUSED_STATS = False
    try:
        a2, b2_ = ops.gn_affine(got, gamma, beta)
    finally:
        ops.USE_FUSED_STATS = True
    torch.testing.assert_close(a1, a2, rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(b1_, b2_, rtol=2e-4, atol=2e-5)


def test_up_phase_and_gn_backward_epilogue_reject_unsupported_shapes(ops):
    """The two late-round conv modes fail loudly (AdmError with the library's message) outside their domain; the callers
    (ops.conv / classifier backward) route such shapes to the general paths instead."""
    from autodiffusion_amd._lib import AdmError
    x8 = torch.zeros(2, 8, 8, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 64, 3, 3, device=DEV)
    b = torch.zeros(64, device=DEV)
    aff = (torch.ones(2, 64, device=DEV), torch.zeros(2, 64, device=DEV))
    x16 = torch.zeros(2, 16, 16, 64, dtype=torch.bfloat16, device=DEV)
    wf = ops.fold_weights(ops.pack_conv_weight(w), ops.pack_conv_weight(torch.zeros(64, 64, 1, 1, device=DEV)))
    with pytest.raises(AdmError, match="fold"):     # the fold needs the GN + SiLU prologue, no residual, an 8-wave tile
        ops.conv(x16, wf, b, 64, 9, fold=(x16, None))
    with pytest.raises(AdmError, match="fold"):
        ops.conv(x16, wf, b, 64, 9, aff=aff, silu=True, fold=(x16, None), variant=8)
    with pytest.raises(AdmError, match="fold"):
        ops.conv(x16[:, :4, :4].contiguous(), wf, b, 64, 9, aff=aff, silu=True, fold=(x16[:, :4, :4].contiguous(), None))
    with pytest.raises(AdmError, match="fused statistics|prologue 3"):
        ops.conv(x8, ops.pack_conv_weight(w), b, 64, 9, gnb=(x8, aff))          # 8x8 map: no 256-pixel tiles
    with pytest.raises(AdmError, match="gnb"):
        ops.conv(x8, ops.pack_conv_weight(w), b, 64, 9, gnb=(x8, aff), res=x8)  # the residual slot carries x
    # an 8x8 source keeps the one-launch virtual upsample even when phase weights are offered
    out = ops.conv(x8, ops.pack_conv_weight(w), b, 64, 9, in_up=True, w_up=ops.pack_conv_weight_up(w))
    assert out.shape == (2, 16, 16, 64)
    a = ops.ConvArgs()
    a.in0, a.w_packed, a.bias, a.out = x8.data_ptr(), ops.pack_conv_weight(w).data_ptr(), b.data_ptr(), out.data_ptr()
    a.n, a.h, a.w, a.c0, a.cout, a.taps, a.up_phase = 2, 8, 8, 64, 64, 9, 2
    from autodiffusion_amd import _lib
    rc = _lib.load().adm_conv(__import__("ctypes").byref(a), None)
    assert rc == -2 and b"up_phase" in _lib.load().adm_last_error()
