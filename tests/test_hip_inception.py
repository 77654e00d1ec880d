"""GPU tests of the Inception-v3 pool3 extractor (autodiffusion_amd/inception.py, csrc/adm_convg.hip) -- PARITY UNPINNED.

The reference's features come from third-party code that is absent offline (a frozen TensorFlow graph,
evaluations/evaluator_v1.py:665-679; pytorch_fid, Stable Diffusion scripts/search_ea.py:95-127) and no reference fixture
holds an Inception output.  What IS checked here: every HIP layer against the plain PyTorch-CPU fp32 op it replaces, and
the assembled network against the CPU restatement of the published architecture (oracle/inception.py) on synthetic
weights.  Tolerances: operands rounded to the 16-bit type on both sides for the single layers (summation order + the
output rounding remain: 4e-3 Frobenius); the ~50-layer network accumulates that rounding: 5e-3 in fp16, 4e-2 in bf16.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd import ops as _ops
    return _ops


def rnd(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def q(x, dt):
    return x.to(dt).to(torch.float32)


def nhwc(x, dt, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad > x.shape[3]:
        x = F.pad(x, (0, cpad - x.shape[3]))
    return x.to(dt).to(DEV)


def rel(got, ref):
    return ((got - ref).norm() / (ref.norm() + 1e-12)).item()


CASES = [  # cin, cout, kh, kw, stride, (ph, pw), h, w
    (3, 32, 3, 3, 2, (0, 0), 31, 31), (32, 32, 3, 3, 1, (0, 0), 17, 19), (32, 64, 3, 3, 1, (1, 1), 16, 16),
    (64, 80, 1, 1, 1, (0, 0), 9, 9), (80, 192, 3, 3, 1, (0, 0), 11, 11), (48, 64, 5, 5, 1, (2, 2), 12, 12),
    (128, 128, 1, 7, 1, (0, 3), 17, 17), (160, 192, 7, 1, 1, (3, 0), 17, 17), (288, 384, 3, 3, 2, (0, 0), 35, 35),
    (384, 384, 1, 3, 1, (0, 1), 8, 8), (384, 384, 3, 1, 1, (1, 0), 8, 8), (448, 384, 3, 3, 1, (1, 1), 8, 8),
    (2048, 320, 1, 1, 1, (0, 0), 8, 8), (96, 96, 3, 3, 2, (0, 0), 8, 9), (64, 36, 3, 3, 2, (1, 1), 16, 16),
]


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,kh,kw,stride,pad,h,w", CASES)
def test_conv2d_matches_torch(ops, dt, cin, cout, kh, kw, stride, pad, h, w):
    n = 3
    x = rnd((n, cin, h, w), 1)
    wt = rnd((cout, cin, kh, kw), 2, (cin * kh * kw) ** -0.5)
    scale, bias = 1 + 0.2 * rnd((cout,), 3), rnd((cout,), 4, 0.3)
    wp = ops.pack_conv2d_weight(wt.to(DEV), scale.to(DEV), dt)
    ref = F.relu(F.conv2d(q(x, dt), q(wt * scale.view(-1, 1, 1, 1), dt), bias, stride=stride, padding=pad))
    got = ops.conv2d(nhwc(x, dt, wp.shape[2]), wp, bias.to(DEV), kh, kw, stride, pad, True)
    assert got.shape == (n, ref.shape[2], ref.shape[3], cout)
    got = got.float().cpu().permute(0, 3, 1, 2)
    err = (got - ref).abs().max().item()
    assert err <= 1e-2 * ref.abs().max().item() and rel(got, ref) <= 4e-3, (err, rel(got, ref))
    # without ReLU / bias, and written into a channel slice of a wider tensor (the concat of a Mixed block)
    big = torch.full((n, ref.shape[2], ref.shape[3], cout + 40), 7.0, dtype=dt, device=DEV)
    ops.conv2d(nhwc(x, dt, wp.shape[2]), wp, None, kh, kw, stride, pad, False, out=big[..., 8:8 + cout])
    ref2 = F.conv2d(q(x, dt), q(wt * scale.view(-1, 1, 1, 1), dt), None, stride=stride, padding=pad)
    got2 = big[..., 8:8 + cout].float().cpu().permute(0, 3, 1, 2)
    assert rel(got2, ref2) <= 4e-3
    assert (big[..., :8] == 7).all() and (big[..., 8 + cout:] == 7).all()   # neighbours untouched


@pytest.mark.parametrize("k,stride,pad,mode", [(3, 2, 0, "max"), (3, 1, 1, "avg"), (3, 1, 1, "max"), (2, 2, 0, "avg")])
def test_pool2d_matches_torch(ops, k, stride, pad, mode):
    x = rnd((2, 40, 13, 11), 5)
    xd = nhwc(x, torch.float16)
    ref = (F.max_pool2d(q(x, torch.float16), k, stride, pad) if mode == "max"
           else F.avg_pool2d(q(x, torch.float16), k, stride, pad, count_include_pad=False))
    got = ops.pool2d(xd, k, stride, pad, mode).float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    torch.testing.assert_close(got, q(ref, torch.float16), rtol=2e-3, atol=2e-3)
    big = torch.zeros((2, ref.shape[2], ref.shape[3], 64), dtype=torch.float16, device=DEV)
    ops.pool2d(xd, k, stride, pad, mode, out=big[..., 16:56])
    assert torch.equal(big[..., 16:56].float().cpu().permute(0, 3, 1, 2), got) and (big[..., :16] == 0).all()


def test_global_avgpool(ops):
    x = rnd((3, 2048, 8, 8), 6)
    got = ops.global_avgpool_f32(nhwc(x, torch.float16)).cpu()
    torch.testing.assert_close(got, q(x, torch.float16).mean((2, 3)), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("size", [64, 256, 299, 512])
def test_resize_matches_both_conventions(ops, size):
    from oracle import inception as oi
    g = torch.Generator().manual_seed(size)
    u8 = torch.randint(0, 256, (2, size, size - 3, 3), generator=g, dtype=torch.uint8)
    ref = oi.prepare(u8, "tf1")                                     # TensorFlow-1 ResizeBilinear, (x - 128) / 128
    got = ops.resize_bilinear(u8.to(DEV), 299, 299, 32, "u8_nhwc", False, 1 / 128.0, -1.0, torch.float16)
    assert (got[..., 3:] == 0).all()
    torch.testing.assert_close(got[..., :3].float().cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=2e-3)
    f = torch.rand((2, 3, size, size + 5), generator=g)
    ref = oi.prepare(f, "pt")                                       # torch bilinear, align_corners=False, 2x - 1
    got = ops.resize_bilinear(f.to(DEV), 299, 299, 32, "f32_nchw", True, 2.0, -1.0, torch.float16)
    torch.testing.assert_close(got[..., :3].float().cpu().permute(0, 3, 1, 2), ref, rtol=0, atol=2e-3)


def _pytorch_fid_names(p):
    """The same tensors under pytorch_fid's own state_dict names (its layers sit in nn.Sequential blocks)."""
    blocks = [["Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3"], ["Conv2d_3b_1x1", "Conv2d_4a_3x3"],
              ["Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"],
              ["Mixed_7a", "Mixed_7b", "Mixed_7c"]]
    where = {nm: (i, j) for i, blk in enumerate(blocks) for j, nm in enumerate(blk)}
    out = {}
    for k, v in p.items():
        top, rest = k.split(".", 1)
        out["blocks.%d.%d.%s" % (*where[top], rest)] = v
    out["fc.weight"] = torch.zeros(1008, 2048)   # a full checkpoint also carries the classifier head
    return out


@pytest.fixture(scope="module")
def nets():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd.inception import InceptionV3
    from oracle import inception as oi
    p = oi.fill_params()
    return oi, p, InceptionV3


@pytest.mark.parametrize("dt,tol", [(torch.float16, 5e-3), (torch.bfloat16, 4e-2)])
def test_network_matches_cpu_restatement_parity_unpinned(nets, dt, tol):
    oi, p, InceptionV3 = nets
    x = torch.rand((3, 3, 64, 64), generator=torch.Generator().manual_seed(11))
    ref = oi.forward(p, oi.prepare(x, "pt"))
    m = InceptionV3([0, 1, 2, 3], dtype=dt).to(DEV)
    m.load_state_dict(p)
    got = m(x.to(DEV))
    assert [tuple(g.shape) for g in got] == [(3, 64, 73, 73), (3, 192, 35, 35), (3, 768, 17, 17), (3, 2048, 1, 1)]
    errs = [rel(g.cpu(), r) for g, r in zip(got, ref)]
    assert max(errs) <= tol, errs
    # pytorch_fid's call shape: InceptionV3([block_idx])(batch)[0]
    only = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[2048]], dtype=dt).to(DEV)
    only.load_state_dict(_pytorch_fid_names(p))
    assert torch.equal(only(x.to(DEV))[0], got[3])


def test_uint8_features_tf1_convention_parity_unpinned(nets):
    oi, p, InceptionV3 = nets
    u8 = torch.randint(0, 256, (5, 64, 64, 3), generator=torch.Generator().manual_seed(12), dtype=torch.uint8)
    m = InceptionV3().to(DEV)
    m.load_state_dict(p)
    got = m.features(u8.to(DEV)).cpu()
    assert got.shape == (5, 2048) and got.dtype == torch.float32
    assert rel(got, oi.pool3(p, u8, "tf1")) <= 5e-3
    assert rel(m.features(u8.to(DEV), "pt").cpu(), oi.pool3(p, u8.permute(0, 3, 1, 2).float() / 255.0, "pt")) <= 5e-3
    # a row does not depend on its batch (chunking, tile position)
    m.CHUNK = 2
    assert torch.equal(m.features(u8.to(DEV)).cpu(), got)
    # the reference evaluator's shape: compute_activations(uint8 NHWC numpy, batch_size) -> (pool_3, spatial)
    from autodiffusion_amd.inception import Evaluator_v1
    acts = Evaluator_v1(m).compute_activations(u8.numpy(), 2)
    assert np.array_equal(acts[0], got.numpy())


def test_fid_of_device_features_against_host_formula(nets):
    """uint8 batch -> HIP Inception -> float64 device sums -> Frechet distance, against the oracle's numpy statistics of
    the same activations (the activations themselves are the HIP ones: this pins the plumbing, not Inception)."""
    oi, p, InceptionV3 = nets
    from autodiffusion_amd.fid import ActivationAccumulator, FIDStatistics
    from oracle import fid as ofid
    m = InceptionV3().to(DEV)
    m.load_state_dict(p)
    g = torch.Generator().manual_seed(13)
    a = torch.randint(0, 256, (24, 32, 32, 3), generator=g, dtype=torch.uint8)
    b = torch.randint(0, 200, (24, 32, 32, 3), generator=g, dtype=torch.uint8)
    fa, fb = m.features(a.to(DEV)), m.features(b.to(DEV))
    acc = ActivationAccumulator(2048, DEV)
    acc.add(fa)
    mu, sig = ofid.statistics(fb.double().cpu().numpy())
    got = acc.statistics().frechet_distance(FIDStatistics(mu, sig))
    want = ofid.frechet_distance(*ofid.statistics(fa.double().cpu().numpy()), mu, sig)
    assert abs(got - want) <= 1e-6 * max(1.0, abs(want))


def test_sd_driver_fid_helpers_parity_unpinned(nets):
    """calculate_fid of the Stable-Diffusion search driver (scripts/search_ea.py:95-182) over the HIP extractor, against
    the same statistics taken from the CPU restatement's activations (dims 2048 and the 192-d block-1 features)."""
    oi, p, InceptionV3 = nets
    from autodiffusion_amd import fid
    from oracle import fid as ofid
    x = torch.rand((10, 3, 48, 48), generator=torch.Generator().manual_seed(14))
    for dims, blk in ((2048, 3), (192, 1)):
        m = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(DEV)
        m.load_state_dict(p)
        acts = fid.get_activations(x, m, batch_size=4, dims=dims, device=DEV)
        with torch.no_grad():
            ref = oi.forward(p, oi.prepare(x, "pt"), upto=blk)[blk].mean((2, 3)).double().numpy()
        assert acts.shape == (10, dims) and acts.dtype == np.float64
        assert np.linalg.norm(acts - ref) <= 5e-3 * np.linalg.norm(ref)
        rng = np.random.RandomState(dims)
        a = rng.randn(dims, dims) / np.sqrt(dims)
        rmu, rsig = rng.randn(dims) * 0.1, a @ a.T + 0.05 * np.eye(dims)
        got = fid.calculate_fid(x, rmu, rsig, 4, DEV, dims, model=m)
        want = ofid.frechet_distance(*ofid.statistics(acts), rmu, rsig)
        assert abs(got - want) <= 1e-8 * max(1.0, abs(want))


def test_overlapped_feature_accumulation_is_bitwise_the_sequential_one(nets):
    """ActivationAccumulator.add_from runs extractor + Gram on a side stream next to the caller's stream: the same launches in
    the same order, so the float64 sums are bitwise those of add(features(u8))."""
    oi, p, InceptionV3 = nets
    from autodiffusion_amd.fid import ActivationAccumulator
    m = InceptionV3().to(DEV)
    m.load_state_dict(p)
    g = torch.Generator().manual_seed(21)
    batches = [torch.randint(0, 256, (6, 32, 32, 3), generator=g, dtype=torch.uint8).to(DEV) for _ in range(3)]
    a, b = ActivationAccumulator(2048, DEV), ActivationAccumulator(2048, DEV)
    for u8 in batches:
        a.add(m.features(u8))
        b.add_from(m.features, u8)
        torch.randn(256, 256, device=DEV) @ torch.randn(256, 256, device=DEV)   # unrelated work on the caller's stream
    na, s1a, s2a = a.pooled()
    nb, s1b, s2b = b.pooled()
    assert na == nb == 18 and torch.equal(s1a, s1b) and torch.equal(s2a, s2b)
