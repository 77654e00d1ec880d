"""FID statistics: Frechet distance (product and oracle) against analytic cases and a direct scipy
evaluation; pooled statistics over 2 ranks (gloo, CPU); GPU accumulation vs numpy float64."""
import json
import os
import socket

import numpy as np
import pytest
import torch
from scipy import linalg

from autodiffusion_amd.fid import (ActivationAccumulator, FIDStatistics, compute_statistics,
                                   frechet_distance_device)
from oracle import fid as ofid


def _spd(d, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((d, d))
    return a @ a.T / d + 0.1 * np.eye(d)


def test_frechet_distance_analytic_and_direct():
    d = 16
    mu1, mu2 = np.arange(d) / d, np.ones(d) * 0.3
    # commuting (diagonal) covariances: tr sqrt(S1 S2) = sum sqrt(s1 s2)
    s1, s2 = np.diag(np.linspace(0.5, 2, d)), np.diag(np.linspace(1, 3, d))
    want = ((mu1 - mu2) ** 2).sum() + (np.sqrt(np.diag(s1)) - np.sqrt(np.diag(s2))).__pow__(2).sum()
    for fn in (lambda: FIDStatistics(mu1, s1).frechet_distance(FIDStatistics(mu2, s2)),
               lambda: ofid.frechet_distance(mu1, s1, mu2, s2)):
        assert abs(fn() - want) < 1e-9
    # general SPD pair vs an independent evaluation through the symmetric form sqrt(S1^1/2 S2 S1^1/2)
    a, b = _spd(d, 1), _spd(d, 2)
    ra = linalg.sqrtm(a).real
    tr = np.trace(linalg.sqrtm(ra @ b @ ra).real)
    want = ((mu1 - mu2) ** 2).sum() + np.trace(a) + np.trace(b) - 2 * tr
    got = FIDStatistics(mu1, a).frechet_distance(FIDStatistics(mu2, b))
    assert abs(got - want) < 1e-8 and abs(ofid.frechet_distance(mu1, a, mu2, b) - want) < 1e-8
    assert abs(FIDStatistics(mu1, a).frechet_distance(FIDStatistics(mu1, a))) < 1e-6


def test_frechet_distance_rank_deficient_product():
    d = 8
    v = np.zeros((d, d)); v[0, 0] = 1.0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # the singular-product branch may warn, depending on LAPACK
        got = FIDStatistics(np.zeros(d), v).frechet_distance(FIDStatistics(np.zeros(d), np.eye(d)))
    assert np.isfinite(got) and abs(got - (1 + d - 2)) < 1e-3
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert abs(ofid.frechet_distance(np.zeros(d), v, np.zeros(d), np.eye(d)) - got) < 1e-9


def test_frechet_distance_symmetric_eigen_form_matches_the_host_formula():
    """frechet_distance_device (float64 eigh; runs on whatever device holds the tensors -- here the CPU) against the
    reference's sqrtm formula: well-conditioned, sample covariances, and rank-deficient (fewer samples than dimensions)."""
    import warnings
    rng = np.random.default_rng(5)
    d = 48
    cases = [(_spd(d, 1), _spd(d, 2))]
    for n in (400, 30):  # 30 < d: singular covariance, as num_samples=1000 against 2048 Inception dimensions
        a = rng.standard_normal((n, d)) @ rng.standard_normal((d, d)) * 0.3
        b = rng.standard_normal((500, d)) @ rng.standard_normal((d, d)) * 0.3 + 0.1
        cases.append((np.cov(a, rowvar=False), np.cov(b, rowvar=False)))
    mu1, mu2 = rng.standard_normal(d), rng.standard_normal(d)
    for s1, s2 in cases:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = FIDStatistics(mu1, s1).frechet_distance(FIDStatistics(mu2, s2))
        got = frechet_distance_device(*(torch.from_numpy(x) for x in (mu1, s1, mu2, s2)))
        assert abs(got - want) <= 1e-7 * abs(want), (got, want)


def test_compute_statistics_matches_oracle():
    acts = np.random.default_rng(0).standard_normal((50, 12)).astype(np.float32)
    st = compute_statistics(acts)
    mu, sigma = ofid.statistics(acts)
    assert np.array_equal(st.mu, mu) and np.array_equal(st.sigma, sigma)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, d, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(100)
    acts = rng.standard_normal((37, d))            # the whole candidate's activations
    mine = acts[rank::world]                         # image-sharded over ranks
    acc = ActivationAccumulator(d, "cpu")
    acc.n = len(mine)
    acc.s1 = torch.from_numpy(mine.sum(0))
    acc.s2 = torch.from_numpy(mine.T @ mine)
    st = acc.statistics()
    if rank == 0:
        np.savez(out, mu=st.mu, sigma=st.sigma)
    dist.barrier()
    dist.destroy_process_group()


def test_pooled_statistics_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    d, out = 6, str(tmp_path / "pooled.npz")
    mp.spawn(_rank_main, args=(2, _free_port(), d, out), nprocs=2, join=True)
    z = np.load(out)
    acts = np.random.default_rng(100).standard_normal((37, d))
    np.testing.assert_allclose(z["mu"], acts.mean(0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(z["sigma"], np.cov(acts, rowvar=False), rtol=1e-10, atol=1e-12)


def _forced_world1_main(rank, port, d, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ADM_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    acts = np.random.default_rng(5).standard_normal((9, d))
    acc = ActivationAccumulator(d, "cpu")
    acc.n, acc.s1, acc.s2 = 9, torch.from_numpy(acts.sum(0)), torch.from_numpy(acts.T @ acts)
    st = acc.statistics()
    np.savez(out, mu=st.mu, sigma=st.sigma, coll=json.dumps(acc.last_collective))
    os.environ["ADM_FORCE_COLLECTIVES"] = "0"
    acc.last_collective = None
    acc.statistics()
    assert acc.last_collective is None          # the single-rank short-cut again
    dist.destroy_process_group()


def test_forced_world_size_one_group_runs_the_all_gather(tmp_path):
    """ADM_FORCE_COLLECTIVES=1 (bench.py --force-dist): a one-rank group still takes pooled() through the process group -- the switch
    the GPU box uses to run the 32 MiB all_gather on RCCL (tests/test_hip_rccl.py); here on gloo."""
    import torch.multiprocessing as mp
    d, out = 5, str(tmp_path / "w1.npz")
    mp.spawn(_forced_world1_main, args=(_free_port(), d, out), nprocs=1, join=True)
    z = np.load(out)
    acts = np.random.default_rng(5).standard_normal((9, d))
    np.testing.assert_allclose(z["sigma"], np.cov(acts, rowvar=False), rtol=1e-10, atol=1e-12)
    assert json.loads(str(z["coll"])) == {"op": "all_gather", "backend": "gloo", "world_size": 1, "bytes_per_rank": 8 * (1 + d + d * d), "device": "cpu"}


def test_lowrank_frechet_distance_equals_the_covariance_forms():
    """n < d samples: the n x n eigen form (frechet_distance_lowrank) against the reference's scipy sqrtm formula on np.cov and against
    the d x d eigen form, float64 on the CPU."""
    from autodiffusion_amd.fid import frechet_distance_lowrank
    rng = np.random.default_rng(3)
    for n, d in ((10, 24), (5, 64), (23, 24)):
        acts = rng.standard_normal((n, d)) * 0.7 + 0.2
        ref = FIDStatistics(rng.standard_normal(d) * 0.1, _spd(d, 11))
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            want = float(compute_statistics(acts).frechet_distance(ref))
        got = frechet_distance_lowrank(torch.from_numpy(acts), torch.from_numpy(ref.mu), torch.from_numpy(ref.sigma))
        st = compute_statistics(acts)
        dev = frechet_distance_device(torch.from_numpy(st.mu), torch.from_numpy(st.sigma), torch.from_numpy(ref.mu), torch.from_numpy(ref.sigma))
        # the d x d form also sums sqrt(|rounding-level eigenvalues|) of the d - n + 1 null directions (~1e-8 each): 1e-7 apart
        assert abs(got - dev) <= 1e-7 * abs(dev), (n, d, got, dev)
        assert abs(got - want) <= 1e-6 * abs(want), (n, d, got, want)


def test_accumulator_refuses_host_activations():
    from autodiffusion_amd._lib import AdmError
    with pytest.raises(AdmError):
        ActivationAccumulator(4, "cpu").add(torch.zeros(3, 4))


@pytest.mark.gpu
def test_gpu_accumulation_matches_numpy_float64():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    d = 2048
    acts = torch.randn(700, d, generator=torch.Generator().manual_seed(0)) * 0.7 + 0.2
    acc = ActivationAccumulator(d, "cuda:0")
    for i in range(0, 700, 256):                     # ragged last batch
        acc.add(acts[i:i + 256].to("cuda:0"))
    acc.add(acts[:10].to("cuda:0"), limit=0)         # arr[:num_samples] truncation
    st = acc.statistics()
    a64 = acts.numpy().astype(np.float64)
    np.testing.assert_allclose(st.mu, a64.mean(0), rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(st.sigma, np.cov(a64, rowvar=False), rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
def test_gpu_frechet_distance_matches_host_sqrtm_at_inception_width():
    """2048 dimensions, 700 generated samples (rank-deficient covariance, the search's regime) against a full-rank
    reference: the device eigen form vs the reference's host sqrtm formula; prints both wall times."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    import warnings
    d = 2048
    g = torch.Generator().manual_seed(1)
    mix = torch.randn(d, d, generator=g) * d ** -0.5
    acts = (torch.randn(700, d, generator=g) @ mix) * 0.7 + 0.2
    ref_acts = ((torch.randn(4096, d, generator=g) @ mix) * 0.8 + 0.25).numpy().astype(np.float64)
    ref = FIDStatistics(ref_acts.mean(0), np.cov(ref_acts, rowvar=False))
    acc = ActivationAccumulator(d, "cuda:0")
    acc.add(acts.to("cuda:0"))
    acc.frechet_distance_device(ref)  # first call loads the solver
    torch.cuda.synchronize()
    t0 = time.time()
    got = acc.frechet_distance_device(ref)
    t_dev = time.time() - t0
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = float(acc.statistics().frechet_distance(ref))
    t_host = time.time() - t0
    print(f"FID {got:.6f} (device, {t_dev:.2f} s) vs {want:.6f} (host sqrtm, {t_host:.2f} s)")
    assert abs(got - want) <= 1e-6 * abs(want)


@pytest.mark.gpu
def test_gpu_accumulator_takes_the_lowrank_form_for_small_candidates():
    """keep_rows: a candidate with fewer samples than feature dimensions (BASELINE config 3: 64 images against 2048) gets its FID from
    the n x n eigenproblem; more rows than announced, or a pooled (multi-rank) call, fall back to the Gram form.  Same value."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import time
    d = 2048
    g = torch.Generator().manual_seed(4)
    mix = torch.randn(d, d, generator=g) * d ** -0.5
    acts = (torch.randn(96, d, generator=g) @ mix) * 0.7 + 0.2
    ref_acts = ((torch.randn(4096, d, generator=g) @ mix) * 0.8 + 0.25).numpy().astype(np.float64)
    ref = FIDStatistics(ref_acts.mean(0), np.cov(ref_acts, rowvar=False))
    gram, low = ActivationAccumulator(d, "cuda:0"), ActivationAccumulator(d, "cuda:0", keep_rows=96)
    for a in (gram, low):
        for i in range(0, 96, 32):
            a.add(acts[i:i + 32].to("cuda:0"))
    assert low.rows is not None and len(low.rows) == 3 and gram.rows is None
    want = gram.frechet_distance_device(ref)
    low.frechet_distance_device(ref)
    torch.cuda.synchronize()
    t0 = time.time(); got = low.frechet_distance_device(ref); t_low = time.time() - t0
    t0 = time.time(); gram.frechet_distance_device(ref); t_gram = time.time() - t0
    print(f"FID of 96 samples in 2048 dimensions: {got:.6f} (n x n form, {t_low * 1e3:.1f} ms) vs {want:.6f} (d x d form, {t_gram * 1e3:.1f} ms)")
    assert abs(got - want) <= 1e-7 * abs(want)
    over = ActivationAccumulator(d, "cuda:0", keep_rows=64)
    for i in range(0, 96, 32):
        over.add(acts[i:i + 32].to("cuda:0"))
    assert over.rows is None and abs(over.frechet_distance_device(ref) - want) <= 1e-12 * abs(want)


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(37, 100), (5, 64), (300, 196), (1, 2048)])
def test_gram_kernel_ragged_shapes_and_symmetry(n, d):
    """The f64-MFMA Gram kernel computes the upper triangle of tiles and mirrors it: ragged sample counts / widths that are
    not multiples of the 64-wide tile or the 16-sample stage, against numpy float64, accumulated over two calls."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    g = torch.Generator().manual_seed(n * 1000 + d)
    a1, a2 = torch.randn(n, d, generator=g) * 1.3 - 0.4, torch.randn(n + 3, d, generator=g)
    acc = ActivationAccumulator(d, "cuda:0")
    acc.add(a1.to("cuda:0"))
    acc.add(a2.to("cuda:0"))
    a = np.concatenate([a1.numpy(), a2.numpy()]).astype(np.float64)
    s2 = acc.s2.cpu().numpy()
    assert acc.n == 2 * n + 3 and np.array_equal(s2, s2.T)          # mirrored tiles are bit-identical
    np.testing.assert_allclose(s2, a.T @ a, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(acc.s1.cpu().numpy(), a.sum(0), rtol=1e-12, atol=1e-12)
