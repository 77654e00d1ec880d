"""FID statistics for candidate scoring: GPU accumulation, RCCL pooling, host Frechet distance.

Mirrors reference evaluations/evaluator_v1.py: ``FIDStatistics`` (:109-157),
``Evaluator.compute_statistics`` (:218-221) and ``cal_fid`` (:730-753).

* The 2048-d activations of a batch are folded into float64 running sums on the GPU
  (``adm_fid_accumulate``): no [N, 2048] array is gathered or copied to the host.
* With several ranks, each rank accumulates the activations of ITS images; ``pooled()`` runs ONE
  ``all_gather`` of (n, s1, s2) per candidate -- RCCL over xGMI on GPUs, gloo in the CPU tests --
  and sums the shards in rank order (bitwise deterministic).  This replaces the reference's
  per-batch uint8 image all_gather (search_imagenet64_classifier_guidance.py:356-361).
* ``frechet_distance`` keeps the reference's float64 scipy ``sqrtm`` formula on the host (seconds per candidate at
  2048 dimensions: with a 700 images/s sampler it is the per-candidate floor).  ``frechet_distance_device``
  evaluates the same quantity on the GPU from the pooled device sums: tr sqrtm(S1 S2) = sum sqrt(eig(S1^1/2 S2 S1^1/2)),
  two symmetric float64 eigendecompositions (rocSOLVER through ``torch.linalg.eigh``: a plain library factorisation).

The Inception-v3 pool3 extractor (in the reference a frozen TensorFlow graph fetched from a URL, evaluator_v1.py:652-679, or
``pytorch_fid``) is ``autodiffusion_amd.inception.InceptionV3`` on HIP layers; its weights are not available offline, so the
features' parity with the reference's is unpinned.  Any callable
``features(uint8 NHWC device batch) -> fp32 [B, D] device tensor`` (or an ``Evaluator_v1``-style object) plugs in.
"""
from __future__ import annotations

import os
import warnings
from typing import Optional

import numpy as np
import torch
from scipy import linalg

from . import _lib
from ._lib import check


class FIDStatistics:
    def __init__(self, mu: np.ndarray, sigma: np.ndarray):
        self.mu = mu
        self.sigma = sigma

    def frechet_distance(self, other, eps=1e-6):
        mu1, sigma1 = np.atleast_1d(self.mu), np.atleast_2d(self.sigma)
        mu2, sigma2 = np.atleast_1d(other.mu), np.atleast_2d(other.sigma)
        assert mu1.shape == mu2.shape, \
            "Training and test mean vectors have different lengths: " + str(mu1.shape) + ', ' + str(mu2.shape)
        assert sigma1.shape == sigma2.shape, \
            "Training and test mean vectors have different lengths: " + str(sigma1.shape) + ', ' + str(sigma2.shape)
        diff = mu1 - mu2
        covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
        if not np.isfinite(covmean).all():
            warnings.warn("fid calculation produces singular product; adding %s to diagonal of cov estimates" % eps)
            offset = np.eye(sigma1.shape[0]) * eps
            covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
        if np.iscomplexobj(covmean):
            if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
                raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
            covmean = covmean.real
        return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean)


def frechet_distance_device(mu1: torch.Tensor, sigma1: torch.Tensor, mu2: torch.Tensor, sigma2: torch.Tensor) -> float:
    """||mu1 - mu2||^2 + tr S1 + tr S2 - 2 tr sqrtm(S1 S2) in float64 on the tensors' device.

    For symmetric PSD S1, S2 the product's square root has the trace sum_i sqrt(lambda_i(R S2 R)), R = S1^(1/2): real
    by construction, and finite for singular products (num_samples < 2048 makes S1 rank-deficient), where the reference's
    Schur-based ``sqrtm`` (evaluator_v1.py:140-152) needs its eps-offset retry or returns a complex matrix whose
    imaginary part it discards.  Agreement with the host formula is tested to 1e-9 relative (tests/test_fid.py)."""
    mu1, sigma1, mu2, sigma2 = (t.to(torch.float64) for t in (mu1, sigma1, mu2, sigma2))
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    w, v = torch.linalg.eigh(sigma1)
    r = (v * w.clamp_min(0).sqrt()) @ v.T
    m = r @ sigma2 @ r
    ev = torch.linalg.eigvalsh((m + m.T) * 0.5)
    diff = mu1 - mu2
    return float(diff.dot(diff) + torch.trace(sigma1) + torch.trace(sigma2) - 2 * ev.clamp_min(0).sqrt().sum())


def frechet_distance_lowrank(acts: torch.Tensor, mu2: torch.Tensor, sigma2: torch.Tensor) -> float:
    """The same quantity from the n < dim activation rows themselves (float64 on their device).

    With n samples in d dimensions, S1 = Xc^T Xc / (n - 1) has rank <= n - 1 and the non-zero eigenvalues of S1 S2 are those of the
    n x n matrix Xc S2 Xc^T / (n - 1) (cyclic invariance), so tr sqrtm(S1 S2) = sum sqrt(eig(Xc S2 Xc^T / (n - 1))): one
    [n, d] x [d, d] x [d, n] product and an n x n symmetric eigenproblem instead of two d x d ones -- 64 x 64 instead of 2048 x 2048
    for BASELINE config 3's 64-image candidates (the search's regime: `num_samples <= 1000`, GD/README.md:22).  Equal to
    frechet_distance_device to ~1e-8 relative (that form also sums the square roots of the rounding-level eigenvalues of the
    d - n + 1 null directions; tests/test_fid.py)."""
    x = acts.to(torch.float64)
    n = x.shape[0]
    mu1 = x.mean(0)
    xc = x - mu1
    mu2, sigma2 = mu2.to(torch.float64), sigma2.to(torch.float64)
    m = (xc @ sigma2 @ xc.T) / (n - 1)
    ev = torch.linalg.eigvalsh((m + m.T) * 0.5)
    diff = mu1 - mu2
    return float(diff.dot(diff) + (xc * xc).sum() / (n - 1) + torch.trace(sigma2) - 2 * ev.clamp_min(0).sqrt().sum())


def compute_statistics(activations: np.ndarray) -> FIDStatistics:
    """Host form (np.mean / np.cov), used when the activations already live on the host."""
    return FIDStatistics(np.mean(activations, axis=0), np.cov(activations, rowvar=False))


class ActivationAccumulator:
    """Running (n, sum a, sum a a^T) of fp32 activations; float64 on the device that produces them."""

    def __init__(self, dim: int, device, keep_rows: int = 0):
        """keep_rows > 0: also keep the activation rows themselves while there are at most that many (callers that know the candidate
        has fewer samples than dimensions: frechet_distance_device then takes the n x n form, frechet_distance_lowrank)."""
        self.dim = dim
        self.device = torch.device(device)
        self.keep_rows = int(keep_rows)
        self.rows = [] if keep_rows > 0 else None
        self._ref_dev = None    # (id(ref), mu, sigma) of the last reference statistics, on the device
        self.n = 0
        self.s1 = torch.zeros(dim, dtype=torch.float64, device=self.device)
        self.s2 = torch.zeros(dim, dim, dtype=torch.float64, device=self.device)
        self._side = None       # stream of add_from()
        self._pending = False
        self.last_collective = None   # what pooled() last ran over the process group (bench.py / tests report it)

    OVERLAP = os.environ.get("ADM_FID_OVERLAP", "1") != "0"

    def add_from(self, features, u8: torch.Tensor):
        """add(features(u8)) on a SIDE stream, behind everything queued so far on the current stream: the extractor's launches
        (small grids on the 8x8 / 17x17 levels at sampling batch sizes) then fill CUs next to the next batch's sampling
        kernels instead of running between two batches.  Same launches in the same order on one stream: bitwise the sums of
        add(); every reader of the sums joins first.

        Stream contract of `features`: the side stream is only used for an extractor that declares `stream_safe = True`,
        i.e. that enqueues ALL its work on torch's CURRENT stream (the bundled HIP Inception-v3 does: inception.pool3_features).
        A plug-in that launches on the null stream or on a stream of its own would no longer be ordered with the
        accumulation kernel queued behind it here (a silent read-before-write of `acts`), so any other callable runs on the
        caller's stream, in order, as add(features(u8)) does."""
        if not (self.OVERLAP and u8.is_cuda and getattr(features, "stream_safe", False)):
            self.add(features(u8))
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)
        u8.record_stream(self._side)
        with torch.cuda.stream(self._side):
            self.add(features(u8), _joined=True)
        self._pending = True

    def join(self):
        """The current stream waits for the side stream's queued feature / accumulation work."""
        if self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._pending = False

    def add(self, acts: torch.Tensor, limit: Optional[int] = None, _joined: bool = False):
        """acts fp32 [B, dim]; `limit` keeps only the first rows (the reference's arr[:num_samples])."""
        if not _joined:
            self.join()
        if limit is not None:
            acts = acts[:limit]
        if acts.shape[0] == 0:
            return
        acts = acts.to(torch.float32).contiguous()
        if self.device.type != "cuda" or not acts.is_cuda:
            raise _lib.AdmError("ActivationAccumulator.add: activations must be device tensors "
                                "(the statistics kernel has no CPU fallback)")
        lib = _lib.load()
        check(lib.adm_fid_accumulate(acts.data_ptr(), self.s1.data_ptr(), self.s2.data_ptr(), acts.shape[0],
                                     self.dim, torch.cuda.current_stream().cuda_stream), "adm_fid_accumulate")
        self.n += int(acts.shape[0])
        if self.rows is not None:
            if self.n <= self.keep_rows:
                self.rows.append(acts.clone())
            else:
                self.rows = None            # more rows than announced: the Gram form serves

    def pooled(self, group=None, local=False):
        """(n, s1, s2) summed over ranks (local=True: this rank's sums only -- population-parallel mode): ONE all_gather per candidate of one packed float64 buffer
        [n | s1 (dim) | s2 (dim^2)] (8 B + 16 KiB + 32 MiB per rank at dim 2048), summed in rank order (deterministic).
        n rides in the buffer (exact in float64), so no per-rank host synchronisation is needed."""
        import torch.distributed as dist
        from .dist_util import collectives_on
        self.join()
        if local or not collectives_on(group):
            return self.n, self.s1, self.s2
        world = dist.get_world_size(group)
        d = self.dim
        # RCCL gathers device buffers over xGMI; gloo (CPU tests, one-GPU rehearsals) gathers host copies
        comm_dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        mine = torch.empty(1 + d + d * d, dtype=torch.float64, device=self.device)
        mine[0] = float(self.n)
        mine[1:1 + d] = self.s1
        mine[1 + d:] = self.s2.reshape(-1)
        mine = mine.to(comm_dev)
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        self.last_collective = {"op": "all_gather", "backend": dist.get_backend(group), "world_size": world,
                                "bytes_per_rank": int(mine.numel() * 8), "device": str(mine.device)}
        tot = parts[0].to(self.device)
        for r in range(1, world):
            tot = tot + parts[r].to(self.device)
        n = int(round(float(tot[0].item())))
        return n, tot[1:1 + d].contiguous(), tot[1 + d:].reshape(d, d).contiguous()

    def statistics(self, group=None, local=False) -> FIDStatistics:
        n, s1, s2 = self.pooled(group, local)
        if n < 2:
            raise ValueError("need at least 2 activations for a covariance")
        s1 = s1.cpu().numpy()
        s2 = s2.cpu().numpy()
        mu = s1 / n
        sigma = (s2 - n * np.outer(mu, mu)) / (n - 1)
        return FIDStatistics(mu, sigma)


    def frechet_distance_device(self, ref: FIDStatistics, group=None, local=False) -> float:
        """FID against host reference statistics without the activations' sums leaving the device."""
        from .dist_util import collectives_on
        if self._ref_dev is None or self._ref_dev[0] is not ref:     # one upload of the 32 MiB reference covariance per accumulator
            self._ref_dev = (ref, torch.as_tensor(np.asarray(ref.mu), dtype=torch.float64, device=self.device),
                             torch.as_tensor(np.asarray(ref.sigma), dtype=torch.float64, device=self.device))
        _, rmu, rsig = self._ref_dev
        if self.rows is not None and 2 <= self.n < self.dim and (local or not collectives_on(group)):
            self.join()
            # fewer samples than dimensions, all of them on this rank: the n x n eigenproblem (same value, float64)
            return frechet_distance_lowrank(torch.cat(self.rows, 0), rmu, rsig)
        n, s1, s2 = self.pooled(group, local)
        if n < 2:
            raise ValueError("need at least 2 activations for a covariance")
        mu = s1 / n
        sigma = (s2 - n * torch.outer(mu, mu)) / (n - 1)
        return frechet_distance_device(mu, sigma, rmu, rsig)


def cal_fid(batches, batch_size, evaluator, ref_stats, ref_stats_spatial=None):
    """Reference signature (evaluator_v1.py:730): uint8 NHWC numpy array + an evaluator object."""
    acts = evaluator.compute_activations(batches, batch_size)
    pool = acts[0] if isinstance(acts, (tuple, list)) else acts
    return compute_statistics(np.asarray(pool)).frechet_distance(ref_stats)


# ------------------------------------------------------------------ Stable-Diffusion search driver's FID helpers
# Same names, arguments and results as the functions of the reference's SD driver (Stable Diffusion scripts/search_ea.py:
# get_activations :95-127, calculate_activation_statistics :129-134, calculate_frechet_distance :136-169, calculate_fid
# :171-182), with ``model`` = autodiffusion_amd.inception.InceptionV3 (the HIP port of pytorch_fid's extractor).
def get_activations(data, model, batch_size=50, dims=2048, device="cuda", num_workers=1):
    """data: float tensor [N, 3, H, W] in [0, 1] -> numpy [N, dims] (float64 array as in the reference)."""
    model.eval()
    if batch_size > data.shape[0]:
        print(("Warning: batch size is bigger than the data size. "
               "Setting batch size to data size"))
        batch_size = data.shape[0]
    pred_arr = np.empty((data.shape[0], dims))
    for i in range(0, data.shape[0], batch_size):
        pred = model(data[i:i + batch_size].to(device))[0]
        if pred.size(2) != 1 or pred.size(3) != 1:
            pred = pred.mean((2, 3), keepdim=True)     # adaptive_avg_pool2d(pred, (1, 1))
        pred_arr[i:i + pred.shape[0]] = pred.squeeze(3).squeeze(2).cpu().numpy()
    return pred_arr


def calculate_activation_statistics(datas, model, batch_size=50, dims=2048, device="cuda", num_workers=1):
    act = get_activations(datas, model, batch_size, dims, device, num_workers)
    return np.mean(act, axis=0), np.cov(act, rowvar=False)


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    return FIDStatistics(mu1, sigma1).frechet_distance(FIDStatistics(mu2, sigma2), eps=eps)


def calculate_fid(data1, ref_mu, ref_sigma, batch_size, device, dims, num_workers=1, model=None):
    """The reference builds ``InceptionV3([block_idx]).to(device)`` per call (pytorch_fid fetches its weights); pass the
    loaded HIP extractor as ``model`` -- without it a fresh one with RANDOM weights is built and says so."""
    from .inception import InceptionV3
    if model is None:
        model = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]]).to(device)
    m1, s1 = calculate_activation_statistics(data1, model, batch_size, dims, device, num_workers)
    return calculate_frechet_distance(m1, s1, ref_mu, ref_sigma)
