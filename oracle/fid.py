"""Oracle: FID statistics and Frechet distance (TEST INFRASTRUCTURE).

numpy/scipy float64 restatement of
  * ``Evaluator.compute_statistics`` -- reference evaluations/evaluator_v1.py:218-221
  * ``FIDStatistics.frechet_distance`` -- ibid. :114-157

The Inception-v3 pool3 feature extractor (a frozen TensorFlow graph fetched
from a URL, ibid. :652-679) is third-party, absent offline and pinned by no
reference fixture: "parity unpinned" for the features; the statistics below
are exact restatements and are pinned by synthetic (mu, sigma) cases checked
against a direct scipy evaluation in tests/test_fid.py.
"""
from __future__ import annotations

import warnings

import numpy as np
from scipy import linalg


def statistics(acts: np.ndarray):
    mu = np.mean(acts, axis=0)
    sigma = np.cov(acts, rowvar=False)
    return mu, sigma


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
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
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))
