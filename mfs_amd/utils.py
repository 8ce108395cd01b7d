"""Initial distributions, mirroring the parts of `mfs.utils` on the hot path (mfs/utils.py:39-74, 77-125).
Host side, NumPy, once per run."""
import math
from typing import NamedTuple

import numpy as np

from mfs_amd.one_dim.moments import raw_moment_of_normal


class GaussianSum1D(NamedTuple):
    """Unidimensional Gaussian-sum distribution (mfs/utils.py:39-74)."""
    means: np.ndarray
    variances: np.ndarray
    weights: np.ndarray
    mean: float
    variance: float
    rms: np.ndarray
    cms: np.ndarray
    scms: np.ndarray

    def pdf(self, xs):
        xs = np.atleast_1d(np.asarray(xs, dtype=np.float64))[:, None]
        pdfs = np.exp(-0.5 * (xs - self.means) ** 2 / self.variances) / np.sqrt(2 * math.pi * self.variances)
        return np.sum(pdfs * self.weights[None, :], axis=1)

    def sampler(self, rng: np.random.Generator, n: int):
        cs = rng.choice(self.means.shape[0], size=n, p=self.weights)
        return self.means[cs] + np.sqrt(self.variances[cs]) * rng.standard_normal(n)

    @classmethod
    def new(cls, means, variances, weights, N: int = 2):
        means, variances, weights = (np.asarray(a, dtype=np.float64) for a in (means, variances, weights))
        centre = float(np.sum(means * weights))
        rms = np.array([sum(raw_moment_of_normal(m, v, p) * w for m, v, w in zip(means, variances, weights))
                        for p in range(2 * N)])
        cms = np.array([sum(raw_moment_of_normal(m - centre, v, p) * w for m, v, w in zip(means, variances, weights))
                        for p in range(2 * N)])
        variance = float(cms[2])
        scms = cms / np.sqrt(variance) ** np.arange(2 * N)
        return cls(means=means, variances=variances, weights=weights, mean=centre, variance=variance,
                   rms=rms, cms=cms, scms=scms)


class GaussianSumND(NamedTuple):
    """Multidimensional Gaussian-sum distribution (mfs/utils.py:77-125)."""
    d: int
    means: np.ndarray
    covs: np.ndarray
    weights: np.ndarray
    mean: np.ndarray
    cov: np.ndarray
    rms: np.ndarray
    cms: np.ndarray

    def sampler(self, rng: np.random.Generator, nsamples: int):
        cs = rng.choice(self.means.shape[0], size=nsamples, p=self.weights)
        chol = np.linalg.cholesky(self.covs[cs])
        return self.means[cs] + np.einsum('...ij,...j->...i', chol, rng.standard_normal((nsamples, self.d)))

    @classmethod
    def new(cls, means, covs, weights, multi_indices):
        from mfs_amd.multi_dims.moments import raw_moments_mvn_kan
        means, covs, weights = (np.asarray(a, dtype=np.float64) for a in (means, covs, weights))
        d = means.shape[1]
        centre = np.sum(means * weights[:, None], axis=0)
        cov = sum(w * (c + np.outer(m, m)) for m, c, w in zip(means, covs, weights)) - np.outer(centre, centre)
        rms = sum(w * np.array([raw_moments_mvn_kan(m, c, mi) for mi in multi_indices])
                  for m, c, w in zip(means, covs, weights))
        cms = sum(w * np.array([raw_moments_mvn_kan(m - centre, c, mi) for mi in multi_indices])
                  for m, c, w in zip(means, covs, weights))
        return cls(d=d, means=means, covs=covs, weights=weights, mean=centre, cov=cov, rms=rms, cms=cms)
