"""Benchmark model definitions for the oracle (TEST INFRASTRUCTURE, see oracle/__init__.py).

Reference files followed:
  mfs/one_dim/ss_models.py:25-56    benes_bernoulli
  mfs/one_dim/ss_models.py:59-93    well_poisson
  dardel/convergence/convergence_mf.py:32-107   OU / Gaussian convergence model + exact Kalman filter
  tests/test_filtering.py:61-77     Kalman filter used as the analytic pin
"""
import math

import numpy as np
import scipy.special
import sympy as sp

from oracle.one_dim import GaussianSum1D, raw_moment_of_normal, central_moment_of_normal


# -- measurement models (jax.scipy.stats semantics: pmf = exp(logpmf)) -------------------------------------------------
def bernoulli_pmf(k, p):
    """jax.scipy.stats.bernoulli.pmf: exp(xlogy(k, p) + xlog1py(1 - k, -p))."""
    k = np.asarray(k, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    with np.errstate(all='ignore'):
        return np.exp(scipy.special.xlogy(k, p) + scipy.special.xlog1py(1. - k, -p))


def poisson_pmf(k, mu):
    """jax.scipy.stats.poisson.pmf: exp(xlogy(k, mu) - gammaln(k + 1) - mu)."""
    k = np.asarray(k, dtype=np.float64)
    mu = np.asarray(mu, dtype=np.float64)
    with np.errstate(all='ignore'):
        return np.exp(scipy.special.xlogy(k, mu) - scipy.special.gammaln(k + 1.) - mu)


def norm_pdf(y, loc, scale):
    """jax.scipy.stats.norm.pdf."""
    with np.errstate(all='ignore'):
        return np.exp(-0.5 * ((y - loc) / scale) ** 2) / (math.sqrt(2 * math.pi) * scale)


# -- mfs/one_dim/ss_models.py:25-56 -------------------------------------------------------------------------------------
def benes_bernoulli(N: int = 2):
    dt = 1e-2
    T = 100
    init_cond = GaussianSum1D.new(means=[-0.5, 0.5], variances=[0.05, 0.05], weights=[0.5, 0.5], N=N)

    def drift(x):  # SymPy-compatible (tme_sympy) and NumPy-compatible
        return sp.tanh(x) if isinstance(x, sp.Basic) else np.tanh(x)

    def dispersion(_):
        return 1.

    def logistic(x):
        return 1. / (1. + np.exp(-x ** 3 / 5.))

    def measurement_cond_pmf(y, x):
        return bernoulli_pmf(y, logistic(x))

    return dt, T, init_cond, drift, dispersion, logistic, measurement_cond_pmf


# -- mfs/one_dim/ss_models.py:59-93 -------------------------------------------------------------------------------------
def well_poisson(N: int = 2):
    dt = 1e-2
    T = 1000
    init_cond = GaussianSum1D.new(means=[-0.5, 0.5], variances=[0.05, 0.05], weights=[0.5, 0.5], N=N)

    def drift(x, p):
        return x * (1 - p * x ** 2)

    def dispersion(_):
        return 1.

    def emission(x, p):
        return np.log(1. + np.exp(p * x))

    def measurement_cond_pmf(y, x, p):
        return poisson_pmf(y, emission(x, p))

    return dt, T, init_cond, drift, dispersion, emission, measurement_cond_pmf


# -- dardel/convergence/convergence_mf.py:32-107 ------------------------------------------------------------------------
def ou_gaussian(N: int, dt: float = 1e-1, ell: float = 1., sigma: float = 0.5, measurement_noise_var: float = 1.,
                mean0: float = 0., var0: float = None):
    """Exact OU discretisation: X' | x ~ N(F x, Sigma); y | x ~ N(x, R)."""
    if var0 is None:
        var0 = sigma ** 2
    F, Sigma = math.exp(-dt / ell), sigma ** 2 * (1 - math.exp(-2 * dt / ell))
    num_moments = 2 * N

    def measurement_cond_pdf(y, x):
        return norm_pdf(y, x, math.sqrt(measurement_noise_var))

    def state_cond_raw_moments(x, n):
        x = np.asarray(x, dtype=np.float64)
        allp = np.stack([raw_moment_of_normal(F * x, Sigma, p) for p in range(num_moments)], axis=-1)
        return allp[..., np.asarray(n, dtype=int)]

    def state_cond_central_moments(x, n, mean):
        x = np.asarray(x, dtype=np.float64)
        allp = np.stack([raw_moment_of_normal(F * x - mean, Sigma, p) for p in range(num_moments)], axis=-1)
        return allp[..., np.asarray(n, dtype=int)]

    def state_cond_scaled_central_moments(x, n, mean, scale):
        return state_cond_central_moments(x, n, mean) / scale ** np.asarray(n, dtype=np.float64)

    def state_cond_mean(x):
        return F * np.asarray(x, dtype=np.float64)

    def state_cond_mean_var(x):
        x = np.asarray(x, dtype=np.float64)
        return F * x, np.full_like(x, Sigma)

    rms0 = np.array([float(raw_moment_of_normal(mean0, var0, p)) for p in range(num_moments)])
    cms0 = np.array([central_moment_of_normal(var0, p) for p in range(num_moments)])

    def kf(ys):
        """Exact Kalman filter (dardel/convergence/convergence_mf.py:64-80)."""
        mf, vf, nell = mean0, var0, 0.
        mfs, vfs = np.zeros(len(ys)), np.zeros(len(ys))
        for k, y in enumerate(ys):
            mp = F * mf
            vp = F * vf * F + Sigma
            s = vp + measurement_noise_var
            gain = vp / s
            mf = mp + gain * (y - mp)
            vf = vp - vp * gain
            nell -= -0.5 * math.log(2 * math.pi * s) - 0.5 * (y - mp) ** 2 / s
            mfs[k], vfs[k] = mf, vf
        return mfs, vfs, nell

    return dict(F=F, Sigma=Sigma, R=measurement_noise_var, mean0=mean0, var0=var0, rms0=rms0, cms0=cms0,
                pdf=measurement_cond_pdf, cond_rms=state_cond_raw_moments, cond_cms=state_cond_central_moments,
                cond_scms=state_cond_scaled_central_moments, cond_mean=state_cond_mean,
                cond_mean_var=state_cond_mean_var, kf=kf)
