"""Test models used in the paper, mirroring `mfs.one_dim.ss_models` (same factory names and return tuples).

The returned callables work on NumPy arrays and on the tracing placeholders of `mfs_amd.sym`, so user closures such as
`lambda x: drift(x, p1)` (dardel/parameter_estimation/mf.py:41-45) reduce to device tables; `p1` / `p2` may be arrays
of shape (B,) for per-replicate parameters.  `simulate_trajectory(x0, rng)` replaces the JAX-PRNG simulator with a
NumPy one (data generation is outside the accelerated path).
"""
import numpy as np

from mfs_amd import sym, stats
from mfs_amd.utils import GaussianSum1D


def _simulator(drift_np, dt, T, integration_steps=100):
    def simulate_trajectory(x0, rng: np.random.Generator):
        ddt = dt / integration_steps
        x = float(x0)
        xs = np.empty(T)
        for k in range(T):
            dws = rng.standard_normal(integration_steps) * np.sqrt(ddt)
            for j in range(integration_steps):
                x = x + drift_np(x) * ddt + dws[j]
            xs[k] = x
        return xs

    return simulate_trajectory


def benes_bernoulli(N: int = 2):
    """The Benes--Bernoulli model (mfs/one_dim/ss_models.py:25-56)."""
    dt = 1e-2
    T = 100
    ts = np.linspace(dt, dt * T, T)
    init_cond = GaussianSum1D.new(means=[-0.5, 0.5], variances=[0.05, 0.05], weights=[0.5, 0.5], N=N)

    def drift(x):
        return sym.tanh(x)

    def dispersion(_):
        return 1.

    def logistic(x):
        return 1 / (1 + sym.exp(-x ** 3 / 5))

    def measurement_cond_pmf(y, x):
        return stats.bernoulli_pmf(y, logistic(x))

    return dt, T, ts, init_cond, drift, dispersion, logistic, measurement_cond_pmf, _simulator(np.tanh, dt, T)


def well_poisson(true_p1, N: int = 2):
    """The Well--Poisson model for parameter estimation (mfs/one_dim/ss_models.py:59-93)."""
    dt = 1e-2
    T = 1000
    ts = np.linspace(dt, dt * T, T)
    init_cond = GaussianSum1D.new(means=[-0.5, 0.5], variances=[0.05, 0.05], weights=[0.5, 0.5], N=N)

    def drift(x, p):
        return x * (1 - p * x ** 2)

    def dispersion(_):
        return 1.

    def emission(x, p):
        return sym.log(1. + sym.exp(p * x))

    def measurement_cond_pmf(y, x, p):
        return stats.poisson_pmf(y, emission(x, p))

    return dt, T, ts, init_cond, drift, dispersion, emission, measurement_cond_pmf, \
        _simulator(lambda x: x * (1 - true_p1 * x * x), dt, T)
