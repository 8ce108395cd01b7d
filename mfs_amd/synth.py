"""Seeded synthetic measurement generators for the benchmark models (NumPy; host side, outside the timed path).

The reference draws trajectories with JAX PRNG streams (mfs/one_dim/ss_models.py:49-54,86-91,
mfs/multi_dims/ss_models.py:69-93, dardel/benes_bernoulli/mf.py:74-80) which cannot be reproduced without JAX, so the
build uses `numpy.random.default_rng(seed)`; all generators are vectorised over the replicate axis B.
"""
import math

import numpy as np


def _mixture_x0(rng, B, means=(-0.5, 0.5), variances=(0.05, 0.05), weights=(0.5, 0.5)):
    cs = rng.choice(len(means), size=B, p=np.asarray(weights))
    return np.asarray(means)[cs] + np.sqrt(np.asarray(variances))[cs] * rng.standard_normal(B)


def _euler_path(rng, x0, drift, T, dt, substeps):
    """Euler--Maruyama with unit dispersion, `substeps` sub-steps per measurement interval; returns (B, T)."""
    B = x0.shape[0]
    ddt = dt / substeps
    sq = math.sqrt(ddt)
    x = x0.copy()
    xs = np.empty((B, T))
    for k in range(T):
        dws = rng.standard_normal((substeps, B)) * sq
        for j in range(substeps):
            x = x + drift(x) * ddt + dws[j]
        xs[:, k] = x
    return xs


def benes_bernoulli_batch(B: int, T: int, dt: float = 1e-2, seed: int = 0, substeps: int = 10, slope: float = 5.):
    """ys[b, k] ~ Bernoulli(1 / (1 + exp(-x^3 / slope))) along a Benes path dx = tanh(x) dt + dW. Returns (ys, xs)."""
    rng = np.random.default_rng(seed)
    xs = _euler_path(rng, _mixture_x0(rng, B), np.tanh, T, dt, substeps)
    with np.errstate(over='ignore'):  # |x| ~ 10 at T = 1000: exp overflows to inf, p -> 0 exactly as intended
        p = 1. / (1. + np.exp(-xs ** 3 / slope))
    ys = (rng.random((B, T)) < p).astype(np.float64)
    return ys, xs


def well_poisson_batch(B: int, T: int, p1=3., p2=3., dt: float = 1e-2, seed: int = 0, substeps: int = 10):
    """ys[b, k] ~ Poisson(log(1 + exp(p2 x))) along dx = x (1 - p1 x^2) dt + dW; p1, p2 scalars or (B,). (ys, xs)."""
    rng = np.random.default_rng(seed)
    p1 = np.broadcast_to(np.asarray(p1, dtype=np.float64), (B,))
    p2 = np.broadcast_to(np.asarray(p2, dtype=np.float64), (B,))
    xs = _euler_path(rng, _mixture_x0(rng, B), lambda x: x * (1. - p1 * x * x), T, dt, substeps)
    ys = rng.poisson(np.log1p(np.exp(p2[:, None] * xs))).astype(np.float64)
    return ys, xs


def ou_gaussian_batch(B: int, T: int, dt: float = 1e-1, ell: float = 1., sigma: float = 0.5, R: float = 1.,
                      mean0: float = 0., var0: float = None, seed: int = 0):
    """Exact OU transitions + Gaussian measurements (dardel/convergence/convergence_mf.py:32-61). (ys, xs)."""
    rng = np.random.default_rng(seed)
    if var0 is None:
        var0 = sigma ** 2
    F, Sigma = math.exp(-dt / ell), sigma ** 2 * (1 - math.exp(-2 * dt / ell))
    x = mean0 + math.sqrt(var0) * rng.standard_normal(B)
    xs = np.empty((B, T))
    for k in range(T):
        x = F * x + math.sqrt(Sigma) * rng.standard_normal(B)
        xs[:, k] = x
    ys = xs + math.sqrt(R) * rng.standard_normal((B, T))
    return ys, xs


def prey_predator_batch(B: int, T: int, dt: float = 1e-3, seed: int = 0, substeps: int = 20,
                        alp=4., beta=4., delta=4., gamma=4., sigma=0.1):
    """Milstein path of the prey--predator SDE (mfs/multi_dims/ss_models.py:57-61,76-92) and Bernoulli measurements
    with p = 1 / (1 + exp(-x_0^3 + 1)) (:63-67). Returns (ys (B, T), xs (B, T, 2))."""
    rng = np.random.default_rng(seed)
    cs = rng.integers(0, 2, size=B)
    std = np.sqrt(np.where(cs == 0, 1e-3, 2e-3))
    x = 1. + std[:, None] * rng.standard_normal((B, 2))
    ddt = dt / substeps
    sq = math.sqrt(ddt)
    xs = np.empty((B, T, 2))
    for k in range(T):
        for _ in range(substeps):
            ddw = sq * rng.standard_normal((B, 2))
            drift = x * (x[:, ::-1] * np.array([-beta, delta]) + np.array([alp, -gamma]))
            x = x + drift * ddt + sigma * x * ddw + 0.5 * sigma ** 2 * x * (ddw ** 2 - ddt)
        xs[:, k] = x
    p = 1. / (1. + np.exp(-xs[:, :, 0] ** 3 + 1.))
    ys = (rng.random((B, T)) < p).astype(np.float64)
    return ys, xs
