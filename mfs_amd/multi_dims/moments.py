"""N-D moment utilities and the TME transition factory, mirroring `mfs.multi_dims.moments`."""
import itertools
import math
from typing import Callable, Sequence

import numpy as np
from scipy.special import comb as _comb, factorial as _factorial

from mfs_amd import sym
from mfs_amd.multi_dims.multi_indices import find_indices
from mfs_amd.tme_poly_nd import (TransitionTablesND, GaussianTablesND, BatchedTablesND, tme_tables_nd,
                                  normal_tables_nd)

__all__ = ['cond_moments_linear_gaussian', 'raw_moments_mvn_kan', 'central_moments_mvn_kan', 'sde_cond_moments_tme', 'sde_cond_moments_tme_normal',
           'sde_cond_moments_euler_maruyama', 'batch_closures', 'extract_moments',
           'extract_mean', 'extract_cov', 'marginalise_moments']


def raw_moments_mvn_kan(mean, cov, multi_index: Sequence[int]) -> float:
    """E[X^n], X ~ N(mean, cov): Kan (2008), Proposition 2 (mfs/multi_dims/moments.py:110-154)."""
    mean, cov = np.asarray(mean, dtype=np.float64), np.asarray(cov, dtype=np.float64)
    n = np.asarray(multi_index, dtype='int64')
    s = int(n.sum())
    grid = np.asarray(tuple(itertools.product(*[range(int(k) + 1) for k in n], range(s // 2 + 1))), dtype='int64')
    vs, rs = grid[:, :-1], grid[:, -1]
    hs = n / 2 - vs
    quad = np.einsum('ij,jk,ik->i', hs, cov, hs) / 2
    terms = (-1.) ** vs.sum(axis=1) * np.prod(_comb(n, vs), axis=1) * quad ** rs * (hs @ mean) ** (s - 2 * rs) \
        / (_factorial(rs) * _factorial(s - 2 * rs))
    return float(terms.sum())


def central_moments_mvn_kan(cov, multi_index: Sequence[int]) -> float:
    """E[X^n], X ~ N(0, cov): Kan (2008), Proposition 1 (mfs/multi_dims/moments.py:66-107)."""
    n = np.asarray(multi_index, dtype='int64')
    if int(n.sum()) % 2:
        return 0.
    return raw_moments_mvn_kan(np.zeros(n.shape[0]), cov, n)


def extract_moments(ms, multi_index):
    """mfs/multi_dims/moments.py:184-200."""
    return np.asarray(ms)[..., find_indices(np.asarray(multi_index))]


def extract_mean(rms, d: int) -> np.ndarray:
    """mfs/multi_dims/moments.py:203-226."""
    return np.stack([extract_moments(rms, np.eye(d, dtype=int)[i]) for i in range(d)], axis=-1)


def extract_cov(ms, d: int) -> np.ndarray:
    """mfs/multi_dims/moments.py:229-254."""
    eye = np.eye(d, dtype=int)
    rows = [np.stack([extract_moments(ms, eye[i] + eye[j]) for j in range(d)], axis=-1) for i in range(d)]
    return np.stack(rows, axis=-2)


def marginalise_moments(ms, d: int, N: int, var_axis: int):
    """Marginal moments of one coordinate (mfs/multi_dims/moments.py:482-504)."""
    mi = np.zeros((2 * N, d), dtype='int64')
    mi[:, var_axis] = np.arange(2 * N)
    return np.asarray(ms)[..., find_indices(mi)]


class TransitionRefND:
    def __init__(self, tables: TransitionTablesND, which: str, mean=None, scale=None):
        self.tables, self.which, self.mean, self.scale = tables, which, mean, scale


class _CondMomentsND:
    def __init__(self, tables, which):
        self.tables, self.which = tables, which

    def __call__(self, x, multi_indices, mean=None, scale=None):
        if x is sym.X or sym.is_symbolic(x):
            return TransitionRefND(self.tables, self.which, mean, scale)
        return self.tables.cond_moments(x, multi_indices, mean, scale)


class _CondMeanND:
    def __init__(self, tables, with_var=False):
        self.tables, self.with_var = tables, with_var

    def __call__(self, x):
        if x is sym.X or sym.is_symbolic(x):
            return TransitionRefND(self.tables, 'mean_var' if self.with_var else 'mean')
        return (self.tables.cond_mean(x), self.tables.cond_var(x)) if self.with_var else self.tables.cond_mean(x)


def sde_cond_moments_tme(drift: Callable, dispersion: Callable, dt: float, tme_order: int, d: int = 2):
    """TME conditional moments without Normal closure (mfs/multi_dims/moments.py:414-479).

    Returns (state_cond_raw_moments, state_cond_central_moments, state_cond_scaled_central_moments, state_cond_mean,
    state_cond_mean_var); the moment closures take the 'multi-index' signature `(x (..., d), multi_indices (z, d), ...)`.
    `drift(x)` / `dispersion(x)` are traced once with an object array of polynomial variables (`d` gives the state
    dimension, which JAX infers from the array it traces with).
    """
    tables = tme_tables_nd(drift, dispersion, d, float(dt), int(tme_order))
    return (_CondMomentsND(tables, 'raw'), _CondMomentsND(tables, 'central'), _CondMomentsND(tables, 'scaled'),
            _CondMeanND(tables), _CondMeanND(tables, with_var=True))


class _CondMomentsNDIndexed:
    """'index' signature of the Normal-closure factories: the second argument selects rows of the moment table the
    factory was built for (mfs/multi_dims/moments.py:293-300: `rms[index]`)."""

    def __init__(self, tables, which, multi_indices):
        self.tables, self.which, self.multi_indices = tables, which, np.asarray(multi_indices)

    def __call__(self, x, index, mean=None, scale=None):
        if x is sym.X or sym.is_symbolic(x):
            return TransitionRefND(self.tables, self.which, mean, scale)
        return self.tables.cond_moments(x, self.multi_indices[np.asarray(index, dtype=int)], mean, scale)


def _indexed_five(tables, multi_indices):
    return (_CondMomentsNDIndexed(tables, 'raw', multi_indices), _CondMomentsNDIndexed(tables, 'central', multi_indices),
            _CondMomentsNDIndexed(tables, 'scaled', multi_indices), _CondMeanND(tables), _CondMeanND(tables, with_var=True))


def sde_cond_moments_tme_normal(drift: Callable, dispersion: Callable, dt: float, tme_order: int, multi_indices):
    """Normal closure with TME mean / covariance (mfs/multi_dims/moments.py:340-411); 'index' signature."""
    mi = np.asarray(multi_indices)
    return _indexed_five(normal_tables_nd(drift, dispersion, mi.shape[-1], float(dt), int(tme_order)), mi)


def sde_cond_moments_euler_maruyama(drift: Callable, dispersion: Callable, dt: float, multi_indices):
    """Euler--Maruyama Normal closure (mfs/multi_dims/moments.py:257-337); 'index' signature."""
    mi = np.asarray(multi_indices)
    return _indexed_five(normal_tables_nd(drift, dispersion, mi.shape[-1], float(dt), 'euler'), mi)


def cond_moments_linear_gaussian(F, Q, multi_indices):
    """Normal closure of an exactly discretised linear SDE, X' | x ~ N(F x, Q) -- what
    /root/reference/examples/2d_bearing_only.ipynb cell 7 writes by hand with `discretise_lti_sde` and `raw_moments_mvn_kan`:
    `cms[index]` of E[(X' - mean)^n | x] for the multi-indices of the table; 'index' signature, same five closures as
    `sde_cond_moments_tme_normal`."""
    from mfs_amd.tme_poly_nd import GaussianTablesND, PolyND
    mi = np.asarray(multi_indices)
    F, Q = np.asarray(F, dtype=np.float64), np.asarray(Q, dtype=np.float64)
    d = mi.shape[-1]
    if F.shape != (d, d) or Q.shape != (d, d):
        raise ValueError(f'F and Q must be ({d}, {d})')
    xs = [PolyND.variable(d, k) for k in range(d)]
    mean = []
    for i in range(d):
        acc = PolyND(np.float64(0.), d)
        for k in range(d):
            acc = acc + float(F[i, k]) * xs[k]
        mean.append(acc.trimmed())
    cov = [[PolyND(np.float64(0.5 * (Q[i, j] + Q[j, i])), d) for j in range(d)] for i in range(d)]
    return _indexed_five(GaussianTablesND(d, mean, cov, 'linear_gaussian'), mi)


def batch_closures(per_replicate):
    """Stack B closure tuples of one factory (one per replicate, e.g. one per theta of a parameter grid) into a single
    tuple whose transition tables carry a leading replicate axis -- the N-D counterpart of calling the 1-D factories
    with array-valued parameters.  The filters then need `ys` of shape (B, T[, ny])."""
    per_replicate = list(per_replicate)
    tables = BatchedTablesND([fns[0].tables for fns in per_replicate])
    first = per_replicate[0]
    if isinstance(first[0], _CondMomentsNDIndexed):
        return _indexed_five(tables, first[0].multi_indices)
    return (_CondMomentsND(tables, 'raw'), _CondMomentsND(tables, 'central'), _CondMomentsND(tables, 'scaled'),
            _CondMeanND(tables), _CondMeanND(tables, with_var=True))
