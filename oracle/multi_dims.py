"""NumPy fp64 restatement of the reference's N-D moment filter (TEST INFRASTRUCTURE, see oracle/__init__.py).

Reference files followed (paths relative to /root/reference):
  mfs/multi_dims/multi_indices.py:25-229   graded-lex multi-indices, Gram/Hankel gather tables
  mfs/multi_dims/quadratures.py:29-87,120-178   Cartesian products, moment_quadrature_nd
  mfs/multi_dims/filtering.py:33-207,210-280,283-344   moment_filter_nd_scms / _cms / _rms
  mfs/multi_dims/moments.py:66-154         Kan--Magnus moments of multivariate normals
  mfs/utils.py:77-125                      GaussianSumND
  mfs/multi_dims/ss_models.py:40-67        prey_predator
"""
import itertools
import math
from typing import Callable, NamedTuple, Sequence, Tuple

import numpy as np
import scipy.linalg
import sympy as sp
from scipy.special import comb as scipy_comb, factorial as scipy_factorial

from oracle.one_dim import _cholesky_nan, _eigh_nan, ldl_chol
from oracle.models import bernoulli_pmf


# ---------------------------------------------------------------------------------------------------------------------
# mfs/multi_dims/multi_indices.py
# ---------------------------------------------------------------------------------------------------------------------
def sizeof_multi_indices(d: int, upper_sum: int, lower_sum: int = 0) -> int:
    """#{x in N^d : lower <= |x| <= upper} (multi_indices.py:25-59)."""
    if upper_sum == lower_sum:
        return math.comb(upper_sum + d - 1, upper_sum)
    if upper_sum < lower_sum:
        return 0
    if lower_sum == 0:
        return math.comb(upper_sum + d, upper_sum)
    return math.comb(upper_sum + d, upper_sum) - math.comb(lower_sum - 1 + d, lower_sum - 1)


def graded_lexico_indexof_multi_index(multi_index: Sequence[int], lower_sum: int = 0) -> int:
    """Position of a multi-index in graded lexicographic order (multi_indices.py:61-112)."""
    d = len(multi_index)
    total_sum = int(sum(multi_index))
    pos = sizeof_multi_indices(d, total_sum - 1, 0)
    sub_sum = total_sum
    for i in range(d):
        ith = int(multi_index[i])
        if ith >= 1:
            pos += sizeof_multi_indices(d - (i + 1), sub_sum, sub_sum - ith + 1)
        sub_sum -= ith
    if lower_sum != 0:
        return pos - sizeof_multi_indices(d, lower_sum - 1)
    return pos


def generate_graded_lexico_multi_indices(d: int, upper_sum: int, lower_sum: int = 0) -> np.ndarray:
    """All multi-indices with lower <= |x| <= upper in graded-lex order, (z, d) int64 (multi_indices.py:139-177).

    Built directly (degree by degree, each degree in lexicographic order of the tuple) rather than by the
    reference's successor iteration; `tests/test_oracle_multi_dims.py` checks it against golden tables generated
    from the reference module itself.
    """
    if d == 1:
        return np.arange(lower_sum, upper_sum + 1).reshape((upper_sum - lower_sum + 1, 1))
    rows = []
    for s in range(lower_sum, upper_sum + 1):
        deg = [t for t in itertools.product(range(s + 1), repeat=d) if sum(t) == s]
        deg.sort()
        rows.extend(deg)
    return np.asarray(rows, dtype='int64')


def find_indices(multi_indices: np.ndarray) -> np.ndarray:
    """Vectorised graded_lexico_indexof_multi_index over the leading axes (multi_indices.py:180-182)."""
    mi = np.asarray(multi_indices)
    flat = mi.reshape(-1, mi.shape[-1])
    out = np.array([graded_lexico_indexof_multi_index(row) for row in flat], dtype='int64')
    return out.reshape(mi.shape[:-1])


def gram_and_hankel_indices_graded_lexico(N: int, d: int) -> np.ndarray:
    """(d + 1, s, s) gather tables: inds[0] Gram, inds[1 + k] multiplication by x_k (multi_indices.py:185-229)."""
    s = math.comb(N - 1 + d, N - 1)
    inds = np.zeros((d + 1, s, s), dtype='int64')
    basis = generate_graded_lexico_multi_indices(d, upper_sum=N - 1, lower_sum=0)
    gram = basis[:, None, :] + basis[None, :, :]
    inds[0] = find_indices(gram)
    for i in range(d):
        gram[:, :, i] += 1
        inds[i + 1] = find_indices(gram)
        gram[:, :, i] -= 1
    return inds


# ---------------------------------------------------------------------------------------------------------------------
# mfs/multi_dims/quadratures.py
# ---------------------------------------------------------------------------------------------------------------------
def nd_cartesian_prod_indices(d: int, n: int) -> np.ndarray:
    """(n**d, d) index combinations, last axis fastest (quadratures.py:29-48)."""
    return np.asarray(tuple(itertools.product(*(d * [list(range(n))]))), dtype='int64')


def moment_quadrature_nd(ms: np.ndarray, inds: np.ndarray, mean: np.ndarray = None, scale: np.ndarray = None,
                         ldl: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """N-D Gauss quadrature (quadratures.py:120-178).

    G = ms[inds[0]], H_k = ms[inds[1 + k]] (:151-152); one Cholesky (:154); K_k = R^{-1} H_k R^{-T} (:156-161);
    d symmetric eigensolves (:163); all s^d combinations (:165-167) with weight
    prod_{k<d-1} <v^(k)_{i_k}, v^(k+1)_{i_{k+1}}> * v^(0)_{i_0}[0] * v^(d-1)_{i_{d-1}}[0] (:169-170).
    """
    ms = np.asarray(ms, dtype=np.float64)
    d, n = inds.shape[0] - 1, inds.shape[1]
    G = ms[inds[0]]
    Hs = ms[inds[1:]]
    r = n ** d
    R = ldl_chol(G) if ldl else _cholesky_nan(G)
    if not np.all(np.isfinite(R)):
        return np.full((r,), np.nan), np.full((r, d), np.nan)
    eigvals = np.zeros((d, n))
    eigvecs = np.zeros((d, n, n))
    for k in range(d):
        try:
            X = scipy.linalg.solve_triangular(R, Hs[k], lower=True, check_finite=False)
            K = scipy.linalg.solve_triangular(R, X.T, lower=True, check_finite=False).T
        except np.linalg.LinAlgError:   # exactly zero diagonal of a completed factor: inf / NaN in-band upstream
            return np.full((r,), np.nan), np.full((r, d), np.nan)
        eigvals[k], eigvecs[k] = _eigh_nan(K)
    combs = nd_cartesian_prod_indices(d, n)
    # combs_eigvectors[c, row, k] = eigvecs[k, row, combs[c, k]]; combs_eigvals[c, k] = eigvals[k, combs[c, k]]
    cev = np.stack([eigvecs[k][:, combs[:, k]].T for k in range(d)], axis=-1)  # (r, n, d)
    cvals = np.stack([eigvals[k][combs[:, k]] for k in range(d)], axis=-1)  # (r, d)
    weights = np.prod(np.einsum('ijk,ijk->ik', cev[:, :, :-1], cev[:, :, 1:]), axis=1) * cev[:, 0, 0] * cev[:, 0, -1]
    if mean is None:
        return weights, cvals
    if scale is None:
        return weights, cvals + mean
    return weights, cvals * scale + mean


# ---------------------------------------------------------------------------------------------------------------------
# mfs/multi_dims/moments.py:66-154, mfs/utils.py:77-125
# ---------------------------------------------------------------------------------------------------------------------
def raw_moments_mvn_kan(mean: np.ndarray, cov: np.ndarray, multi_index: Sequence[int]) -> float:
    """E[X^n], X ~ N(mean, cov), by Kan (2008) Proposition 2 (moments.py:110-154)."""
    multi_index = np.asarray(multi_index, dtype='int64')
    s = int(multi_index.sum())
    ranges = [tuple(range(int(sn) + 1)) for sn in multi_index] + [tuple(range(int(s / 2) + 1))]
    vs_and_r = np.asarray(tuple(itertools.product(*ranges)), dtype='int64')
    vs, rs = vs_and_r[:, :-1], vs_and_r[:, -1]
    hs = multi_index / 2 - vs
    signs = (-1.) ** np.sum(vs, axis=1)
    combs = np.prod(scipy_comb(multi_index, vs), axis=1)
    quad = (hs[:, None, :] @ cov @ hs[:, :, None] / 2).ravel() ** rs * (hs @ mean) ** (s - 2 * rs) \
        / (scipy_factorial(rs, exact=False) * scipy_factorial(s - 2 * rs, exact=False))
    return float(np.einsum('i,i,i', signs, combs, quad))


class GaussianSumND(NamedTuple):
    d: int
    means: np.ndarray
    covs: np.ndarray
    weights: np.ndarray
    mean: np.ndarray
    cov: np.ndarray
    rms: np.ndarray
    cms: np.ndarray

    @classmethod
    def new(cls, means, covs, weights, multi_indices):
        """Mixture raw / central moments for every multi-index (mfs/utils.py:107-125)."""
        means, covs, weights = (np.asarray(a, dtype=np.float64) for a in (means, covs, weights))
        d = means.shape[1]
        centre = np.sum(means * weights[:, None], axis=0)
        cov = sum(w * (c + np.outer(m, m)) for m, c, w in zip(means, covs, weights)) - np.outer(centre, centre)
        rms = sum(w * np.array([raw_moments_mvn_kan(m, c, mi) for mi in multi_indices])
                  for m, c, w in zip(means, covs, weights))
        cms = sum(w * np.array([raw_moments_mvn_kan(m - centre, c, mi) for mi in multi_indices])
                  for m, c, w in zip(means, covs, weights))
        return cls(d=d, means=means, covs=covs, weights=weights, mean=centre, cov=cov, rms=rms, cms=cms)


# ---------------------------------------------------------------------------------------------------------------------
# mfs/multi_dims/filtering.py
# ---------------------------------------------------------------------------------------------------------------------
def _pdf_vec_nd(measurement_cond_pdf: Callable, y, nodes: np.ndarray) -> np.ndarray:
    return np.array([float(measurement_cond_pdf(y, x)) for x in nodes])


def _state_indices(multi_indices, signature):
    return multi_indices if signature == 'multi-index' else np.arange(multi_indices.shape[0])


def moment_filter_nd_rms(state_cond_raw_moments, measurement_cond_pdf, ys, moments_partial_order, rms0,
                         stable: bool = False):
    """mfs/multi_dims/filtering.py:283-344 (scan body :326-341)."""
    multi_indices, inds = moments_partial_order
    multi_indices = np.asarray(multi_indices)
    if multi_indices.shape[0] != rms0.shape[0]:
        raise ValueError(f'The size of multi_indices {multi_indices.shape[0]} must match that of cms0 {rms0.shape[0]}.')
    fn, signature = state_cond_raw_moments
    sidx = _state_indices(multi_indices, signature)
    rms = np.asarray(rms0, dtype=np.float64).copy()
    T = len(ys)
    rmss = np.zeros((T, rms.shape[0]))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature_nd(rms, inds, ldl=stable)
            rms = np.einsum('ij,i->j', fn(x, sidx), w)
            w, x = moment_quadrature_nd(rms, inds, ldl=stable)
            lik = _pdf_vec_nd(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            integrand = np.prod(x[:, None, :] ** multi_indices[None, :, :], axis=-1) * lik[:, None]
            rms = np.einsum('ij,i->j', integrand, w) / pdf_y
            nell -= np.log(pdf_y)
            rmss[k] = rms
    return rmss, nell


def moment_filter_nd_cms(state_cond_central_moments, state_cond_mean, measurement_cond_pdf, ys,
                         moments_partial_order, cms0, mean0, stable: bool = False):
    """mfs/multi_dims/filtering.py:210-280 (scan body :258-277)."""
    multi_indices, inds = moments_partial_order
    multi_indices = np.asarray(multi_indices)
    if multi_indices.shape[0] != cms0.shape[0]:
        raise ValueError(f'The size of multi_indices {multi_indices.shape[0]} must match that of cms0 {cms0.shape[0]}.')
    fn, signature = state_cond_central_moments
    sidx = _state_indices(multi_indices, signature)
    cms = np.asarray(cms0, dtype=np.float64).copy()
    mean = np.asarray(mean0, dtype=np.float64).copy()
    d = multi_indices.shape[-1]
    T = len(ys)
    cmss, means = np.zeros((T, cms.shape[0])), np.zeros((T, d))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature_nd(cms, inds, mean, ldl=stable)
            mean = np.einsum('ij,i->j', state_cond_mean(x), w)
            cms = np.einsum('ij,i->j', fn(x, sidx, mean), w)
            w, x = moment_quadrature_nd(cms, inds, mean, ldl=stable)
            lik = _pdf_vec_nd(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            mean = np.einsum('ij,i->j', x * lik[:, None], w) / pdf_y
            integrand = np.prod((x - mean)[:, None, :] ** multi_indices[None, :, :], axis=-1) * lik[:, None]
            cms = np.einsum('ij,i->j', integrand, w) / pdf_y
            nell -= np.log(pdf_y)
            cmss[k], means[k] = cms, mean
    return cmss, means, nell


def moment_filter_nd_scms(state_cond_scms, state_cond_mean_vars, measurement_cond_pdf, ys, moments_partial_order,
                          scms0, mean0, scale0, stable: bool = False):
    """mfs/multi_dims/filtering.py:33-207 (scan body :181-204)."""
    multi_indices, inds = moments_partial_order
    multi_indices = np.asarray(multi_indices)
    if multi_indices.shape[0] != scms0.shape[0]:
        raise ValueError(f'The size of multi_indices {multi_indices.shape[0]} '
                         f'must match that of cms0 {scms0.shape[0]}.')
    fn, signature = state_cond_scms
    sidx = _state_indices(multi_indices, signature)
    scms = np.asarray(scms0, dtype=np.float64).copy()
    mean = np.asarray(mean0, dtype=np.float64).copy()
    scale = np.asarray(scale0, dtype=np.float64).copy()
    d = multi_indices.shape[-1]
    T = len(ys)
    scmss, means, scales = np.zeros((T, scms.shape[0])), np.zeros((T, d)), np.zeros((T, d))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature_nd(scms, inds, mean, scale, ldl=stable)
            cond_means, cond_vars = state_cond_mean_vars(x)
            mean = np.einsum('ij,i->j', cond_means, w)
            scale = np.sqrt(np.einsum('ij,i->j', cond_vars, w))
            scms = np.einsum('ij,i->j', fn(x, sidx, mean, scale), w)
            w, x = moment_quadrature_nd(scms, inds, mean, scale, ldl=stable)
            lik = _pdf_vec_nd(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            mean = np.einsum('ij,i->j', x * lik[:, None], w) / pdf_y
            scale = np.sqrt(np.einsum('ij,i->j', (x - mean) ** 2 * lik[:, None], w) / pdf_y)
            integrand = np.prod(((x - mean) / scale)[:, None, :] ** multi_indices[None, :, :], axis=-1) * lik[:, None]
            scms = np.einsum('ij,i->j', integrand, w) / pdf_y
            nell -= np.log(pdf_y)
            scmss[k], means[k], scales[k] = scms, mean, scale
    return scmss, means, scales, nell


# ---------------------------------------------------------------------------------------------------------------------
# mfs/multi_dims/ss_models.py:40-67
# ---------------------------------------------------------------------------------------------------------------------
def prey_predator(multi_indices):
    dt = 1e-3
    T = 2000
    alp, beta, delta, gamma, sigma = 4., 4., 4., 4., 0.1
    means = np.array([[1., 1.], [1., 1.]])
    covs = np.array([[[1., 0.], [0., 1.]], [[2., 0.], [0., 2.]]]) * 0.001
    weights = np.array([0.5, 0.5])
    gs = GaussianSumND.new(means, covs, weights, multi_indices)

    def drift(x):  # list of 2 SymPy symbols (tme_sympy) -> list of 2 expressions
        return [x[0] * (x[1] * (-beta) + alp), x[1] * (x[0] * delta - gamma)]

    def dispersion(x):
        return [[sigma * x[0], 0], [0, sigma * x[1]]]

    def emission(x):
        return 1. / (1. + np.exp(-x ** 3 + 1.))

    def measurement_cond_pmf(y, x):
        return bernoulli_pmf(y, emission(x[0]))

    return dt, T, gs, drift, dispersion, emission, measurement_cond_pmf
