"""NumPy fp64 restatement of the reference's 1-D moment filter (TEST INFRASTRUCTURE, see oracle/__init__.py).

Reference files followed (paths relative to /root/reference):
  mfs/one_dim/quadtures.py:29-60,83-133   hankel_indices, moment_quadrature
  mfs/one_dim/filtering.py:32-89,92-161,164-240   moment_filter_rms / _cms / _scms
  mfs/one_dim/moments.py:31-138           normal moments and raw/central/scaled conversions
  mfs/utils.py:39-74,495-538              GaussianSum1D, ldl, ldl_chol

LAPACK via NumPy/SciPy plays the role XLA's LAPACK custom calls play in the JAX reference, so the dense
linear algebra is the same family of routines (potrf / trsm / syevd).
"""
import math
import warnings
from typing import Callable, NamedTuple, Tuple

import numpy as np
import scipy.linalg
import scipy.special


# ---------------------------------------------------------------------------------------------------------------------
# mfs/utils.py:495-538
# ---------------------------------------------------------------------------------------------------------------------
def ldl(mat: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """LDL^T of a symmetric matrix, no pivoting (mfs/utils.py:495-522)."""
    n = mat.shape[0]
    l = np.eye(n)
    l[1:, 0] = mat[1:, 0] / mat[0, 0]
    d = np.ones((n,)) * mat[0, 0]
    for j in range(1, n):
        v = l[j, :j] * d[:j]
        _d = mat[j, j] - np.dot(l[j, :j], v)
        d[j] = _d
        l[j + 1:, j] = (mat[j + 1:, j] - l[j + 1:, :j] @ v) / _d
    return l, d


def ldl_chol(mat: np.ndarray, eps: float = None) -> np.ndarray:
    """PD completion: R = L diag(d < 0 ? eps : sqrt(d)), eps = 1e-8 ||mat||_F (mfs/utils.py:525-538)."""
    if eps is None:
        eps = 1e-8 * np.linalg.norm(mat, 'fro')
    l, d = ldl(mat)
    with np.errstate(invalid='ignore'):
        return l * np.where(d < 0, eps, np.sqrt(d))[None, :]


# ---------------------------------------------------------------------------------------------------------------------
# mfs/one_dim/quadtures.py
# ---------------------------------------------------------------------------------------------------------------------
def hankel_indices(n: int) -> Tuple[np.ndarray, np.ndarray]:
    """G_idx[i, j] = i + j, H_idx = G_idx + 1 (mfs/one_dim/quadtures.py:29-60)."""
    inds = np.arange(n)[:, None] + np.arange(n)[None, :]
    return inds, inds + 1


def _cholesky_nan(G: np.ndarray) -> np.ndarray:
    """Lower Cholesky; a non-PD input gives an all-NaN factor, as XLA's potrf wrapper does."""
    G = 0.5 * (G + G.T)  # jax.lax.linalg.cholesky(symmetrize_input=True)
    if not np.all(np.isfinite(G)):
        return np.full_like(G, np.nan)
    try:
        return np.linalg.cholesky(G)
    except np.linalg.LinAlgError:
        return np.full_like(G, np.nan)


def _eigh_nan(K: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(values, vectors) of (K + K^T)/2 (jax.lax.linalg.eigh default symmetrize_input=True); NaN in -> NaN out."""
    Ks = 0.5 * (K + K.T)
    n = K.shape[0]
    if not np.all(np.isfinite(Ks)):
        return np.full((n,), np.nan), np.full((n, n), np.nan)
    vals, vecs = np.linalg.eigh(Ks)
    return vals, vecs


def moment_quadrature(ms: np.ndarray, mean: float = 0., scale: float = 1., ldl: bool = False
                      ) -> Tuple[np.ndarray, np.ndarray]:
    """Weights and nodes from 2n moments (mfs/one_dim/quadtures.py:83-133).

    G = ms[i+j], H = ms[i+j+1]; R = chol(G) lower (:127); K = R^{-1} H R^{-T} by two triangular solves (:128-129);
    eigh of the symmetrised K (:131); weights = V[0, :]**2, nodes = scale * lambda + mean (:133).
    """
    ms = np.asarray(ms, dtype=np.float64)
    n = ms.shape[0] // 2
    G_inds, H_inds = hankel_indices(n)
    G, H = ms[G_inds], ms[H_inds]

    R = ldl_chol(G) if ldl else _cholesky_nan(G)
    if not np.all(np.isfinite(R)):
        nan = np.full((n,), np.nan)
        return nan, nan.copy()
    try:
        X = scipy.linalg.solve_triangular(R, H, lower=True, check_finite=False)  # R X = H
        K = scipy.linalg.solve_triangular(R, X.T, lower=True, check_finite=False).T  # K R^T = X
    except np.linalg.LinAlgError:   # an exactly zero diagonal (ldl=True with d_j == 0): XLA divides and yields inf / NaN in-band
        nan = np.full((n,), np.nan)
        return nan, nan.copy()
    vals, vecs = _eigh_nan(K)
    return vecs[0, :] ** 2, scale * vals + mean


# ---------------------------------------------------------------------------------------------------------------------
# mfs/one_dim/moments.py:31-138
# ---------------------------------------------------------------------------------------------------------------------
def central_moment_of_normal(variance: float, p: int) -> float:
    """mfs/one_dim/moments.py:31-38."""
    if p % 2 == 0:
        return math.sqrt(variance) ** p * float(scipy.special.factorial2(p - 1, exact=True)) if p > 0 else 1.
    return 0.


def raw_moment_of_standard_normal(p: int) -> float:
    """mfs/one_dim/moments.py:41-67."""
    if p % 2 == 0:
        return math.factorial(p) / (2 ** (p / 2) * math.factorial(int(p / 2)))
    return 0.


def raw_moment_of_normal(mean, variance, p: int):
    """sum_m C(p, m) mean^m variance^((p-m)/2) E[Z^(p-m)] (mfs/one_dim/moments.py:70-74); vectorised in mean/variance."""
    mean = np.asarray(mean, dtype=np.float64)
    variance = np.asarray(variance, dtype=np.float64)
    out = np.zeros(np.broadcast(mean, variance).shape)
    for m in range(p + 1):
        z = raw_moment_of_standard_normal(p - m)
        if z == 0.:
            # the reference multiplies variance ** (odd / 2) by 0.; identical unless variance < 0 (NaN there)
            continue
        out = out + math.comb(p, m) * mean ** m * variance ** ((p - m) // 2) * z
    return out


def raw_to_central(rms: np.ndarray) -> np.ndarray:
    """c_n = sum_{j<=n} C(n, j) (-1)^(n-j) r_j r_1^(n-j) (mfs/one_dim/moments.py:86-101)."""
    s = rms.shape[0]
    bn = scipy.linalg.pascal(s, kind='lower', exact=True).astype(np.float64)
    out = np.zeros(s)
    for n in range(s):
        for j in range(n + 1):
            out[n] += bn[n, j] * (-1) ** (n - j) * rms[j] * rms[1] ** (n - j)
    return out


def central_to_raw(cms: np.ndarray, mean: float) -> np.ndarray:
    """r_n = sum_{j<=n} C(n, j) c_j mean^(n-j) (mfs/one_dim/moments.py:104-123)."""
    s = cms.shape[0]
    bn = scipy.linalg.pascal(s, kind='lower', exact=True).astype(np.float64)
    out = np.zeros(s)
    for n in range(s):
        for j in range(n + 1):
            out[n] += bn[n, j] * cms[j] * mean ** (n - j)
    return out


def raw_to_scaled(rms: np.ndarray, scale: float = None) -> np.ndarray:
    """mfs/one_dim/moments.py:126-132."""
    if scale is None:
        scale = math.sqrt(rms[2] - rms[1] ** 2)
    return raw_to_central(rms) / np.array([scale ** n for n in range(rms.shape[0])])


def scaled_to_central(sms: np.ndarray, scale: float) -> np.ndarray:
    """mfs/one_dim/moments.py:135-138."""
    return sms * np.array([scale ** n for n in range(sms.shape[0])])


# ---------------------------------------------------------------------------------------------------------------------
# mfs/utils.py:39-74
# ---------------------------------------------------------------------------------------------------------------------
class GaussianSum1D(NamedTuple):
    means: np.ndarray
    variances: np.ndarray
    weights: np.ndarray
    mean: float
    variance: float
    rms: np.ndarray
    cms: np.ndarray
    scms: np.ndarray

    @classmethod
    def new(cls, means, variances, weights, N: int = 2):
        """Mixture raw / central / scaled-central moments (mfs/utils.py:60-74)."""
        means, variances, weights = (np.asarray(a, dtype=np.float64) for a in (means, variances, weights))
        centre = float(np.sum(means * weights))
        rms = np.array([sum(float(raw_moment_of_normal(m, v, p)) * w for m, v, w in zip(means, variances, weights))
                        for p in range(2 * N)])
        cms = np.array([sum(float(raw_moment_of_normal(m - centre, v, p)) * w
                            for m, v, w in zip(means, variances, weights)) for p in range(2 * N)])
        variance = cms[2]
        scms = cms / np.sqrt(variance) ** np.arange(2 * N)
        return cls(means=means, variances=variances, weights=weights, mean=centre, variance=variance,
                   rms=rms, cms=cms, scms=scms)


# ---------------------------------------------------------------------------------------------------------------------
# mfs/one_dim/filtering.py
# ---------------------------------------------------------------------------------------------------------------------
def _pdf_vec(measurement_cond_pdf: Callable, y, nodes: np.ndarray) -> np.ndarray:
    """jax.vmap(measurement_cond_pdf, in_axes=[None, 0])(y, nodes)."""
    out = measurement_cond_pdf(y, nodes)
    out = np.asarray(out, dtype=np.float64)
    if out.shape != nodes.shape:
        out = np.array([float(measurement_cond_pdf(y, x)) for x in nodes])
    return out


def moment_filter_rms(state_cond_raw_moments: Callable, measurement_cond_pdf: Callable,
                      rms0: np.ndarray, ys: np.ndarray, stable: bool = False):
    """mfs/one_dim/filtering.py:32-89 (scan body :73-86)."""
    rms = np.asarray(rms0, dtype=np.float64).copy()
    num_moments = rms.shape[0]
    powers = np.arange(num_moments)
    if num_moments % 2 != 0:
        warnings.warn(f'The order of moments {num_moments - 1} is not odd.')
    T = len(ys)
    rmss = np.zeros((T, num_moments))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature(rms, ldl=stable)
            rms = np.einsum('ij,i->j', state_cond_raw_moments(x, powers), w)

            w, x = moment_quadrature(rms, ldl=stable)
            lik = _pdf_vec(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            rms = np.einsum('ij,i->j', x[:, None] ** powers[None, :] * lik[:, None], w) / pdf_y
            nell -= np.log(pdf_y)
            rmss[k] = rms
    return rmss, nell


def moment_filter_cms(state_cond_central_moments: Callable, state_cond_mean: Callable,
                      measurement_cond_pdf: Callable, cms0: np.ndarray, mean0: float, ys: np.ndarray,
                      stable: bool = False):
    """mfs/one_dim/filtering.py:92-161 (scan body :140-158)."""
    cms = np.asarray(cms0, dtype=np.float64).copy()
    mean = float(mean0)
    num_moments = cms.shape[0]
    orders = np.arange(num_moments)
    if num_moments % 2 != 0:
        warnings.warn(f'The order of moments {num_moments - 1} is not odd.')
    T = len(ys)
    cmss, means = np.zeros((T, num_moments)), np.zeros((T,))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature(cms, mean, ldl=stable)
            mean = np.dot(state_cond_mean(x), w)
            cms = np.einsum('ij,i->j', state_cond_central_moments(x, orders, mean), w)

            w, x = moment_quadrature(cms, mean, ldl=stable)
            lik = _pdf_vec(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            mean = np.dot(x * lik, w) / pdf_y
            cms = np.einsum('ij,i->j', (x - mean)[:, None] ** orders[None, :] * lik[:, None], w) / pdf_y
            nell -= np.log(pdf_y)
            cmss[k], means[k] = cms, mean
    return cmss, means, nell


def moment_filter_scms(state_cond_scaled_central_moments: Callable, state_cond_mean_var: Callable,
                       measurement_cond_pdf: Callable, scms0: np.ndarray, mean0: float, scale0: float,
                       ys: np.ndarray, stable: bool = False):
    """mfs/one_dim/filtering.py:164-240 (scan body :217-237)."""
    scms = np.asarray(scms0, dtype=np.float64).copy()
    mean, scale = float(mean0), float(scale0)
    num_moments = scms.shape[0]
    orders = np.arange(num_moments)
    if num_moments % 2 != 0:
        warnings.warn(f'The order of moments {num_moments - 1} is not odd.')
    T = len(ys)
    scmss, means, scales = np.zeros((T, num_moments)), np.zeros((T,)), np.zeros((T,))
    nell = 0.
    with np.errstate(all='ignore'):
        for k in range(T):
            y = ys[k]
            w, x = moment_quadrature(scms, mean, scale, ldl=stable)
            cond_means, cond_vars = state_cond_mean_var(x)
            mean, scale = np.dot(cond_means, w), np.sqrt(np.dot(cond_vars, w))
            scms = np.einsum('ij,i->j', state_cond_scaled_central_moments(x, orders, mean, scale), w)

            w, x = moment_quadrature(scms, mean, scale, ldl=stable)
            lik = _pdf_vec(measurement_cond_pdf, y, x)
            pdf_y = np.dot(lik, w)
            mean = np.dot(x * lik, w) / pdf_y
            scale = np.sqrt(np.dot((x - mean) ** 2 * lik, w) / pdf_y)
            scms = np.einsum('ij,i->j', ((x - mean) / scale)[:, None] ** orders[None, :] * lik[:, None], w) / pdf_y
            nell -= np.log(pdf_y)
            scmss[k], means[k], scales[k] = scms, mean, scale
    return scmss, means, scales, nell


# ---------------------------------------------------------------------------------------------------------------------
# mfs/one_dim/moments.py:309-337 (the "next" row f1: characteristic function from moments)
# ---------------------------------------------------------------------------------------------------------------------
def characteristic_fn(z, ms: np.ndarray, mean: float = 0., scale: float = 1.):
    """sum_n w_n exp(i z x_n) (mfs/one_dim/moments.py:309-337); vectorised over z."""
    w, x = moment_quadrature(ms, mean, scale)
    z = np.asarray(z, dtype=np.float64)
    return np.exp(1.j * z[..., None] * x) @ w
