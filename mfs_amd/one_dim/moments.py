"""Moment utilities and transition-moment factories, mirroring `mfs.one_dim.moments`.

Same names, argument order and return tuples as the reference (mfs/one_dim/moments.py); the factories return
closures that evaluate numerically on arrays (inspection) and hand the filters a device description when traced.
"""
import math
from typing import Callable, Tuple

import numpy as np
import scipy.linalg

from mfs_amd import sym
from mfs_amd.tme_poly import TransitionTables, tme_tables, euler_tables, normal_tables

__all__ = ['central_moment_of_normal', 'raw_moment_of_standard_normal', 'raw_moment_of_normal', 'raw_to_central',
           'central_to_raw', 'raw_to_scaled', 'scaled_to_central', 'sde_cond_moments_tme',
           'sde_cond_moments_tme_normal', 'sde_cond_moments_euler', 'sde_cond_moments_normal', 'characteristic_fn']


def central_moment_of_normal(variance: float, p: int) -> float:
    """p-th central moment of a Normal (mfs/one_dim/moments.py:31-38)."""
    if p % 2 == 0:
        return math.sqrt(variance) ** p * float(math.prod(range(p - 1, 0, -2)))
    return 0.


def raw_moment_of_standard_normal(p: int) -> float:
    """E[Z^p], Z ~ N(0, 1) (mfs/one_dim/moments.py:41-67)."""
    if p % 2 == 0:
        return math.factorial(p) / (2 ** (p / 2) * math.factorial(p // 2))
    return 0.


def raw_moment_of_normal(mean, variance, p: int):
    """E[X^p], X ~ N(mean, variance) (mfs/one_dim/moments.py:70-74); vectorised in mean / variance."""
    mean = np.asarray(mean, dtype=np.float64)
    variance = np.asarray(variance, dtype=np.float64)
    out = np.zeros(np.broadcast(mean, variance).shape)
    for m in range(p % 2, p + 1, 2):  # odd p - m terms vanish
        out = out + math.comb(p, m) * mean ** m * variance ** ((p - m) // 2) * raw_moment_of_standard_normal(p - m)
    return out if out.ndim else float(out)


def _pascal(s: int) -> np.ndarray:
    return scipy.linalg.pascal(s, kind='lower', exact=True).astype(np.float64)


def raw_to_central(rms) -> np.ndarray:
    """c_n = sum_{j<=n} C(n, j) (-r_1)^(n-j) r_j (mfs/one_dim/moments.py:86-101)."""
    rms = np.asarray(rms, dtype=np.float64)
    s = rms.shape[-1]
    bn = _pascal(s)
    n, j = np.meshgrid(np.arange(s), np.arange(s), indexing='ij')
    m1 = rms[..., 1:2, None] if s > 1 else np.zeros(rms.shape[:-1] + (1, 1))
    terms = bn * np.where(n >= j, (-m1) ** np.maximum(n - j, 0), 0.) * rms[..., None, :]
    return terms.sum(axis=-1)


def central_to_raw(cms, mean) -> np.ndarray:
    """r_n = sum_{j<=n} C(n, j) mean^(n-j) c_j (mfs/one_dim/moments.py:104-123)."""
    cms = np.asarray(cms, dtype=np.float64)
    s = cms.shape[-1]
    bn = _pascal(s)
    n, j = np.meshgrid(np.arange(s), np.arange(s), indexing='ij')
    mean = np.asarray(mean, dtype=np.float64)[..., None, None]
    terms = bn * np.where(n >= j, mean ** np.maximum(n - j, 0), 0.) * cms[..., None, :]
    return terms.sum(axis=-1)


def raw_to_scaled(rms, scale=None) -> np.ndarray:
    """E[((X - mean) / scale)^n] from raw moments (mfs/one_dim/moments.py:126-132)."""
    rms = np.asarray(rms, dtype=np.float64)
    if scale is None:
        scale = np.sqrt(rms[..., 2] - rms[..., 1] ** 2)
    scale = np.asarray(scale, dtype=np.float64)[..., None]
    return raw_to_central(rms) / scale ** np.arange(rms.shape[-1])


def scaled_to_central(sms, scale) -> np.ndarray:
    """mfs/one_dim/moments.py:135-138."""
    sms = np.asarray(sms, dtype=np.float64)
    return sms * np.asarray(scale, dtype=np.float64)[..., None] ** np.arange(sms.shape[-1])


# ---------------------------------------------------------------------------------------------------------------------
# transition-moment closures
# ---------------------------------------------------------------------------------------------------------------------
class TransitionRef:
    """What a traced transition closure returns: the tables plus the (mean, scale) arguments it was traced with."""

    def __init__(self, tables: TransitionTables, which: str, mean=None, scale=None):
        self.tables, self.which, self.mean, self.scale = tables, which, mean, scale


class _CondMoments:
    def __init__(self, tables: TransitionTables, which: str):
        self.tables, self.which = tables, which

    def __call__(self, x, n, mean=0., scale=1.):
        if sym.is_symbolic(x) or sym.is_symbolic(n):
            return TransitionRef(self.tables, self.which, mean, scale)
        return self.tables.cond_moments(x, n, mean, scale)


class _CondMean:
    def __init__(self, tables: TransitionTables):
        self.tables = tables

    def __call__(self, x):
        if sym.is_symbolic(x):
            return TransitionRef(self.tables, 'mean')
        return self.tables.cond_mean(x)


class _CondMeanVar:
    def __init__(self, tables: TransitionTables):
        self.tables = tables

    def __call__(self, x):
        if sym.is_symbolic(x):
            return TransitionRef(self.tables, 'mean_var')
        return self.tables.cond_mean(x), self.tables.cond_var(x)


def _five(tables: TransitionTables):
    return (_CondMoments(tables, 'raw'), _CondMoments(tables, 'central'), _CondMoments(tables, 'scaled'),
            _CondMean(tables), _CondMeanVar(tables))


def _trace_sde(drift: Callable, dispersion: Callable) -> Tuple[sym.Poly, sym.Poly]:
    a = sym.as_poly(drift(sym.X), 'drift')
    b = sym.as_poly(dispersion(sym.X), 'dispersion')
    return a, b


def sde_cond_moments_tme(drift: Callable, dispersion: Callable, dt: float, tme_order: int):
    """Conditional moments by TME expansion (mfs/one_dim/moments.py:141-179).

    Returns (state_cond_raw_moments, state_cond_central_moments, state_cond_scaled_central_moments, state_cond_mean,
    state_cond_mean_var), as the reference does.  `drift` / `dispersion` are traced once with `mfs_amd.sym.X`.
    """
    a, b = _trace_sde(drift, dispersion)
    return _five(tme_tables(a, b, float(dt), int(tme_order), gaussian=False))


def sde_cond_moments_tme_normal(drift: Callable, dispersion: Callable, dt: float, tme_order: int, N: int):
    """TME mean / variance + Normal closure (mfs/one_dim/moments.py:182-219).  The scaled variant normalises by
    scale**n (the reference's prod(scale**arange) at :205-207 is an untested quirk)."""
    a, b = _trace_sde(drift, dispersion)
    return _five(tme_tables(a, b, float(dt), int(tme_order), gaussian=True))


def sde_cond_moments_euler(drift: Callable, dispersion: Callable, dt: float, N: int):
    """Euler--Maruyama + Normal closure (mfs/one_dim/moments.py:222-255)."""
    a, b = _trace_sde(drift, dispersion)
    return _five(euler_tables(a, b, float(dt)))


def sde_cond_moments_normal(cond_mean: Callable, cond_var: Callable, N: int = None):
    """Normal transition X' | x ~ N(cond_mean(x), cond_var(x)) given directly, e.g. the exact OU discretisation
    `lambda x: F * x`, `lambda x: Sigma` of dardel/convergence/convergence_mf.py:86-107 (extension: the reference
    writes these closures by hand with raw_moment_of_normal)."""
    m = sym.as_poly(cond_mean(sym.X), 'cond_mean')
    v = sym.as_poly(cond_var(sym.X), 'cond_var')
    return _five(normal_tables(m, v))


def characteristic_fn(z, ms, mean=0., scale=1., *, device: int = 0):
    """Characteristic function computed from moments, E[exp(i z X)] ~= sum_n w_n exp(i z x_n)
    (mfs/one_dim/moments.py:309-337), on the device.  Vectorised the way the reference's post-processing vmaps it
    (dardel/benes_bernoulli/post_processing_mf.py:37-60): `z` scalar or (m,), `ms` (..., 2N), `mean` / `scale` scalars or
    (...,); returns complex128 of shape ms.shape[:-1] + z.shape."""
    from mfs_amd import _lib
    z_arr = np.ascontiguousarray(np.atleast_1d(np.asarray(z, dtype=np.float64)))
    ms = np.asarray(ms, dtype=np.float64)
    lead = ms.shape[:-1]
    ms2 = np.ascontiguousarray(ms.reshape(-1, ms.shape[-1]))
    count, M2 = ms2.shape
    N = M2 // 2
    if M2 % 2 or not 2 <= N <= _lib.MAX_N:
        raise ValueError(f'need 2N moments with 2 <= N <= {_lib.MAX_N}, got {M2}')
    mean_a = np.ascontiguousarray(np.broadcast_to(np.asarray(mean, dtype=np.float64), lead).reshape(-1))
    scale_a = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), lead).reshape(-1))
    out = np.empty((count, z_arr.shape[0]), dtype=np.complex128)
    _lib.check(_lib.lib().mfs_characteristic_1d(N, count, _lib.ptr(ms2), _lib.ptr(mean_a), _lib.ptr(scale_a),
                                                z_arr.shape[0], _lib.ptr(z_arr), out.ctypes.data_as(_lib.C.c_void_p),
                                                device, None))
    out = out.reshape(lead + z_arr.shape)
    return out[..., 0] if np.ndim(z) == 0 else out
