"""The 1-D moment filters on MI355X, mirroring `mfs.one_dim.filtering`.

Same names, positional order and return tuples as the reference (mfs/one_dim/filtering.py:32-36, 92-98, 164-172).
The time-step loop runs in hand-written HIP (mfs_amd/csrc/filter1d_kernel.hpp) through the C ABI of
include/mfs_hip.h.  Extension over the reference: `ys` may carry a leading replicate axis (B, T), initial moments may
be (2N,) shared or (B, 2N), and model parameters may be per-replicate arrays; outputs gain the same leading axis.

There is no CPU fallback: a callable that cannot be reduced to device tables raises `NotDeviceDescribable`, and a
missing libmfs_hip.so raises `MfsError`.
"""
import ctypes as C
import warnings
from typing import Callable, Tuple

import numpy as np

from mfs_amd import _lib, sym
from mfs_amd.one_dim.moments import TransitionRef
from mfs_amd.tme_poly import TransitionTables

__all__ = ['moment_filter_rms', 'moment_filter_cms', 'moment_filter_scms', 'trace_model', 'build_model_struct']


# ---------------------------------------------------------------------------------------------------------------------
# tracing the reference-style callables into device tables
# ---------------------------------------------------------------------------------------------------------------------
def _is_zero(v):
    return not sym.is_symbolic(v) and np.all(np.asarray(v) == 0.)


def _is_one(v):
    return not sym.is_symbolic(v) and np.all(np.asarray(v) == 1.)


def _trace_transition(fn: Callable, mode: str) -> TransitionTables:
    if mode == 'raw':
        ref = fn(sym.X, sym.ORDER)
        ok = isinstance(ref, TransitionRef) and _is_zero(ref.mean) and _is_one(ref.scale)
    elif mode == 'central':
        ref = fn(sym.X, sym.ORDER, sym.MEAN)
        ok = isinstance(ref, TransitionRef) and ref.mean is sym.MEAN and _is_one(ref.scale)
    else:
        ref = fn(sym.X, sym.ORDER, sym.MEAN, sym.SCALE)
        ok = isinstance(ref, TransitionRef) and ref.mean is sym.MEAN and ref.scale is sym.SCALE
    if not isinstance(ref, TransitionRef):
        raise sym.NotDeviceDescribable(
            f'the {mode} transition-moment callable returned {type(ref).__name__} when traced; build it with '
            'mfs_amd.one_dim.moments.sde_cond_moments_* (mfs_amd has no CPU path for arbitrary callables)')
    if not ok or ref.which not in ('raw', 'central', 'scaled'):
        raise sym.NotDeviceDescribable(f'the {mode} transition-moment callable does not forward its mean / scale '
                                       'arguments the way the filter passes them')
    return ref.tables


def _check_mean_fn(fn: Callable, tables: TransitionTables, want_var: bool):
    ref = fn(sym.X)
    if isinstance(ref, TransitionRef) and ref.which in ('mean', 'mean_var'):
        if ref.tables is not tables:
            a, _ = ref.tables.table()
            b, _ = tables.table()
            if a.shape != b.shape or not np.array_equal(a, b) or ref.tables.umap != tables.umap:
                raise sym.NotDeviceDescribable('state_cond_mean belongs to a different transition model than the '
                                               'conditional moments')
        return
    # a hand-written polynomial mean (e.g. `lambda x: F * x`): must agree with the transition tables
    if want_var:
        m, v = ref
        m, v = sym.as_poly(m, 'state_cond_mean_var mean'), sym.as_poly(v, 'state_cond_mean_var variance')
    else:
        m, v = sym.as_poly(ref, 'state_cond_mean'), None
    xs = np.linspace(-1.3, 1.7, 7)
    if not np.allclose(m(xs), tables.cond_mean(xs), rtol=1e-13, atol=1e-15):
        raise sym.NotDeviceDescribable('state_cond_mean disagrees with the mean implied by the transition moments')
    if v is not None and not np.allclose(v(xs), tables.cond_var(xs), rtol=1e-13, atol=1e-15):
        raise sym.NotDeviceDescribable('state_cond_mean_var disagrees with the variance of the transition model')


def _trace_likelihood(fn: Callable) -> sym.LikelihoodSpec:
    spec = fn(sym.Y, sym.X)
    if not isinstance(spec, sym.LikelihoodSpec):
        raise sym.NotDeviceDescribable(
            f'measurement_cond_pdf returned {type(spec).__name__} when traced; use mfs_amd.stats.bernoulli_pmf / '
            'poisson_pmf / norm_pdf with mfs_amd.sym.exp / log (mfs_amd has no CPU path for arbitrary callables)')
    return spec


def trace_model(mode, transition_fn, mean_fn, pdf_fn):
    tables = _trace_transition(transition_fn, mode)
    if mean_fn is not None:
        _check_mean_fn(mean_fn, tables, want_var=(mode == 'scaled'))
    return tables, _trace_likelihood(pdf_fn)


def build_model_struct(tables: TransitionTables, lik: sym.LikelihoodSpec, B: int):
    """Fill `struct mfs_model_1d`; returns (struct, keep-alive arrays)."""
    coef, J = tables.table(B)
    coef_batched = int(coef.ndim == 3)
    lp = np.ascontiguousarray(lik.params, dtype=np.float64)
    if lp.ndim == 1:
        lik_batched = 0
    elif lp.shape[:-1] == (B,):
        lik_batched = 1
    else:
        raise ValueError(f'likelihood parameters are batched with shape {lp.shape[:-1]}, but the filter batch is {B}')
    m = _lib.MfsModel1d()
    m.trans_kind = _lib.TRANS[tables.kind]
    m.umap = _lib.UMAP[tables.umap]
    m.n_terms = tables.n_terms
    m.degree = J
    m.n_rows = coef.shape[-2]
    m.coef_batched = coef_batched
    m.lik_kind = _lib.LIK[lik.kind]
    m.n_lik = lp.shape[-1]
    m.lik_batched = lik_batched
    m.mean_x_coef = tables.mean_x_coef
    m.coef = coef.ctypes.data_as(_lib.c_double_p)
    m.lik = lp.ctypes.data_as(_lib.c_double_p)
    return m, (coef, lp)


# ---------------------------------------------------------------------------------------------------------------------
# running
# ---------------------------------------------------------------------------------------------------------------------
def _prep_inputs(ms0, mean0, scale0, ys, tables, lik):
    ys = np.asarray(ys)
    ys = ys.astype(np.float64) if ys.dtype != np.float64 else ys
    squeeze = ys.ndim == 1
    ys2 = np.ascontiguousarray(ys[None, :] if squeeze else ys)
    if ys2.ndim != 2:
        raise ValueError(f'ys must have shape (T,) or (B, T), got {ys.shape}')
    B, T = ys2.shape
    ms0 = np.ascontiguousarray(ms0, dtype=np.float64)
    if ms0.ndim not in (1, 2) or (ms0.ndim == 2 and ms0.shape[0] != B):
        raise ValueError(f'initial moments must have shape (2N,) or ({B}, 2N), got {ms0.shape}')
    num_moments = ms0.shape[-1]
    if num_moments % 2 != 0:
        # the reference warns and proceeds (mfs/one_dim/filtering.py:65-66) with N = floor(M / 2)
        # (mfs/one_dim/quadtures.py:122): the last entry is carried along, computed by the N-node rule, never read
        warnings.warn(f'The order of moments {num_moments - 1} is not odd.')
    batched = ms0.ndim == 2
    nb = B if batched else 1

    def aux(v):
        if v is None:
            return None
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 0:
            return np.full((nb,), float(v))
        if v.shape != (nb,):
            raise ValueError(f'mean0 / scale0 must be scalar or shape ({nb},) to match the initial moments')
        return np.ascontiguousarray(v)

    if squeeze and (tables.batch_shape() != () or lik.params.ndim != 1):
        raise ValueError('per-replicate model parameters need ys of shape (B, T)')
    return ys2, squeeze, B, T, ms0, num_moments // 2, batched, aux(mean0), aux(scale0)


def _run(mode, tables, lik, ms0, mean0, scale0, ys, stable, device=0, want_first_nan=False):
    ys2, squeeze, B, T, ms0, N, batched, mean0, scale0 = _prep_inputs(ms0, mean0, scale0, ys, tables, lik)
    if not 2 <= N <= _lib.MAX_N:
        raise ValueError(f'N = {N} outside the supported range [2, {_lib.MAX_N}]')
    model, keep = build_model_struct(tables, lik, B)
    odd_tail = ms0.shape[-1] - 2 * N           # 1 for an odd moment count
    # results land in page-locked memory from the library's pool (PCIe-rate copies; MFS_PINNED_OUTPUTS=0 disables)
    out_m = _lib.pinned_empty((B, T, 2 * N + odd_tail), device=device)
    out_mean = _lib.pinned_empty((B, T), device=device) if mode != 'raw' else None
    out_scale = _lib.pinned_empty((B, T), device=device) if mode == 'scaled' else None
    out_nell = np.empty((B,))
    out_fn = np.empty((B,), dtype=np.int32)
    L = _lib.lib()
    _lib.check(L.mfs_filter_1d(C.byref(model), _lib.MODE[mode] | (_lib.MODE_ODD_TAIL if odd_tail else 0), N, T, B,
                               _lib.ptr(ms0), int(batched),
                               _lib.ptr(mean0), _lib.ptr(scale0), _lib.ptr(ys2), int(bool(stable)),
                               _lib.ptr(out_m), _lib.ptr(out_mean), _lib.ptr(out_scale), _lib.ptr(out_nell),
                               _lib.ptr(out_fn), device, None))
    del keep
    if squeeze:
        out_m, out_nell, out_fn = out_m[0], out_nell[0], out_fn[0]
        out_mean = None if out_mean is None else out_mean[0]
        out_scale = None if out_scale is None else out_scale[0]
    return out_m, out_mean, out_scale, out_nell, out_fn


def moment_filter_rms(state_cond_raw_moments: Callable, measurement_cond_pdf: Callable, rms0, ys,
                      stable: bool = False, *, device: int = 0, return_first_nan: bool = False):
    """Moment filter with raw moments (mfs/one_dim/filtering.py:32-89): returns (rmss (T, 2N), nell).

    With `ys` of shape (B, T) the returns are (B, T, 2N) and (B,).  `return_first_nan=True` appends the per-replicate
    index of the first NaN-poisoned step (-1 if none).
    """
    tables, lik = trace_model('raw', state_cond_raw_moments, None, measurement_cond_pdf)
    m, _, _, nell, fn = _run('raw', tables, lik, rms0, None, None, ys, stable, device)
    return (m, nell, fn) if return_first_nan else (m, nell)


def moment_filter_cms(state_cond_central_moments: Callable, state_cond_mean: Callable, measurement_cond_pdf: Callable,
                      cms0, mean0, ys, stable: bool = False, *, device: int = 0, return_first_nan: bool = False):
    """Moment filter with central moments (mfs/one_dim/filtering.py:92-161): returns (cmss, means, nell)."""
    tables, lik = trace_model('central', state_cond_central_moments, state_cond_mean, measurement_cond_pdf)
    m, means, _, nell, fn = _run('central', tables, lik, cms0, mean0, None, ys, stable, device)
    return (m, means, nell, fn) if return_first_nan else (m, means, nell)


def moment_filter_scms(state_cond_scaled_central_moments: Callable, state_cond_mean_var: Callable,
                       measurement_cond_pdf: Callable, scms0, mean0, scale0, ys, stable: bool = False, *,
                       device: int = 0, return_first_nan: bool = False):
    """Moment filter with scaled central moments (mfs/one_dim/filtering.py:164-240): (scmss, means, scales, nell)."""
    tables, lik = trace_model('scaled', state_cond_scaled_central_moments, state_cond_mean_var, measurement_cond_pdf)
    m, means, scales, nell, fn = _run('scaled', tables, lik, scms0, mean0, scale0, ys, stable, device)
    return (m, means, scales, nell, fn) if return_first_nan else (m, means, scales, nell)
