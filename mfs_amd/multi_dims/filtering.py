"""The N-D moment filters on MI355X, mirroring `mfs.multi_dims.filtering`.

Same names, positional order and return tuples as the reference (mfs/multi_dims/filtering.py:283-288, 210-217,
33-41); both closure signatures ('multi-index' for sde_cond_moments_tme, 'index' for the Normal closures, :245-249).
d = 2 runs on `filternd_kernel` (mfs_amd/csrc/filternd_kernel.hpp); d = 1 is routed to the 1-D kernels (the reference
guarantees the d = 1 N-D path equals the 1-D path, tests/test_filtering.py:304-329).  Extensions over the reference:
`ys` may carry a leading replicate axis -- (B, T) for scalar measurements, (B, T, ny) for vector ones -- initial
moments may be (z,) shared or (B, z), model parameters may be per-replicate.  No CPU fallback.

The measurement likelihood must be a product of factors, each a function of ONE state component and one measurement
column: `bernoulli.pmf(y, logistic(x[0]))` (mfs/multi_dims/ss_models.py:63-67) or
`math.prod(norm.pdf(y, x, sd))` on vector y, x (reference tests/test_filtering.py:44-46).
"""
import ctypes as C
from typing import Callable, Tuple

import numpy as np

from mfs_amd import _lib, sym
from mfs_amd.multi_dims.moments import TransitionRefND

__all__ = ['moment_filter_nd_rms', 'moment_filter_nd_cms', 'moment_filter_nd_scms']


def _trace_transition(fn_and_flag, mode, moments_partial_order=None):
    fn, signature = fn_and_flag
    if signature not in ('multi-index', 'index'):
        raise sym.NotDeviceDescribable(f"unknown transition-moment signature {signature!r}")
    second = sym.ORDER
    if mode == 'raw':
        ref = fn(sym.X, second)
        ok = isinstance(ref, TransitionRefND) and ref.mean is None
    elif mode == 'scaled':
        ref = fn(sym.X, second, sym.MEAN, sym.SCALE)
        ok = isinstance(ref, TransitionRefND) and ref.mean is sym.MEAN and ref.scale is sym.SCALE
    else:
        ref = fn(sym.X, second, sym.MEAN)
        ok = isinstance(ref, TransitionRefND) and ref.mean is sym.MEAN
    if not isinstance(ref, TransitionRefND):
        raise sym.NotDeviceDescribable('the transition-moment callable is not device-describable; build it with '
                                       'mfs_amd.multi_dims.moments.sde_cond_moments_tme / _tme_normal / '
                                       '_euler_maruyama')
    if not ok:
        raise sym.NotDeviceDescribable('the transition-moment callable does not forward its mean / scale arguments')
    gaussian = ref.tables.is_gaussian
    if gaussian != (signature == 'index'):
        raise sym.NotDeviceDescribable(f"signature {signature!r} does not match the closure: the Normal closures take "
                                       "'index', sde_cond_moments_tme takes 'multi-index' "
                                       '(mfs/multi_dims/filtering.py:245-249)')
    if gaussian and moments_partial_order is not None:
        # an 'index' closure looks moments up in the table it was built for (mfs/multi_dims/moments.py:293-300)
        built_for = np.asarray(fn.multi_indices)
        if built_for.shape != np.asarray(moments_partial_order[0]).shape or \
                np.any(built_for != np.asarray(moments_partial_order[0])):
            raise ValueError("the 'index' closure was built for a different multi-index table than the filter's")
    return ref.tables


def _trace_likelihood(fn, d):
    """Call the measurement model with the measurement placeholder and the d tagged state components; returns the list
    of likelihood factors (each with .component and .ycol)."""
    xs = sym.state_vector(d)
    spec = fn(sym.Y, xs if d > 1 else xs[0])
    if isinstance(spec, sym.LikelihoodVector):
        if len(spec) == 1:
            spec = spec.factors[0]
        else:
            raise sym.NotDeviceDescribable('measurement_cond_pdf returned one value per component; reduce it with '
                                           'math.prod(...) as the reference does (tests/test_filtering.py:44-46)')
    if not isinstance(spec, (sym.LikelihoodSpec, sym.LikelihoodProduct)):
        raise sym.NotDeviceDescribable(
            f'measurement_cond_pdf returned {type(spec).__name__} when traced; use mfs_amd.stats.bernoulli_pmf / '
            'poisson_pmf / norm_pdf with mfs_amd.sym.exp / log (mfs_amd has no CPU path for arbitrary callables)')
    factors = spec.factors
    if not 1 <= len(factors) <= _lib.ND_MAX_FACTORS:
        raise sym.NotDeviceDescribable(f'{len(factors)} likelihood factors; the device takes 1..{_lib.ND_MAX_FACTORS}')
    return factors


def _model_struct(tables, factors, B=1):
    if tables.d != 2:
        raise sym.NotDeviceDescribable('the device N-D kernel is for d = 2')
    if isinstance(factors, sym.LikelihoodSpec):
        factors = [factors]
    dense, D = tables.dense_table()
    batched_coef = dense.ndim == 4          # (B, rows, D, D): per-replicate drift / dispersion parameters
    if batched_coef and dense.shape[0] != B:
        raise ValueError(f'transition tables are batched over {dense.shape[0]} replicates, the filter batch is {B}')
    lead = (B,) if batched_coef else ()
    if tables.is_gaussian:
        if D > _lib.ND_MAX_EXTENT:
            raise sym.NotDeviceDescribable(f'coefficient extent {D} exceeds MFS_ND_MAX_EXTENT = {_lib.ND_MAX_EXTENT}')
        coef = np.zeros(lead + (_lib.ND_ROWS, D, D))
        coef[..., :5, :, :] = dense        # mu_0, mu_1, S_00, S_01, S_11
        kind, last = _lib.ND_TRANS_GAUSSIAN, 5
    else:
        kind, rows_of = _lib.ND_TRANS_OPERATOR, []
        for kap in tables.kappas:
            kap = tuple(int(v) for v in kap)
            if kap not in _lib.ND_KAPPAS:
                raise sym.NotDeviceDescribable(f'derivative term {kap} needs |kappa| <= 6, i.e. tme_order <= 3 on the '
                                               'device')
            rows_of.append(_lib.ND_KAPPAS.index(kap))
        last = max(rows_of) + 1 if rows_of else 0
        n_rows = _lib.nd_table_rows(last)                 # 16 rows (|kappa| <= 4) or 29 (TME order 3)
        max_extent = _lib.ND_MAX_EXTENT_HI if n_rows == _lib.ND_ROWS_MAX else _lib.ND_MAX_EXTENT
        if D > max_extent:
            raise sym.NotDeviceDescribable(f'coefficient extent {D} exceeds the device limit {max_extent}')
        coef = np.zeros(lead + (n_rows, D, D))
        for t, row in enumerate(rows_of):
            coef[..., row, :, :] = dense[..., t, :, :]
        coef[..., n_rows - 2:, :, :] = tables.var_blocks(D)   # diagonal of tme.mean_and_cov (scaled mode)
    # likelihood factors: [n_factors][MAX_LIK], or [B][n_factors][MAX_LIK] when a factor has per-replicate parameters
    nf = len(factors)
    lik_batched = any(np.asarray(f.params).ndim > 1 for f in factors)
    lp = np.zeros(((B,) if lik_batched else ()) + (nf, _lib.MAX_LIK))
    for i, f in enumerate(factors):
        prm = np.asarray(f.params, dtype=np.float64)
        if prm.ndim > 2 or (prm.ndim == 2 and prm.shape[0] != B):
            raise ValueError(f'likelihood parameters are batched with shape {prm.shape[:-1]}, the filter batch is {B}')
        lp[..., i, :prm.shape[-1]] = prm
    m = _lib.MfsModelNd()
    m.d, m.n_terms, m.extent = 2, last, D
    m.trans_kind = kind
    m.n_factors = nf
    m.ny = max(f.ycol for f in factors) + 1
    for i, f in enumerate(factors):
        m.fac_kind[i], m.fac_component[i], m.fac_ycol[i] = _lib.LIK[f.kind], int(f.component), int(f.ycol)
        m.fac_n_par[i] = int(np.asarray(f.params).shape[-1])
    m.coef_batched, m.lik_batched = int(batched_coef), int(lik_batched)
    coef, lp = np.ascontiguousarray(coef), np.ascontiguousarray(lp)
    m.coef = coef.ctypes.data_as(_lib.c_double_p)
    m.lik = lp.ctypes.data_as(_lib.c_double_p)
    return m, (coef, lp)


def _split_ys(ys, ny):
    """-> (ys (B, T, ny) contiguous float64, squeeze): (T,) / (B, T) for scalar measurements, (T, ny) / (B, T, ny) for
    vector ones (the reference's ys_2d is (T, 2))."""
    ys = np.asarray(ys, dtype=np.float64)
    if ny == 1:
        if ys.ndim == 3 and ys.shape[-1] == 1:
            ys = ys[..., 0]
        if ys.ndim not in (1, 2):
            raise ValueError(f'ys must have shape (T,) or (B, T) for scalar measurements, got {ys.shape}')
        squeeze = ys.ndim == 1
        ys3 = (ys[None, :] if squeeze else ys)[..., None]
    else:
        if ys.ndim not in (2, 3) or ys.shape[-1] < ny:
            raise ValueError(f'ys must have shape (T, {ny}) or (B, T, {ny}) for this likelihood, got {ys.shape}')
        squeeze = ys.ndim == 2
        ys3 = (ys[None] if squeeze else ys)[..., :ny]
    return np.ascontiguousarray(ys3), squeeze


def _run_1d(mode, tables, factors, ys, ms0, mean0, scale0, stable, device):
    """d = 1: the N-D filter IS the 1-D filter (reference tests/test_filtering.py:304-329) -- run the 1-D kernels."""
    from mfs_amd.one_dim import filtering as f1
    t1 = tables.as_one_dim()
    if len(factors) != 1:
        raise sym.NotDeviceDescribable('a d = 1 filter takes a single likelihood factor')
    ys = np.asarray(ys, dtype=np.float64)
    if ys.ndim >= 2 and ys.shape[-1] == 1:
        ys = ys[..., 0]
    m0 = None if mean0 is None else np.asarray(mean0, dtype=np.float64)[..., 0]
    s0 = None if scale0 is None else np.asarray(scale0, dtype=np.float64)[..., 0]
    m, means, scales, nell, fn = f1._run(mode, t1, factors[0], ms0, m0, s0, ys, stable, device)
    means = None if means is None else means[..., None]
    scales = None if scales is None else scales[..., None]
    if mode == 'scaled':
        return m, means, scales, nell, fn
    return m, means, nell, fn


def _run_nd(mode, tables, factors, ys, moments_partial_order, ms0, mean0, stable, device, scale0=None):
    multi_indices, inds = moments_partial_order
    multi_indices = np.asarray(multi_indices)
    ms0 = np.ascontiguousarray(ms0, dtype=np.float64)
    if multi_indices.shape[0] != ms0.shape[-1]:  # the reference's only raise (mfs/multi_dims/filtering.py:238-239)
        raise ValueError(f'The size of multi_indices {multi_indices.shape[0]} must match that of cms0 {ms0.shape[-1]}.')
    d = multi_indices.shape[-1]
    if d != tables.d:
        raise ValueError(f'the transition closure is {tables.d}-dimensional, the multi-index table {d}-dimensional')
    if d == 1:
        return _run_1d(mode, tables, factors, ys, ms0, mean0, scale0, stable, device)
    inds = np.asarray(inds)
    s = inds.shape[1]
    N = next((n for n in range(2, 8) if n * (n + 1) // 2 == s), None)
    if d != 2 or N is None:
        raise sym.NotDeviceDescribable(f'the device N-D path supports d <= 2 with 2 <= N <= 7 (got d = {d}, s = {s})')
    ny = max(f.ycol for f in factors) + 1
    ys3, squeeze = _split_ys(ys, ny)
    B, T = ys3.shape[:2]
    batched = ms0.ndim == 2
    if batched and ms0.shape[0] != B:
        raise ValueError(f'initial moments batch {ms0.shape[0]} does not match ys batch {B}')
    z = multi_indices.shape[0]
    mean_a = scale_a = None
    if mode != 'raw':
        mean_a = np.ascontiguousarray(np.broadcast_to(np.asarray(mean0, dtype=np.float64), ((B, 2) if batched else (2,))))
    if mode == 'scaled':
        scale_a = np.ascontiguousarray(np.broadcast_to(np.asarray(scale0, dtype=np.float64),
                                                       ((B, 2) if batched else (2,))))
    model, keep = _model_struct(tables, factors, B)
    if squeeze and (model.coef_batched or model.lik_batched):
        raise ValueError('per-replicate model parameters need ys with a leading replicate axis')
    mi32 = np.ascontiguousarray(multi_indices, dtype=np.int32)
    inds32 = np.ascontiguousarray(inds, dtype=np.int32)
    out_m = _lib.pinned_empty((B, T, z), device=device)
    out_mean = _lib.pinned_empty((B, T, 2), device=device) if mode != 'raw' else None
    out_scale = _lib.pinned_empty((B, T, 2), device=device) if mode == 'scaled' else None
    out_nell, out_fn = np.empty((B,)), np.empty((B,), dtype=np.int32)
    _lib.check(_lib.lib().mfs_filter_nd(C.byref(model), _lib.MODE[mode], N, T, B, z, _lib.ptr(mi32), _lib.ptr(inds32),
                                        _lib.ptr(ms0), int(batched), _lib.ptr(mean_a), _lib.ptr(scale_a), _lib.ptr(ys3),
                                        int(bool(stable)), _lib.ptr(out_m), _lib.ptr(out_mean), _lib.ptr(out_scale),
                                        _lib.ptr(out_nell), _lib.ptr(out_fn), device, None))
    del keep
    if squeeze:
        out_m, out_nell, out_fn = out_m[0], out_nell[0], out_fn[0]
        out_mean = None if out_mean is None else out_mean[0]
        out_scale = None if out_scale is None else out_scale[0]
    if mode == 'scaled':
        return out_m, out_mean, out_scale, out_nell, out_fn
    return out_m, out_mean, out_nell, out_fn


def moment_filter_nd_rms(state_cond_raw_moments: Tuple[Callable, str], measurement_cond_pdf: Callable, ys,
                         moments_partial_order, rms0, stable: bool = False, *, device: int = 0,
                         return_first_nan: bool = False):
    """Filtering with raw moments (mfs/multi_dims/filtering.py:283-344): returns (rmss (T, z), nell)."""
    tables = _trace_transition(state_cond_raw_moments, 'raw', moments_partial_order)
    lik = _trace_likelihood(measurement_cond_pdf, tables.d)
    m, _, nell, fn = _run_nd('raw', tables, lik, ys, moments_partial_order, rms0, None, stable, device)
    return (m, nell, fn) if return_first_nan else (m, nell)


def moment_filter_nd_cms(state_cond_central_moments: Tuple[Callable, str], state_cond_mean: Callable,
                         measurement_cond_pdf: Callable, ys, moments_partial_order, cms0, mean0,
                         stable: bool = False, *, device: int = 0, return_first_nan: bool = False):
    """Filtering with central moments (mfs/multi_dims/filtering.py:210-280): returns (cmss, means (T, d), nell)."""
    tables = _trace_transition(state_cond_central_moments, 'central', moments_partial_order)
    ref = state_cond_mean(sym.X)
    if not (isinstance(ref, TransitionRefND) and ref.tables is tables):
        raise sym.NotDeviceDescribable('state_cond_mean must come from the same sde_cond_moments_* call as the '
                                       'conditional central moments')
    lik = _trace_likelihood(measurement_cond_pdf, tables.d)
    m, means, nell, fn = _run_nd('central', tables, lik, ys, moments_partial_order, cms0, mean0, stable, device)
    return (m, means, nell, fn) if return_first_nan else (m, means, nell)


def moment_filter_nd_scms(state_cond_scms: Tuple[Callable, str], state_cond_mean_vars: Callable,
                          measurement_cond_pdf: Callable, ys, moments_partial_order, scms0, mean0, scale0,
                          stable: bool = False, *, device: int = 0, return_first_nan: bool = False):
    """Filtering with scaled central moments (mfs/multi_dims/filtering.py:33-207): returns
    (scmss (T, z), means (T, d), scales (T, d), nell)."""
    tables = _trace_transition(state_cond_scms, 'scaled', moments_partial_order)
    ref = state_cond_mean_vars(sym.X)
    if not (isinstance(ref, TransitionRefND) and ref.tables is tables and ref.which == 'mean_var'):
        raise sym.NotDeviceDescribable('state_cond_mean_vars must be the mean-and-variance closure of the same '
                                       'sde_cond_moments_* call as the conditional scaled moments')
    lik = _trace_likelihood(measurement_cond_pdf, tables.d)
    m, means, scales, nell, fn = _run_nd('scaled', tables, lik, ys, moments_partial_order, scms0, mean0, stable, device,
                                         scale0)
    return (m, means, scales, nell, fn) if return_first_nan else (m, means, scales, nell)
