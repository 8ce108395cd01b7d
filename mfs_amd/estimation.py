"""Maximum-likelihood parameter estimation on top of the moment filter.

The reference minimises `obj_func(params, ys) -> nell` with jaxopt's L-BFGS-B, differentiating through the scan by JAX
autodiff (dardel/parameter_estimation/mf.py:37-54, 70-73).  Two gradients are offered here:

* `nell_and_grad_forward` / `minimise_nell_forward` -- the derivative propagated INSIDE the kernel's time loop in forward
  mode (mfs_amd/csrc/filter1d_grad.hpp, C entry `mfs_filter_1d_grad`): the filter's state is carried as dual numbers, one
  launch returns the NLL and its exact gradient.  The model is given the way the reference's objective builds it -- a
  function of the parameter vector returning the transition closures and the measurement model -- and is traced once per
  evaluation at the parameter point and a five-point stencil around it, which yields the value tables and their
  parameter derivatives (exact for tables polynomial in the parameters up to degree four, which covers TME orders <= 2 of
  drifts linear in theta; ~1e-10 otherwise).  The recursion itself -- Cholesky pivots, eigenvalues, weights, quadrature
  sums over T steps -- is differentiated exactly.
* `nell_and_grad` / `minimise_nell` -- a central finite difference of the plain filter evaluated in the SAME launch as the
  objective, 2P + 1 replicates per optimiser step: any model the filters accept, N up to 32.
"""
import ctypes as C
from typing import Callable, Sequence

import numpy as np
import scipy.optimize

__all__ = ['nell_and_grad_forward', 'minimise_nell_forward', 'nell_and_grad', 'minimise_nell']


# ---------------------------------------------------------------------------------------------------------------------
# forward mode inside the kernel
# ---------------------------------------------------------------------------------------------------------------------
_STENCIL = np.array([-2., -1., 1., 2.])
_WEIGHTS = np.array([1., -8., 8., -1.]) / 12.


def _trace(model, flat, mode):
    from mfs_amd.one_dim import filtering
    closures = model(flat)
    if len(closures) == 3:
        trans, mean_fn, pdf = closures
    else:
        trans, pdf = closures
        mean_fn = None
    tables, lik = filtering.trace_model(mode, trans, mean_fn, pdf)
    nb = flat.shape[0]
    coef, J = tables.table(nb)
    if coef.ndim == 2:
        coef = np.broadcast_to(coef, (nb,) + coef.shape)
    lp = np.asarray(lik.params)
    if lp.ndim == 1:
        lp = np.broadcast_to(lp, (nb,) + lp.shape)
    return tables, lik, coef, lp, J


def _tables_and_tangents(model: Callable, params: np.ndarray, mode: str, rel_step: float, tangents: str = 'complex-step'):
    """Trace `model` at params and obtain d tables / d theta.  params (P,) or (R, P).  Returns the centre tables / likelihood
    and d coef (R?, P, rows, J + 1), d lik (R?, P, n_lik).

    tangents = 'complex-step' (default): the model is traced ONCE more with complex parameters theta + i h e_p (h = 1e-30 |theta|);
        every table entry is an analytic function of theta built from the model's own arithmetic, so Im(entry) / h is its exact
        derivative -- the derivative of the traced polynomials themselves, with no truncation or cancellation error whatever the
        dependence on theta (exp, softplus, products, ...).  The model must not apply non-analytic operations to its
        parameters (abs, comparisons, casts to float): if the complex trace raises, or disagrees with the stencil below by
        more than 1e-5, the stencil is used and a warning says so.
    tangents = 'stencil': five-point central differences on the table builder (`rel_step`), exact for tables polynomial in
        theta up to degree four."""
    from mfs_amd import sym
    params = np.asarray(params, dtype=np.float64)
    batched = params.ndim == 2
    pts0 = params if batched else params[None, :]
    R, P = pts0.shape

    def stencil():
        h = rel_step * np.maximum(np.abs(pts0), 1.0)                     # (R, P)
        pts = np.broadcast_to(pts0, (1 + 4 * P, R, P)).copy()
        for j in range(P):
            for s_, off in enumerate(_STENCIL):
                pts[1 + 4 * j + s_, :, j] += off * h[:, j]
        tables, lik, coef, lp, J = _trace(model, pts.reshape(-1, P), mode)
        coef = coef.reshape((1 + 4 * P, R) + coef.shape[1:])
        lp = lp.reshape((1 + 4 * P, R) + lp.shape[1:])
        dcoef = np.empty((R, P) + coef.shape[2:])
        dlik = np.empty((R, P) + lp.shape[2:])
        for j in range(P):
            sl = slice(1 + 4 * j, 5 + 4 * j)
            dcoef[:, j] = np.tensordot(_WEIGHTS, coef[sl], axes=(0, 0)) / h[:, j][:, None, None]
            dlik[:, j] = np.tensordot(_WEIGHTS, lp[sl], axes=(0, 0)) / h[:, j][:, None]
        return tables, lik, coef[0], lp[0], dcoef, dlik, J

    if tangents == 'stencil':
        tables, lik, coef0, lp0, dcoef, dlik, J = stencil()
        return tables, lik, coef0, lp0, dcoef, dlik, batched, J
    if tangents != 'complex-step':
        raise ValueError("tangents must be 'complex-step' or 'stencil'")
    try:
        h = 1e-30 * np.maximum(np.abs(pts0), 1.0)
        pts = np.broadcast_to(pts0.astype(np.complex128), (1 + P, R, P)).copy()
        for j in range(P):
            pts[1 + j, :, j] += 1j * h[:, j]
        with sym.coefficient_dtype(np.complex128):
            tables_c, lik_c, coef, lp, J = _trace(model, pts.reshape(-1, P), mode)
        coef = coef.reshape((1 + P, R) + coef.shape[1:])
        lp = lp.reshape((1 + P, R) + lp.shape[1:])
        if np.abs(coef[0].imag).max(initial=0.) != 0. or np.abs(lp[0].imag).max(initial=0.) != 0.:
            raise TypeError('the unperturbed trace is not real')
        dcoef = np.stack([coef[1 + j].imag / h[:, j][:, None, None] for j in range(P)], axis=1)
        dlik = np.stack([lp[1 + j].imag / h[:, j][:, None] for j in range(P)], axis=1)
        # the value tables and the model description come from a plain float64 trace (what the filters themselves use)
        tables, lik, coef0, lp0, J0 = _trace(model, pts0, mode)
        if J0 != J or coef0.shape != coef[0].shape or not np.allclose(coef0, coef[0].real, rtol=1e-13, atol=1e-300):
            raise TypeError('complex and real traces disagree')
        return tables, lik, coef0, lp0, dcoef, dlik, batched, J
    except (TypeError, ValueError, ArithmeticError) as e:
        import warnings
        warnings.warn(f'complex-step trace of the model failed ({e!r}); using the five-point stencil for the parameter '
                      'tangents of the tables (exact only for tables polynomial in the parameters up to degree four)')
        tables, lik, coef0, lp0, dcoef, dlik, J = stencil()
        return tables, lik, coef0, lp0, dcoef, dlik, batched, J


def nell_and_grad_forward(model: Callable, params, ms0, mean0, ys, scale0=None, mode: str = 'central',
                          rel_step: float = 1e-3, device: int = 0, return_first_nan: bool = False,
                          tangents: str = 'complex-step'):
    """NLL and d NLL / d params of the moment filter, forward mode in the kernel.

    model(params) -> (state_cond_moments, state_cond_mean[_var], measurement_cond_pdf) for `mode` in
                     {'central', 'scaled'}, (state_cond_raw_moments, measurement_cond_pdf) for 'raw'; it is called with
                     an array of shape (K, P) of parameter vectors and must close over them the way
                     dardel/parameter_estimation/mf.py:41-53 does (`drift(x, params[:, 0])`, ...)
    params           (P,): one parameter point for every trajectory in `ys`;  (R, P): one point per row of ys (R, T)
    ys               (T,) or (B, T)
    Returns (nell (B,) or scalar, grad (B, P) or (P,)).  N <= 16, P <= 4.
    """
    from mfs_amd import _lib
    from mfs_amd.one_dim import filtering
    tables, lik, coef, lp, dcoef, dlik, batched, J = _tables_and_tangents(model, params, mode, rel_step, tangents)
    P = dcoef.shape[1]
    ys = np.asarray(ys, dtype=np.float64)
    squeeze = ys.ndim == 1 and not batched
    ys2 = np.ascontiguousarray(ys[None, :] if ys.ndim == 1 else ys)
    if batched and ys2.shape[0] == 1:
        ys2 = np.ascontiguousarray(np.broadcast_to(ys2, (dcoef.shape[0], ys2.shape[1])))
    B, T = ys2.shape
    if batched and dcoef.shape[0] != B:
        raise ValueError(f'{dcoef.shape[0]} parameter points for {B} measurement rows')
    ms0 = np.ascontiguousarray(ms0, dtype=np.float64)
    N = ms0.shape[-1] // 2
    m = _lib.MfsModel1d()
    m.trans_kind = _lib.TRANS[tables.kind]
    m.umap = _lib.UMAP[tables.umap]
    m.n_terms = tables.n_terms
    m.degree = J
    m.n_rows = coef.shape[-2]
    m.coef_batched = m.lik_batched = int(batched)
    m.lik_kind = _lib.LIK[lik.kind]
    m.n_lik = lp.shape[-1]
    m.mean_x_coef = tables.mean_x_coef
    coef_c = np.ascontiguousarray(coef if batched else coef[0])
    lp_c = np.ascontiguousarray(lp if batched else lp[0])
    dcoef_c = np.ascontiguousarray(dcoef if batched else dcoef[0])
    dlik_c = np.ascontiguousarray(dlik if batched else dlik[0])
    m.coef = coef_c.ctypes.data_as(_lib.c_double_p)
    m.lik = lp_c.ctypes.data_as(_lib.c_double_p)
    mean_a = None if mean0 is None else np.ascontiguousarray(np.atleast_1d(np.asarray(mean0, dtype=np.float64)))
    scale_a = None if scale0 is None else np.ascontiguousarray(np.atleast_1d(np.asarray(scale0, dtype=np.float64)))
    out_nell, out_grad, out_fn = np.empty((B,)), np.empty((B, P)), np.empty((B,), dtype=np.int32)
    _lib.check(_lib.lib().mfs_filter_1d_grad(C.byref(m), _lib.ptr(dcoef_c), _lib.ptr(dlik_c), P, _lib.MODE[mode], N, T, B,
                                             _lib.ptr(ms0), int(ms0.ndim == 2), _lib.ptr(mean_a), _lib.ptr(scale_a),
                                             _lib.ptr(ys2), _lib.ptr(out_nell), _lib.ptr(out_grad), _lib.ptr(out_fn), device,
                                             None))
    if squeeze:
        out_nell, out_grad, out_fn = out_nell[0], out_grad[0], out_fn[0]
    return (out_nell, out_grad, out_fn) if return_first_nan else (out_nell, out_grad)


def minimise_nell_forward(model: Callable, init_params: Sequence[float], ms0, mean0, ys, scale0=None, mode='central',
                          bounds=None, device: int = 0, **options):
    """L-BFGS-B on the summed NLL of the trajectories in `ys` with the in-kernel forward-mode gradient: one launch per
    objective evaluation (the role of jaxopt.ScipyMinimize in dardel/parameter_estimation/mf.py:70-73).  The result carries
    `launches` and `nonfinite_evaluations`."""
    state = {'launches': 0, 'nonfinite': 0, 'last_grad': None}

    def fun(p):
        f, g = nell_and_grad_forward(model, p, ms0, mean0, ys, scale0, mode, device=device)
        state['launches'] += 1
        f, g = float(np.sum(f)), np.atleast_2d(g).sum(axis=0)
        if not (np.isfinite(f) and np.all(np.isfinite(g))):
            # a NaN-poisoned filter: a wall, with the last finite gradient so that the line search backs off downhill
            state['nonfinite'] += 1
            return 1e300, (state['last_grad'] if state['last_grad'] is not None else np.zeros_like(p))
        state['last_grad'] = g
        return f, g

    res = scipy.optimize.minimize(fun, np.asarray(init_params, dtype=np.float64), jac=True, method='L-BFGS-B',
                                  bounds=bounds, options=options or None)
    res.launches, res.nonfinite_evaluations = state['launches'], state['nonfinite']
    return res


# ---------------------------------------------------------------------------------------------------------------------
# finite differences in one launch (any model, any N)
# ---------------------------------------------------------------------------------------------------------------------
def nell_and_grad(nell_batch: Callable[[np.ndarray], np.ndarray], params: np.ndarray, rel_step: float = 1e-5,
                  return_info: bool = False):
    """Objective and central-difference gradient from ONE batched evaluation.

    `nell_batch(P)` maps an array (R, P) of parameter vectors to the (R,) negative log-likelihoods (one filter per
    row, all rows in one launch).  A probe that NaN-poisons is replaced by a one-sided difference from the centre; if both
    probes of a parameter poison, that component is NaN (never silently zero).  Returns (nell, grad (P,)[, info])."""
    params = np.asarray(params, dtype=np.float64)
    P = params.shape[0]
    h = rel_step * np.maximum(np.abs(params), 1.0)
    pts = np.tile(params, (2 * P + 1, 1))
    for i in range(P):
        pts[1 + 2 * i, i] += h[i]
        pts[2 + 2 * i, i] -= h[i]
    vals = np.asarray(nell_batch(pts), dtype=np.float64)
    up, dn, f0 = vals[1::2], vals[2::2], vals[0]
    grad = (up - dn) / (2 * h)
    one_sided = []
    for i in range(P):
        if np.isfinite(grad[i]):
            continue
        if np.isfinite(up[i]) and np.isfinite(f0):
            grad[i] = (up[i] - f0) / h[i]
            one_sided.append(i)
        elif np.isfinite(dn[i]) and np.isfinite(f0):
            grad[i] = (f0 - dn[i]) / h[i]
            one_sided.append(i)
    info = {'one_sided': one_sided, 'nonfinite_probes': int((~np.isfinite(vals[1:])).sum())}
    return (float(f0), grad, info) if return_info else (float(f0), grad)


def minimise_nell(nell_batch: Callable[[np.ndarray], np.ndarray], init_params: Sequence[float],
                  bounds=None, rel_step: float = 1e-5, **options):
    """L-BFGS-B on a batched NLL with in-launch finite-difference gradients.  A non-finite objective (NaN-poisoned filter)
    is a wall returned with the last finite gradient, so that the line search backs off instead of seeing a stationary
    point; the result carries `launches`, `nonfinite_evaluations` and `one_sided_gradients`."""
    state = {'launches': 0, 'nonfinite': 0, 'one_sided': 0, 'last_grad': None}

    def fun(p):
        f, g, info = nell_and_grad(nell_batch, p, rel_step, return_info=True)
        state['launches'] += 1
        state['one_sided'] += len(info['one_sided'])
        if not (np.isfinite(f) and np.all(np.isfinite(g))):
            state['nonfinite'] += 1
            return 1e300, (state['last_grad'] if state['last_grad'] is not None else np.zeros_like(p))
        state['last_grad'] = g
        return f, g

    res = scipy.optimize.minimize(fun, np.asarray(init_params, dtype=np.float64), jac=True, method='L-BFGS-B',
                                  bounds=bounds, options=options or None)
    res.launches, res.nonfinite_evaluations, res.one_sided_gradients = state['launches'], state['nonfinite'], state['one_sided']
    return res
