"""Maximum-likelihood parameter estimation on top of the batched NLL.

The reference minimises `obj_func(params, ys) -> nell` with jaxopt's L-BFGS-B, differentiating through the scan by JAX
autodiff (dardel/parameter_estimation/mf.py:37-54, 70-73).  The HIP path has no reverse mode; it has something JAX on
CPU does not: thousands of filters per launch for the price of one.  The gradient is therefore a central finite
difference evaluated in the SAME launch as the objective -- 2P + 1 replicates of the filter per optimiser step (P = number
of parameters) -- and SciPy's L-BFGS-B drives the search.  Several starting points / data sets can be optimised at once
by stacking them on the replicate axis.
"""
from typing import Callable, Sequence

import numpy as np
import scipy.optimize


def nell_and_grad(nell_batch: Callable[[np.ndarray], np.ndarray], params: np.ndarray, rel_step: float = 1e-5):
    """Objective and central-difference gradient from ONE batched evaluation.

    `nell_batch(P)` maps an array (R, P) of parameter vectors to the (R,) negative log-likelihoods (one filter per
    row, all rows in one launch).  Returns (nell, grad (P,))."""
    params = np.asarray(params, dtype=np.float64)
    P = params.shape[0]
    h = rel_step * np.maximum(np.abs(params), 1.0)
    pts = np.tile(params, (2 * P + 1, 1))
    for i in range(P):
        pts[1 + 2 * i, i] += h[i]
        pts[2 + 2 * i, i] -= h[i]
    vals = np.asarray(nell_batch(pts), dtype=np.float64)
    grad = (vals[1::2] - vals[2::2]) / (2 * h)
    return float(vals[0]), grad


def minimise_nell(nell_batch: Callable[[np.ndarray], np.ndarray], init_params: Sequence[float],
                  bounds=None, rel_step: float = 1e-5, **options):
    """L-BFGS-B on a batched NLL with in-launch finite-difference gradients (the role of jaxopt.ScipyMinimize in
    dardel/parameter_estimation/mf.py:70-73).  Non-finite objective values (NaN-poisoned filters) are treated as +inf
    walls with a zero gradient so that the line search backs off."""

    def fun(p):
        f, g = nell_and_grad(nell_batch, p, rel_step)
        if not np.isfinite(f):
            return 1e300, np.zeros_like(p)
        g = np.where(np.isfinite(g), g, 0.)
        return f, g

    return scipy.optimize.minimize(fun, np.asarray(init_params, dtype=np.float64), jac=True, method='L-BFGS-B',
                                   bounds=bounds, options=options or None)
