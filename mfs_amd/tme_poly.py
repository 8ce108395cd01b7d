"""Taylor moment expansion in operator form, by polynomial-ring recursion (host side, once per model).

The reference obtains E[phi(X_{t+dt}) | x] ~= sum_{r<=M} dt^r/r! A^r phi(x) from the third-party `tme` package by
nesting JAX autodiff N x 2N times per step (mfs/one_dim/moments.py:141-179).  For a 1-D SDE the generator
A = a(x) D + 1/2 b(x)^2 D^2 satisfies

    sum_{r<=M} dt^r/r! A^r  =  sum_{k<=2M} Q_k(x; dt) D^k,

where the Q_k depend on the model only, NOT on the test function.  With phi(u) = (u - c)^n this gives

    E[(X' - c)^n | x] ~= sum_{k<=2M} Q_k(x) n!/(n-k)! (x - c)^(n-k),

which is what the HIP kernel evaluates per quadrature node (MFS_TRANS_OPERATOR, include/mfs_hip.h).  The Q_k are
built here by the recursion (product rule on q(x) D^k)

    A (q D^k) = (a q' + 1/2 g q'') D^k + (a q + g q') D^(k+1) + 1/2 g q D^(k+2),       g = b^2,

inside the ring of polynomials in u with u = x or u = tanh(x) (closed under d/dx because tanh' = 1 - tanh^2).

`tme.mean_and_cov` (mfs/one_dim/moments.py:175,190) truncates the covariance in powers of dt:
    cov = sum_{r=1}^{M} dt^r/r! [ A^r(x^2) - sum_{s=0}^{r} C(r,s) A^s x A^(r-s) x ]
        = sum_{r=1}^{M} dt^r/r! [ 2 q_{r,2} - sum_{s=1}^{r-1} C(r,s) q_{s,1} q_{r-s,1} ]
(the s = 0 and s = r terms cancel 2 x q_{r,1} of A^r(x^2) = 2 x q_{r,1} + 2 q_{r,2}).
"""
import math
from typing import Dict, List

import numpy as np

from mfs_amd.sym import Poly, NotDeviceDescribable, _zeros as sym_zeros

MAX_TERMS = 8    # MFS_MAX_TERMS
MAX_DEGREE = 15  # MFS_MAX_DEGREE


def _zero():
    return Poly(np.zeros(1))


def generator_power_tables(a: Poly, g: Poly, order: int) -> List[Dict[int, Poly]]:
    """q[r][k] with A^r = sum_k q[r][k] D^k, r = 0..order."""
    umap = Poly._merge_umap(a, g)
    if umap is None:
        umap = 'x'
    a = Poly(a.coef, umap)
    g = Poly(g.coef, umap)
    q = [{0: Poly(np.ones(1), umap)}]
    for _ in range(order):
        nxt: Dict[int, Poly] = {}

        def acc(k, p):
            nxt[k] = nxt[k] + p if k in nxt else p

        for k, p in q[-1].items():
            p = Poly(p.coef, umap)
            dp = p.dx()
            ddp = dp.dx()
            acc(k, a * dp + 0.5 * (g * ddp))
            acc(k + 1, a * p + g * dp)
            acc(k + 2, 0.5 * (g * p))
        q.append({k: Poly(v.coef, umap) for k, v in nxt.items()})
    return q


def _clean(p: Poly) -> Poly:
    """Drop trailing coefficients that are rounding dust relative to the largest one."""
    c = p.coef
    scale = np.max(np.abs(c)) if c.size else 0.
    return p.trimmed(tol=1e-15 * scale)


class TransitionTables:
    """Device description of a 1-D transition-moment family (fills `mfs_model_1d`).

    kind 'operator': Q[0..K-1] = Q_1..Q_K, cond mean = x + Q_1(u), `var` = tme.mean_and_cov variance.
    kind 'gaussian': normal closure with mu(x) = mean_x_coef x + mean_poly(u), var(x) = var(u).
    """

    def __init__(self, kind, umap, Q, mean_x_coef, mean_poly, var, label):
        self.kind = kind
        self.umap = umap or 'x'
        self.Q = Q
        self.mean_x_coef = float(mean_x_coef)
        self.mean_poly = mean_poly
        self.var = var
        self.label = label

    @property
    def n_terms(self):
        return len(self.Q) if self.kind == 'operator' else 0

    def rows(self):
        return (list(self.Q) + [self.var]) if self.kind == 'operator' else [self.mean_poly, self.var]

    def batch_shape(self):
        return np.broadcast_shapes(*[r.coef.shape[:-1] for r in self.rows()])

    def table(self, B=None):
        """(n_rows, J+1) or (B, n_rows, J+1) float64, and J."""
        rows = self.rows()
        J = max(r.degree for r in rows)
        if J > MAX_DEGREE:
            raise NotDeviceDescribable(f'coefficient polynomial degree {J} exceeds MFS_MAX_DEGREE = {MAX_DEGREE}')
        bs = self.batch_shape()
        out = sym_zeros(bs + (len(rows), J + 1))
        for i, r in enumerate(rows):
            out[..., i, :r.degree + 1] = np.broadcast_to(r.coef, bs + (r.degree + 1,))
        if bs not in ((), (B,)) and B is not None:
            raise ValueError(f'model parameters are batched with shape {bs}, but the filter batch is {B}')
        return np.ascontiguousarray(out), J

    # -- numeric evaluation of the closures (for inspection / host-side tests; the filters never call these)
    def cond_mean(self, x):
        x = np.asarray(x, dtype=np.float64)
        return self.mean_x_coef * x + self.mean_poly(x)

    def cond_var(self, x):
        return self.var(np.asarray(x, dtype=np.float64))

    def cond_moments(self, x, orders, mean=0., scale=1.):
        """E[((X' - mean) / scale)^n | x] for n in orders; shape x.shape + (len(orders),)."""
        x = np.asarray(x, dtype=np.float64)
        orders = np.atleast_1d(np.asarray(orders, dtype=int))
        nmax = int(orders.max()) if orders.size else 0
        if self.kind == 'operator':
            Qv = [np.ones_like(x)] + [q(x) for q in self.Q]
            dx = x - mean
            out = []
            for n in range(nmax + 1):
                val = np.zeros_like(x)
                for k in range(min(n, len(self.Q)) + 1):
                    ff = math.perm(n, k)
                    val = val + Qv[k] * ff * dx ** (n - k)
                out.append(val)
        else:
            m = self.cond_mean(x) - mean
            v = self.cond_var(x)
            out = [np.ones_like(x), m]
            for n in range(2, nmax + 1):
                out.append(m * out[-1] + (n - 1) * v * out[-2])
        allp = np.stack(out[:nmax + 1], axis=-1)
        return allp[..., orders] / np.asarray(scale, dtype=np.float64) ** orders


def tme_tables(a: Poly, b: Poly, dt: float, order: int, gaussian: bool) -> TransitionTables:
    """TME-`order` tables for drift a(u) and dispersion b(u): operator form, or its normal closure."""
    if order < 1:
        raise ValueError('tme_order must be >= 1')
    if 2 * order > MAX_TERMS and not gaussian:
        raise NotDeviceDescribable(f'tme_order {order} needs {2 * order} operator terms > MFS_MAX_TERMS = {MAX_TERMS}')
    g = b * b
    q = generator_power_tables(a, g, order)
    umap = q[0][0].umap
    K = 2 * order
    Q = []
    for k in range(1, K + 1):
        acc = _zero()
        for r in range(1, order + 1):
            if k in q[r]:
                acc = acc + (dt ** r / math.factorial(r)) * q[r][k]
        Q.append(_clean(Poly(acc.coef, umap)))
    var = _zero()
    for r in range(1, order + 1):
        term = 2. * q[r].get(2, _zero())
        for s in range(1, r):
            term = term - math.comb(r, s) * (q[s].get(1, _zero()) * q[r - s].get(1, _zero()))
        var = var + (dt ** r / math.factorial(r)) * term
    var = _clean(Poly(var.coef, umap))
    if not gaussian:
        return TransitionTables('operator', umap, Q, 1., Q[0], var, f'tme_{order}')
    if umap == 'tanh':
        return TransitionTables('gaussian', umap, [], 1., Q[0], var, f'tme_normal_{order}')
    x = Poly(np.array([0., 1.]), 'x')
    return TransitionTables('gaussian', 'x', [], 0., _clean(x + Q[0]), var, f'tme_normal_{order}')


def euler_tables(a: Poly, b: Poly, dt: float) -> TransitionTables:
    """Euler--Maruyama normal closure: mu = x + a dt, var = b^2 dt (mfs/one_dim/moments.py:222-255)."""
    umap = Poly._merge_umap(a, b) or 'x'
    var = Poly(((b * b) * dt).coef, umap)
    if umap == 'tanh':
        return TransitionTables('gaussian', umap, [], 1., Poly((a * dt).coef, umap), var, 'euler')
    x = Poly(np.array([0., 1.]), 'x')
    return TransitionTables('gaussian', 'x', [], 0., Poly((x + a * dt).coef, 'x'), var, 'euler')


def normal_tables(mean: Poly, var: Poly, label='normal') -> TransitionTables:
    """A user-specified normal transition X' | x ~ N(mean(x), var(x)) with polynomial mean / variance in x or tanh x
    (e.g. the exact OU discretisation of dardel/convergence/convergence_mf.py:86-107)."""
    umap = Poly._merge_umap(mean, var) or 'x'
    return TransitionTables('gaussian', umap, [], 0., Poly(mean.coef, umap), Poly(var.coef, umap), label)
