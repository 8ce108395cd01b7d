"""N-D Taylor moment expansion in operator form over multivariate polynomial rings (host side, once per model).

For an SDE in R^d with polynomial drift a(x) and dispersion b(x) (diffusion matrix g = b b^T), the generator
A = sum_i a_i d_i + 1/2 sum_ij g_ij d_i d_j satisfies

    sum_{r<=M} dt^r/r! A^r  =  sum_{|kappa| <= 2M} Q_kappa(x; dt) d^kappa,

with Q_kappa independent of the test function, so for phi(u) = prod_k (u_k - c_k)^{n_k}

    E[phi(X') | x] ~= sum_kappa Q_kappa(x) prod_k n_k!/(n_k - kappa_k)! (x_k - c_k)^(n_k - kappa_k).

This replaces the z x r nested-autodiff evaluations of `tme.expectation` per step in
mfs/multi_dims/moments.py:414-479.  Product rule used by the recursion:

    a_i d_i (q d^kappa)        = a_i (d_i q) d^kappa + a_i q d^(kappa + e_i)
    g_ij d_i d_j (q d^kappa)   = g_ij [ (d_i d_j q) d^kappa + (d_i q) d^(kappa+e_j) + (d_j q) d^(kappa+e_i)
                                        + q d^(kappa + e_i + e_j) ]
"""
import itertools
import math
from typing import Dict, List, Tuple

import numpy as np

from mfs_amd.sym import NotDeviceDescribable


class PolyND:
    """Dense polynomial in d variables: coef[a_0, ..., a_{d-1}] multiplies prod_k x_k^{a_k}."""
    __array_priority__ = 1000

    def __init__(self, coef, d=None):
        coef = np.asarray(coef, dtype=np.float64)
        if coef.ndim == 0:
            if d is None:
                raise ValueError('dimension needed for a constant')
            coef = coef.reshape((1,) * d)
        self.coef = coef
        self.d = coef.ndim

    @classmethod
    def variable(cls, d, k):
        shape = [1] * d
        shape[k] = 2
        c = np.zeros(shape)
        idx = [0] * d
        idx[k] = 1
        c[tuple(idx)] = 1.
        return cls(c)

    def _lift(self, o):
        if isinstance(o, PolyND):
            return o
        if isinstance(o, (int, float, np.floating, np.integer)):
            return PolyND(np.float64(o), self.d)
        raise NotDeviceDescribable(f'{type(o).__name__} cannot enter a polynomial drift / dispersion expression')

    @staticmethod
    def _pad(c, shape):
        out = np.zeros(shape)
        out[tuple(slice(0, n) for n in c.shape)] = c
        return out

    def __add__(self, o):
        o = self._lift(o)
        shape = tuple(max(a, b) for a, b in zip(self.coef.shape, o.coef.shape))
        return PolyND(self._pad(self.coef, shape) + self._pad(o.coef, shape))

    __radd__ = __add__

    def __neg__(self):
        return PolyND(-self.coef)

    def __sub__(self, o):
        return self + (-self._lift(o))

    def __rsub__(self, o):
        return self._lift(o) + (-self)

    def __mul__(self, o):
        o = self._lift(o)
        shape = tuple(a + b - 1 for a, b in zip(self.coef.shape, o.coef.shape))
        out = np.zeros(shape)
        for idx in itertools.product(*[range(n) for n in self.coef.shape]):
            v = self.coef[idx]
            if v != 0.:
                out[tuple(slice(i, i + n) for i, n in zip(idx, o.coef.shape))] += v * o.coef
        return PolyND(out)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, (int, float, np.floating, np.integer)):
            return PolyND(self.coef / float(o))
        raise NotDeviceDescribable('division by a non-constant expression')

    def __pow__(self, k):
        if not isinstance(k, (int, np.integer)) or k < 0:
            raise NotDeviceDescribable('only non-negative integer powers are polynomial')
        out = PolyND(np.float64(1.), self.d)
        for _ in range(int(k)):
            out = out * self
        return out

    def diff(self, k):
        c = self.coef
        if c.shape[k] == 1:
            return PolyND(np.float64(0.), self.d)
        sl = [slice(None)] * self.d
        sl[k] = slice(1, None)
        shape = [1] * self.d
        shape[k] = c.shape[k] - 1
        return PolyND(c[tuple(sl)] * np.arange(1, c.shape[k], dtype=np.float64).reshape(shape))

    def is_zero(self):
        return not np.any(self.coef)

    def trimmed(self):
        c = self.coef
        scale = np.max(np.abs(c)) if c.size else 0.
        for k in range(self.d):
            while c.shape[k] > 1 and np.all(np.abs(np.take(c, c.shape[k] - 1, axis=k)) <= 1e-15 * scale):
                c = np.take(c, range(c.shape[k] - 1), axis=k)
        return PolyND(c)

    def __call__(self, x):
        """x (..., d) -> (...)."""
        x = np.asarray(x, dtype=np.float64)
        out = np.zeros(x.shape[:-1])
        for idx in itertools.product(*[range(n) for n in self.coef.shape]):
            v = self.coef[idx]
            if v != 0.:
                out = out + v * np.prod(x ** np.asarray(idx, dtype=np.float64), axis=-1)
        return out


def trace_sde_nd(drift, dispersion, d: int) -> Tuple[List[PolyND], List[List[PolyND]]]:
    """Call the model's drift / dispersion with an object array of variables; returns (a_i, g_ij = (b b^T)_ij)."""
    xs = np.empty((d,), dtype=object)
    for k in range(d):
        xs[k] = PolyND.variable(d, k)
    a = np.asarray(drift(xs), dtype=object).reshape(-1)
    if a.shape[0] != d:
        raise NotDeviceDescribable(f'drift returned {a.shape[0]} components for a {d}-dimensional state')
    zero = PolyND(np.float64(0.), d)
    a = [zero._lift(v) for v in a]
    b = np.asarray(dispersion(xs), dtype=object)
    if b.ndim != 2 or b.shape[0] != d:
        raise NotDeviceDescribable('dispersion must return a (d, w) matrix')
    b = [[zero._lift(v) for v in row] for row in b]
    g = [[sum((b[i][m] * b[j][m] for m in range(len(b[i]))), zero) for j in range(d)] for i in range(d)]
    return a, g


def generator_power_tables_nd(a: List[PolyND], g: List[List[PolyND]], order: int) -> List[Dict[tuple, PolyND]]:
    d = len(a)
    one = PolyND(np.float64(1.), d)
    q = [{(0,) * d: one}]
    for _ in range(order):
        nxt: Dict[tuple, PolyND] = {}

        def acc(kappa, p):
            if p.is_zero():
                return
            nxt[kappa] = nxt[kappa] + p if kappa in nxt else p

        def bump(kappa, *dims):
            k = list(kappa)
            for i in dims:
                k[i] += 1
            return tuple(k)

        for kappa, p in q[-1].items():
            dp = [p.diff(i) for i in range(d)]
            for i in range(d):
                acc(kappa, a[i] * dp[i])
                acc(bump(kappa, i), a[i] * p)
                for j in range(d):
                    if g[i][j].is_zero():
                        continue
                    h = 0.5 * g[i][j]
                    acc(kappa, h * dp[i].diff(j))
                    acc(bump(kappa, j), h * dp[i])
                    acc(bump(kappa, i), h * dp[j])
                    acc(bump(kappa, i, j), h * p)
        q.append(nxt)
    return q


class TransitionTablesND:
    """Device description of an N-D TME transition family (operator form).

    kappas : (n_terms, d) derivative multi-indices (kappa = 0 excluded: Q_0 = 1)
    Q      : list of PolyND, one per kappa
    Conditional mean_k = x_k + Q_{e_k}(x).
    """

    def __init__(self, d, kappas, Q, var, label):
        self.d, self.kappas, self.Q, self.var, self.label = d, np.asarray(kappas, dtype=np.int32), Q, var, label

    def dense_table(self):
        """(n_terms, D, D, ...) coefficient block with a common per-variable degree bound D - 1."""
        D = max(max(p.coef.shape) for p in self.Q)
        out = np.zeros((len(self.Q),) + (D,) * self.d)
        for t, p in enumerate(self.Q):
            out[(t,) + tuple(slice(0, n) for n in p.coef.shape)] = p.coef
        return np.ascontiguousarray(out), D

    # numeric evaluation (inspection / host tests)
    def cond_mean(self, x):
        x = np.asarray(x, dtype=np.float64)
        out = x.copy()
        for t, kap in enumerate(self.kappas):
            if kap.sum() == 1:
                out[..., int(np.argmax(kap))] += self.Q[t](x)
        return out

    def cond_var(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.stack([v(x) for v in self.var], axis=-1)

    def cond_moments(self, x, multi_indices, mean=None, scale=None):
        x = np.asarray(x, dtype=np.float64)
        mi = np.asarray(multi_indices, dtype=int)
        d = self.d
        mean = np.zeros(d) if mean is None else np.broadcast_to(np.asarray(mean, dtype=np.float64), (d,))
        dx = x - mean
        Qv = [q(x) for q in self.Q]
        out = np.zeros(x.shape[:-1] + (mi.shape[0],))
        for zi, n in enumerate(mi):
            val = np.prod(dx ** n, axis=-1)
            for t, kap in enumerate(self.kappas):
                if np.all(kap <= n):
                    ff = math.prod(math.perm(int(nk), int(kk)) for nk, kk in zip(n, kap))
                    val = val + Qv[t] * ff * np.prod(dx ** (n - kap), axis=-1)
            out[..., zi] = val
        if scale is not None:
            out = out / np.prod(np.asarray(scale, dtype=np.float64) ** mi, axis=-1)
        return out


def tme_tables_nd(drift, dispersion, d: int, dt: float, order: int) -> TransitionTablesND:
    a, g = trace_sde_nd(drift, dispersion, d)
    q = generator_power_tables_nd(a, g, order)
    zero = PolyND(np.float64(0.), d)
    total: Dict[tuple, PolyND] = {}
    for r in range(1, order + 1):
        for kappa, p in q[r].items():
            term = (dt ** r / math.factorial(r)) * p
            total[kappa] = total[kappa] + term if kappa in total else term
    total.pop((0,) * d, None)
    kappas = sorted(total.keys(), key=lambda k: (sum(k), k))
    Q = [total[k].trimmed() for k in kappas]

    # tme.mean_and_cov diagonal (used by the scaled mode): truncated in dt like the 1-D case
    def e(i):
        return tuple(1 if m == i else 0 for m in range(d))

    var = []
    for i in range(d):
        acc = zero
        ei, eii = e(i), tuple(2 if m == i else 0 for m in range(d))
        for r in range(1, order + 1):
            term = 2. * q[r].get(eii, zero)
            for s in range(1, r):
                term = term - math.comb(r, s) * (q[s].get(ei, zero) * q[r - s].get(ei, zero))
            acc = acc + (dt ** r / math.factorial(r)) * term
        var.append(acc.trimmed())
    return TransitionTablesND(d, kappas, Q, var, f'tme_{order}')
