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
    if d == 1 and b.ndim < 2:      # a scalar dispersion is a 1 x 1 matrix (the reference's 1-D models return a float)
        b = b.reshape(1, 1)
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

    is_gaussian = False

    def __init__(self, d, kappas, Q, var, label):
        self.d, self.kappas, self.Q, self.var, self.label = d, np.asarray(kappas, dtype=np.int32), Q, var, label

    def var_blocks(self, D):
        """(d, D, D, ...) coefficient blocks of the conditional variances (diagonal of tme.mean_and_cov)."""
        out = np.zeros((self.d,) + (D,) * self.d)
        for k, v in enumerate(self.var):
            c = v.coef.reshape(v.coef.shape if v.coef.ndim == self.d else (1,) * self.d)
            out[(k,) + tuple(slice(0, n) for n in c.shape)] = c
        return out

    def as_one_dim(self):
        """The same family as a 1-D `mfs_amd.tme_poly.TransitionTables` (d = 1 only): the N-D filter with d = 1 is the
        1-D filter (reference tests/test_filtering.py:304-329) and runs on the 1-D kernels."""
        if self.d != 1:
            raise ValueError('only a d = 1 family has a 1-D form')
        from mfs_amd.sym import Poly
        from mfs_amd.tme_poly import TransitionTables
        K = int(self.kappas.max())
        Q = [Poly(np.zeros(1), 'x') for _ in range(K)]
        for kap, q in zip(self.kappas, self.Q):
            Q[int(kap[0]) - 1] = Poly(np.atleast_1d(q.coef), 'x')
        return TransitionTables('operator', 'x', Q, 1., Q[0], Poly(np.atleast_1d(self.var[0].coef), 'x'), self.label)

    def dense_table(self):
        """(n_terms, D, D, ...) coefficient block with a common per-variable degree bound D - 1."""
        D = max(max(np.atleast_1d(p.coef).shape) for p in list(self.Q) + list(self.var))
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


# ---------------------------------------------------------------------------------------------------------------------
# Normal closures (mfs/multi_dims/moments.py:257-337 Euler--Maruyama, :340-411 TME mean / covariance)
# ---------------------------------------------------------------------------------------------------------------------
class GaussianTablesND:
    """X' | x ~ N(mu(x), Sigma(x)) with polynomial mean vector and covariance matrix (d = 2 on the device).

    The reference evaluates E[prod (X'_k - c_k)^{n_k}] with Kan's formula per multi-index
    (mfs/multi_dims/moments.py:110-154); for d = 2 the same numbers follow from Stein's lemma as a two-term recursion,
        M(a, b) = m_0 M(a-1, b) + (a-1) S_00 M(a-2, b) + b S_01 M(a-1, b-1),     M(0, b) = m_1 M(0, b-1) + (b-1) S_11 M(0, b-2),
    with m = mu(x) - c, which is what the kernel runs per node.
    """

    is_gaussian = True

    def __init__(self, d, mean, cov, label):
        self.d, self.mean, self.cov, self.label = d, mean, cov, label

    def as_one_dim(self):
        if self.d != 1:
            raise ValueError('only a d = 1 family has a 1-D form')
        from mfs_amd.sym import Poly
        from mfs_amd.tme_poly import normal_tables
        return normal_tables(Poly(np.atleast_1d(self.mean[0].coef), 'x'), Poly(np.atleast_1d(self.cov[0][0].coef), 'x'),
                             self.label)

    def dense_table(self):
        """(5, D, D): mu_0, mu_1, S_00, S_01, S_11."""
        if self.d != 2:
            raise NotDeviceDescribable('the device N-D path supports d = 2')
        polys = [self.mean[0], self.mean[1], self.cov[0][0], self.cov[0][1], self.cov[1][1]]
        D = max(max(p.coef.shape) for p in polys)
        out = np.zeros((5, D, D))
        for t, p in enumerate(polys):
            out[(t,) + tuple(slice(0, n) for n in p.coef.shape)] = p.coef
        return np.ascontiguousarray(out), D

    def cond_mean(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.stack([m(x) for m in self.mean], axis=-1)

    def cond_var(self, x):
        x = np.asarray(x, dtype=np.float64)
        return np.stack([self.cov[i][i](x) for i in range(self.d)], axis=-1)

    def cond_moments(self, x, multi_indices, mean=None, scale=None):
        if self.d != 2:
            raise NotImplementedError
        x = np.asarray(x, dtype=np.float64)
        mi = np.asarray(multi_indices, dtype=int)
        c = np.zeros(2) if mean is None else np.broadcast_to(np.asarray(mean, dtype=np.float64), (2,))
        m0, m1 = self.mean[0](x) - c[0], self.mean[1](x) - c[1]
        s00, s01, s11 = self.cov[0][0](x), self.cov[0][1](x), self.cov[1][1](x)
        P = int(mi.max()) + 1
        M = {}
        for b in range(P):
            M[0, b] = np.ones_like(m0) if b == 0 else m1 * M[0, b - 1] + ((b - 1) * s11 * M[0, b - 2] if b >= 2 else 0.)
        for a in range(1, P):
            for b in range(P):
                v = m0 * M[a - 1, b]
                if a >= 2:
                    v = v + (a - 1) * s00 * M[a - 2, b]
                if b >= 1:
                    v = v + b * s01 * M[a - 1, b - 1]
                M[a, b] = v
        out = np.stack([M[int(n0), int(n1)] for n0, n1 in mi], axis=-1)
        if scale is not None:
            out = out / np.prod(np.asarray(scale, dtype=np.float64) ** mi, axis=-1)
        return out


def normal_tables_nd(drift, dispersion, d: int, dt: float, order) -> GaussianTablesND:
    """`order` = 'euler' (mean x + a dt, covariance b b^T dt) or an int TME order (tme.mean_and_cov, truncated in dt):
        cov_ij = sum_{r=1}^{M} dt^r/r! [ (1 + delta_ij) q_{r, e_i + e_j} - sum_{s=1}^{r-1} C(r, s) q_{s, e_i} q_{r-s, e_j} ]."""
    a, g = trace_sde_nd(drift, dispersion, d)
    xs = [PolyND.variable(d, k) for k in range(d)]
    zero = PolyND(np.float64(0.), d)
    if order == 'euler':
        mean = [(xs[k] + dt * a[k]).trimmed() for k in range(d)]
        cov = [[(dt * g[i][j]).trimmed() for j in range(d)] for i in range(d)]
        return GaussianTablesND(d, mean, cov, 'euler')
    order = int(order)
    q = generator_power_tables_nd(a, g, order)

    def e(i):
        return tuple(1 if m == i else 0 for m in range(d))

    def e2(i, j):
        return tuple((1 if m == i else 0) + (1 if m == j else 0) for m in range(d))

    mean = []
    for k in range(d):
        acc = xs[k]
        for r in range(1, order + 1):
            acc = acc + (dt ** r / math.factorial(r)) * q[r].get(e(k), zero)
        mean.append(acc.trimmed())
    cov = [[None] * d for _ in range(d)]
    for i in range(d):
        for j in range(d):
            acc = zero
            for r in range(1, order + 1):
                term = (2. if i == j else 1.) * q[r].get(e2(i, j), zero)
                for s in range(1, r):
                    term = term - math.comb(r, s) * (q[s].get(e(i), zero) * q[r - s].get(e(j), zero))
                acc = acc + (dt ** r / math.factorial(r)) * term
            cov[i][j] = acc.trimmed()
    return GaussianTablesND(d, mean, cov, f'tme_normal_{order}')


class BatchedTablesND:
    """B transition families of one kind stacked along a leading replicate axis (per-replicate drift / dispersion
    parameters: the theta-grid of BASELINE config 4 carried over to the N-D filters).  Built by
    `mfs_amd.multi_dims.moments.batch_closures`."""

    def __init__(self, members):
        self.members = list(members)
        first = self.members[0]
        if any(m.is_gaussian != first.is_gaussian or m.d != first.d for m in self.members):
            raise NotDeviceDescribable('batched transition families must be of one kind and dimension')
        self.d, self.is_gaussian, self.label = first.d, first.is_gaussian, first.label + f' x{len(self.members)}'
        if not self.is_gaussian:
            kaps = sorted({tuple(int(v) for v in k) for m in self.members for k in m.kappas}, key=lambda k: (sum(k), k))
            self.kappas = np.asarray(kaps, dtype=np.int32)

    def dense_table(self):
        tabs = [m.dense_table() for m in self.members]
        D = max(t[1] for t in tabs)
        if self.is_gaussian:
            out = np.zeros((len(tabs), 5) + (D,) * self.d)
            for b, (t, _) in enumerate(tabs):
                out[(b, slice(None)) + tuple(slice(0, n) for n in t.shape[1:])] = t
            return out, D
        out = np.zeros((len(tabs), len(self.kappas)) + (D,) * self.d)
        index = {tuple(int(v) for v in k): i for i, k in enumerate(self.kappas)}
        for b, (m, (t, _)) in enumerate(zip(self.members, tabs)):
            for r, kap in enumerate(m.kappas):
                out[(b, index[tuple(int(v) for v in kap)]) + tuple(slice(0, n) for n in t.shape[1:])] = t[r]
        return out, D

    def var_blocks(self, D):
        return np.stack([m.var_blocks(D) for m in self.members])
