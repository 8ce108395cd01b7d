"""The reference's 1-D moment filter in (effectively) exact arithmetic -- mpmath at 80+ digits (TEST INFRASTRUCTURE, see
oracle/__init__.py).

Why: at N = 15 the Hankel matrices of the Benes--Bernoulli filter reach cond ~ 1e12 ... 1e16, so two correct fp64
implementations of mfs/one_dim/filtering.py (XLA/LAPACK upstream, NumPy/LAPACK in oracle/one_dim.py, the C port, the HIP
kernel) differ by far more than 1e-16 and disagree on WHEN a replicate NaN-poisons.  None of them is "the truth"; the
truth is what the reference's ALGORITHM yields without rounding.  This module runs exactly that algorithm --

    mfs/one_dim/quadtures.py:122-133   G = ms[i+j], H = ms[i+j+1]; R = chol(G); K = R^-1 H R^-T; eigh(K);
                                       weights V[0,:]^2, nodes scale * lambda + mean
    mfs/one_dim/filtering.py:140-158   central-moment scan body (and :217-237, scaled)
    mfs/one_dim/moments.py:141-179     TME conditional moments, here in operator form with EXACT rational tables:
                                       sum_{r<=M} dt^r/r! A^r = sum_k Q_k(u) D^k, u = tanh x, for the Benes drift
    mfs/one_dim/ss_models.py:25-56     Benes--Bernoulli model (a = tanh x, b = 1, p = 1 / (1 + exp(-x^3 / 5)))

-- at a working precision where rounding is invisible at the 1e-30 level (checked: 80 and 120 digits agree), so that
every fp64 implementation can be scored by its distance from it.  dt is the fp64 value 1e-2 (what every implementation
is given), taken exactly.
"""
from fractions import Fraction
import math

import mpmath as mp


# ---------------------------------------------------------------------------------------------------------------------
# exact operator tables for a(x) = tanh x, b = 1: polynomials in u = tanh x with rational coefficients
# ---------------------------------------------------------------------------------------------------------------------
def _padd(p, q):
    n = max(len(p), len(q))
    return [(p[i] if i < len(p) else 0) + (q[i] if i < len(q) else 0) for i in range(n)]


def _pmul(p, q):
    out = [Fraction(0)] * (len(p) + len(q) - 1)
    for i, a in enumerate(p):
        for j, b in enumerate(q):
            out[i + j] += a * b
    return out


def _pscale(p, c):
    return [a * c for a in p]


def _pdx(p):
    """d/dx of a polynomial in u = tanh x: dp/du (1 - u^2)."""
    du = [Fraction(j) * p[j] for j in range(1, len(p))] or [Fraction(0)]
    return _pmul(du, [Fraction(1), Fraction(0), Fraction(-1)])


def benes_operator_tables(dt: float, order: int):
    """Q_1 .. Q_{2 order} and the tme.mean_and_cov variance as lists of Fractions (ascending powers of u)."""
    a = [Fraction(0), Fraction(1)]        # a(u) = u
    g = [Fraction(1)]                     # b^2 = 1
    q = [{0: [Fraction(1)]}]
    for _ in range(order):
        nxt = {}

        def acc(k, p):
            nxt[k] = _padd(nxt[k], p) if k in nxt else p
        for k, p in q[-1].items():
            dp = _pdx(p)
            ddp = _pdx(dp)
            acc(k, _padd(_pmul(a, dp), _pscale(_pmul(g, ddp), Fraction(1, 2))))
            acc(k + 1, _padd(_pmul(a, p), _pmul(g, dp)))
            acc(k + 2, _pscale(_pmul(g, p), Fraction(1, 2)))
        q.append(nxt)
    dtf = Fraction(dt)                    # the fp64 value, exactly
    K = 2 * order
    Q = []
    for k in range(1, K + 1):
        tot = [Fraction(0)]
        for r in range(1, order + 1):
            if k in q[r]:
                tot = _padd(tot, _pscale(q[r][k], dtf ** r / math.factorial(r)))
        Q.append(tot)
    var = [Fraction(0)]
    for r in range(1, order + 1):
        term = _pscale(q[r].get(2, [Fraction(0)]), Fraction(2))
        for s in range(1, r):
            term = _padd(term, _pscale(_pmul(q[s].get(1, [Fraction(0)]), q[r - s].get(1, [Fraction(0)])),
                                       Fraction(-math.comb(r, s))))
        var = _padd(var, _pscale(term, dtf ** r / math.factorial(r)))
    return Q, var


def _mpf(fr):
    return mp.mpf(fr.numerator) / mp.mpf(fr.denominator)


def _horner(coefs, u):
    acc = mp.mpf(0)
    for c in reversed(coefs):
        acc = acc * u + c
    return acc


# ---------------------------------------------------------------------------------------------------------------------
# quadrature and filter
# ---------------------------------------------------------------------------------------------------------------------
def _forward(R, B):
    """X with R X = B, R lower triangular, B a matrix."""
    n, m = B.rows, B.cols
    X = mp.matrix(n, m)
    for c in range(m):
        for i in range(n):
            X[i, c] = (B[i, c] - mp.fsum(R[i, k] * X[k, c] for k in range(i))) / R[i, i]
    return X


def moment_quadrature(ms, mean=0, scale=1):
    """Weights and nodes from 2n moments (mfs/one_dim/quadtures.py:122-133); None when G is not positive definite."""
    n = len(ms) // 2
    G = mp.matrix(n, n)
    H = mp.matrix(n, n)
    for i in range(n):
        for j in range(n):
            G[i, j] = ms[i + j]
            H[i, j] = ms[i + j + 1]
    # Cholesky by hand: a non-positive pivot is the reference's NaN-poisoning event
    R = mp.matrix(n, n)
    for j in range(n):
        s = G[j, j] - mp.fsum(R[j, k] ** 2 for k in range(j))
        if not s > 0:
            return None
        R[j, j] = mp.sqrt(s)
        for i in range(j + 1, n):
            R[i, j] = (G[i, j] - mp.fsum(R[i, k] * R[j, k] for k in range(j))) / R[j, j]
    X = _forward(R, H)                    # R X = H                  (:128, inner solve)
    Kt = _forward(R, X.T)                 # R K^T = X^T, K = X R^-T  (:129)
    K = (Kt + Kt.T) / 2                   # eigh symmetrises its input
    lam, V = mp.eigsy(K)
    w = [V[0, i] ** 2 for i in range(n)]
    x = [scale * lam[i] + mean for i in range(n)]
    return w, x


def benes_bernoulli_cms(cms0, mean0, ys, dt=1e-2, tme_order=3, slope=5., dps=80, scaled=False, scale0=None):
    """Central-moment (or scaled-central) Benes--Bernoulli filter in `dps`-digit arithmetic.

    Returns dict(moments (T, 2N) as lists of mpf, means, scales, nell, first_nan, nell_cum = the running NLL after
    each step).  After a poisoning event every
    later entry is None.
    """
    mp.mp.dps = dps
    Qf, varf = benes_operator_tables(dt, tme_order)
    Q = [[_mpf(c) for c in row] for row in Qf]
    var = [_mpf(c) for c in varf]
    M2 = len(cms0)
    ms = [mp.mpf(float(v)) for v in cms0]
    mean = mp.mpf(float(mean0))
    scale = mp.mpf(float(scale0)) if scaled else mp.mpf(1)
    slope = mp.mpf(float(slope))
    ff = [[mp.mpf(math.perm(n, k)) if k <= n else mp.mpf(0) for k in range(len(Q) + 1)] for n in range(M2)]
    out_m, out_mean, out_scale, out_nell = [], [], [], []
    nell = mp.mpf(0)
    first_nan = -1
    for t, y in enumerate(ys):
        if first_nan >= 0:
            out_m.append(None); out_mean.append(None); out_scale.append(None); out_nell.append(None)
            continue
        rule = moment_quadrature(ms, mean, scale)
        if rule is None:
            first_nan = t
            out_m.append(None); out_mean.append(None); out_scale.append(None); out_nell.append(None)
            continue
        w, x = rule
        u = [mp.tanh(xi) for xi in x]
        Qv = [[mp.mpf(1)] + [_horner(row, ui) for row in Q] for ui in u]
        new_mean = mp.fsum(wi * (xi + Qi[1]) for wi, xi, Qi in zip(w, x, Qv))   # E[X' | x] = x + Q_1
        new_scale = mp.sqrt(mp.fsum(wi * _horner(var, ui) for wi, ui in zip(w, u))) if scaled else mp.mpf(1)
        new = []
        for n in range(M2):
            tot = mp.mpf(0)
            for wi, xi, Qi in zip(w, x, Qv):
                dx = xi - new_mean
                tot += wi * mp.fsum(Qi[k] * ff[n][k] * dx ** (n - k) for k in range(min(n, len(Q)) + 1))
            new.append(tot / new_scale ** n)
        ms, mean, scale = new, new_mean, new_scale
        rule = moment_quadrature(ms, mean, scale)
        if rule is None:
            first_nan = t
            out_m.append(None); out_mean.append(None); out_scale.append(None); out_nell.append(None)
            continue
        w, x = rule
        p = [1 / (1 + mp.exp(-xi ** 3 / slope)) for xi in x]
        lik = [pi if y > 0.5 else 1 - pi for pi in p]
        py = mp.fsum(wi * li for wi, li in zip(w, lik))
        new_mean = mp.fsum(wi * xi * li for wi, xi, li in zip(w, x, lik)) / py
        if scaled:
            new_scale = mp.sqrt(mp.fsum(wi * (xi - new_mean) ** 2 * li for wi, xi, li in zip(w, x, lik)) / py)
        ms = [mp.fsum(wi * ((xi - new_mean) / new_scale) ** n * li for wi, xi, li in zip(w, x, lik)) / py
              for n in range(M2)]
        mean, scale = new_mean, new_scale
        nell -= mp.log(py)
        out_m.append(list(ms)); out_mean.append(mean); out_scale.append(scale); out_nell.append(nell)
    return dict(moments=out_m, means=out_mean, scales=out_scale, nell=nell if first_nan < 0 else None,
                first_nan=first_nan, nell_cum=out_nell)
