"""Pin the 1-D oracle with the reference's own analytic known-answer tests (restated).

reference tests/test_filtering.py:82-164, tests/test_one_dim_quadrature.py:48-113,
tests/test_one_dim_moments.py:19-52,90-118, tests/test_utils.py:198-209.
"""
import math

import numpy as np
import numpy.testing as npt
import pytest
import scipy.stats
import sympy as sp

from oracle import one_dim as o
from oracle import models, tme_sympy

# measurement set-up of reference tests/test_filtering.py:17-36 (NumPy RNG there too)
rs = np.random.RandomState(666)
dt, T = 1e-2, 100
ts = np.linspace(dt, dt * T, T)
ell, sigma = 1., 0.5
cov = np.exp(-np.abs(ts[None, :] - ts[:, None]) / ell) * sigma ** 2
R_noise = 1.
ys = np.linalg.cholesky(cov) @ rs.randn(T) + math.sqrt(R_noise) * rs.randn(T)


def drift(x):
    return -x / ell


def dispersion(_):
    return math.sqrt(2) * sigma / math.sqrt(ell)


def pdf(y, x):
    return models.norm_pdf(y, x, math.sqrt(R_noise))


def kf(F, Sigma, mean0, var0):
    mf, vf, nell = mean0, var0, 0.
    mfs, vfs = np.zeros(T), np.zeros(T)
    for k, y in enumerate(ys):
        mp, vp = F * mf, F * vf * F + Sigma
        s = vp + R_noise
        g = vp / s
        mf, vf = mp + g * (y - mp), vp - vp * g
        nell -= scipy.stats.norm.logpdf(y, mp, math.sqrt(s))
        mfs[k], vfs[k] = mf, vf
    return mfs, vfs, nell


def test_1d_convergence_to_kalman():
    """reference tests/test_filtering.py:82-111 (N = 10, TME-3 raw filter vs exact KF)."""
    F, Sigma = math.exp(-dt / ell), sigma ** 2 * (1 - math.exp(-2 / ell * dt))
    N = 10
    cond_rms, *_ = tme_sympy.sde_cond_moments_tme_1d(drift, dispersion, dt, 3, 2 * N)
    mean0, var0 = 0.1, 0.1
    rms0 = np.array([float(o.raw_moment_of_normal(mean0, var0, p)) for p in range(2 * N)])
    rmss, nell = o.moment_filter_rms(cond_rms, pdf, rms0, ys)
    mfs, vfs, true_nell = kf(F, Sigma, mean0, var0)
    npt.assert_allclose(rmss[:, 1], mfs, rtol=1e-2)
    npt.assert_allclose(rmss[:, 2] - rmss[:, 1] ** 2, vfs, rtol=1e-3)
    npt.assert_allclose(nell, true_nell, rtol=1e-5)


def test_routines_equivalence():
    """reference tests/test_filtering.py:113-164 (rms vs cms vs scms, N = 4, TME-2)."""
    N = 4
    mean0, var0 = 0., 0.5
    rms0 = np.array([float(o.raw_moment_of_normal(mean0, var0, p)) for p in range(2 * N)])
    cms0, scms0 = o.raw_to_central(rms0), o.raw_to_scaled(rms0)
    cond_rms, cond_cms, cond_scms, cond_mean, cond_mean_var = tme_sympy.sde_cond_moments_tme_1d(
        drift, dispersion, dt, 2, 2 * N)
    rmss, nell_r = o.moment_filter_rms(cond_rms, pdf, rms0, ys)
    cmss, means_c, nell_c = o.moment_filter_cms(cond_cms, cond_mean, pdf, cms0, mean0, ys)
    scmss, means, scales, nell_s = o.moment_filter_scms(cond_scms, cond_mean_var, pdf, scms0, mean0,
                                                         math.sqrt(var0), ys)
    npt.assert_array_almost_equal(cmss, np.array([o.raw_to_central(r) for r in rmss]), decimal=11)
    npt.assert_array_almost_equal(scmss, np.array([o.raw_to_scaled(r) for r in rmss]), decimal=10)
    npt.assert_array_almost_equal(means_c, means, decimal=15)
    npt.assert_array_almost_equal(rmss[:, 2] - rmss[:, 1] ** 2, scales ** 2, decimal=12)
    for nell in (nell_s, nell_c):
        npt.assert_array_almost_equal(nell_r, nell, decimal=11)


@pytest.mark.parametrize('N', [2, 5, 8])
def test_quadrature_gaussian_expectations(N):
    """reference tests/test_one_dim_quadrature.py:48-113: invariance across modes, polynomial exactness, E[exp]."""
    mean, var = 0.3, 0.7
    rms = np.array([float(o.raw_moment_of_normal(mean, var, p)) for p in range(2 * N)])
    cms = o.raw_to_central(rms)
    scms = o.raw_to_scaled(rms)
    w_r, x_r = o.moment_quadrature(rms)
    w_c, x_c = o.moment_quadrature(cms, mean)
    w_s, x_s = o.moment_quadrature(scms, mean, math.sqrt(var))
    for w, x in ((w_c, x_c), (w_s, x_s)):
        npt.assert_allclose(np.sort(x), np.sort(x_r), rtol=1e-8, atol=1e-10)
        npt.assert_allclose(w[np.argsort(x)], w_r[np.argsort(x_r)], rtol=1e-7, atol=1e-12)
    # exact for polynomials up to degree 2N - 1
    for p in range(2 * N):
        npt.assert_allclose(np.dot(w_r, x_r ** p), rms[p], rtol=1e-9, atol=1e-12)
    # Gauss--Hermite agreement
    gh_x, gh_w = np.polynomial.hermite_e.hermegauss(N)
    npt.assert_allclose(np.sort(x_r), mean + math.sqrt(var) * gh_x, rtol=1e-8, atol=1e-10)
    npt.assert_allclose(w_r[np.argsort(x_r)], gh_w / math.sqrt(2 * math.pi), rtol=1e-7, atol=1e-12)
    if N >= 5:
        npt.assert_allclose(np.dot(w_r, np.exp(0.5 * x_r)), math.exp(0.5 * mean + 0.125 * var), rtol=1e-5)


def test_raw_central_conversions():
    """reference tests/test_one_dim_moments.py:19-52: closed-form Gaussian moments and round trip."""
    mean, var = -0.4, 1.3
    s = 10
    rms = np.array([float(o.raw_moment_of_normal(mean, var, p)) for p in range(s)])
    cms = np.array([o.central_moment_of_normal(var, p) for p in range(s)])
    npt.assert_allclose(o.raw_to_central(rms), cms, rtol=1e-10, atol=1e-10)
    npt.assert_allclose(o.central_to_raw(cms, mean), rms, rtol=1e-12, atol=1e-12)
    npt.assert_allclose(o.scaled_to_central(o.raw_to_scaled(rms), math.sqrt(var)), cms, rtol=1e-10, atol=1e-10)
    npt.assert_allclose(rms[:4], [1., mean, mean ** 2 + var, mean ** 3 + 3 * mean * var], rtol=1e-13)


def test_tme_normal_vs_exact_lti():
    """reference tests/test_one_dim_moments.py:90-118: TME-3 of an OU step vs the exact discretisation."""
    F, Sigma = math.exp(-dt / ell), sigma ** 2 * (1 - math.exp(-2 * dt / ell))
    N = 3
    rms_f, cms_f, _, mean_f, mv_f = tme_sympy.sde_cond_moments_tme_1d(drift, dispersion, dt, 3, 2 * N)
    nrm = tme_sympy.sde_cond_moments_tme_normal_1d(drift, dispersion, dt, 3, N)
    x = np.array([-1.2, 0.1, 0.9])
    m, v = mv_f(x)
    npt.assert_allclose(m, F * x, rtol=1e-6)
    npt.assert_allclose(v, Sigma, rtol=1e-4)
    exact = np.stack([o.raw_moment_of_normal(F * x, Sigma, p) for p in range(2 * N)], axis=-1)
    npt.assert_allclose(rms_f(x, np.arange(2 * N)), exact, rtol=1e-4, atol=1e-8)
    npt.assert_allclose(nrm[0](x, np.arange(2 * N)), exact, rtol=1e-4, atol=1e-8)
    exact_c = np.stack([o.raw_moment_of_normal(F * x - 0.2, Sigma, p) for p in range(2 * N)], axis=-1)
    npt.assert_allclose(cms_f(x, np.arange(2 * N), 0.2), exact_c, rtol=1e-4, atol=1e-8)


def test_benes_tme_matches_exact_law():
    """Anchor for the (otherwise unpinned) Benes model: X' | x is the mixture 1/2 (1 +- tanh x) N(x +- dt, dt); the
    TME mean x + tanh(x) dt is exact and so is the TME-(>=2) variance dt + (1 - tanh^2 x) dt^2."""
    bdt, _, _, bdrift, bdisp, _, _ = models.benes_bernoulli(3)
    x = np.linspace(-2.5, 2.5, 11)
    t = np.tanh(x)
    for order in (2, 3):
        rms_f, _, _, mean_f, mv_f = tme_sympy.sde_cond_moments_tme_1d(bdrift, bdisp, bdt, order, 6)
        m, v = mv_f(x)
        npt.assert_allclose(m, x + t * bdt, rtol=1e-14)
        npt.assert_allclose(v, bdt + (1 - t ** 2) * bdt ** 2, rtol=1e-12)
        npt.assert_allclose(mean_f(x), x + t * bdt, rtol=1e-14)
    # exact raw moments of the mixture vs TME-3 raw moments: agreement to O(dt^4)
    rms_f, *_ = tme_sympy.sde_cond_moments_tme_1d(bdrift, bdisp, bdt, 3, 6)
    exact = np.stack([0.5 * (1 + t) * o.raw_moment_of_normal(x + bdt, bdt, p)
                      + 0.5 * (1 - t) * o.raw_moment_of_normal(x - bdt, bdt, p) for p in range(6)], axis=-1)
    npt.assert_allclose(rms_f(x, np.arange(6)), exact, rtol=0, atol=5e-6)


def test_tme_direct_differentiation_small_case():
    """The operator form is checked against brute-force symbolic differentiation of phi for a cubic drift."""
    xs = sp.Symbol('x', real=True)
    a, g = xs * (1 - 3 * xs ** 2), sp.Matrix([[1]])
    phi = (xs - sp.Rational(1, 5)) ** 5
    e = tme_sympy.expectation_expr(phi, [xs], [a], g, 0.01, 2)
    A1 = a * sp.diff(phi, xs) + sp.Rational(1, 2) * sp.diff(phi, xs, 2)
    A2 = a * sp.diff(A1, xs) + sp.Rational(1, 2) * sp.diff(A1, xs, 2)
    ref = phi + 0.01 * A1 + 0.01 ** 2 / 2 * A2
    for xv in (-0.7, 0.2, 1.1):
        assert abs(float(e.subs(xs, xv)) - float(ref.subs(xs, xv))) < 1e-13


def test_ldl_chol_equals_cholesky_on_pd():
    """reference tests/test_utils.py:198-209."""
    rs2 = np.random.RandomState(1)
    a = rs2.randn(6, 6)
    a = a @ a.T + 6 * np.eye(6)
    l, d = o.ldl(a)
    npt.assert_allclose(l @ np.diag(d) @ l.T, a, rtol=1e-12)
    npt.assert_allclose(o.ldl_chol(a), np.linalg.cholesky(a), rtol=1e-12)


def test_gaussian_sum_and_nan_poisoning():
    ic = o.GaussianSum1D.new([-0.5, 0.5], [0.05, 0.05], [0.5, 0.5], N=4)
    npt.assert_allclose(ic.rms[:3], [1., 0., 0.05 + 0.25])
    npt.assert_allclose(ic.cms, ic.rms)  # centre is 0
    npt.assert_allclose(ic.scms[2], 1.)
    # a non-PD Hankel matrix poisons the quadrature with NaN rather than raising
    w, x = o.moment_quadrature(np.array([1., 0., -1., 0.]))
    assert np.all(np.isnan(w)) and np.all(np.isnan(x))


def test_odd_moment_count_warns():
    m = models.ou_gaussian(2)
    with pytest.warns(UserWarning):
        o.moment_filter_rms(m['cond_rms'], m['pdf'], m['rms0'][:3], ys[:0])
