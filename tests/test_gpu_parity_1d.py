"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on identical seeded inputs.

Tolerance (BASELINE.json north_star): 1e-6 relative on fp64 moments / NLL.  Moments that are mathematically zero
(odd central moments of symmetric laws, cms[1]) get an absolute floor scaled by the moment's natural magnitude.
"""
import math

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth, sym
from mfs_amd.one_dim import filtering, moments, ss_models, quadtures
from oracle import one_dim as o, models as om, tme_sympy

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def _assert_moments(got, ref, rtol=RTOL):
    """Relative error per moment order, with the denominator max(|ref|, typical magnitude of that order)."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape
    nan_g, nan_r = np.isnan(got), np.isnan(ref)
    assert np.array_equal(nan_g, nan_r), 'NaN pattern differs'
    ok = ~nan_r
    # per-order magnitude: the largest |moment| of that order over the run, floored by the neighbouring even order
    # (odd central moments of near-symmetric laws and cms[1] are rounding noise around zero)
    colmax = np.nanmax(np.abs(ref), axis=0)
    even_neighbour = np.maximum(colmax, np.concatenate([colmax[1:], colmax[-1:]]))
    scale = np.maximum(np.abs(ref), even_neighbour[None, :] * 1e-3 + 1e-300)
    err = np.abs(got - ref)[ok] / scale[ok]
    assert err.size == 0 or err.max() <= rtol, f'max scaled error {err.max():.3e} > {rtol}'


def _benes(N, tme_order, normal=False):
    dt, T, ts, ic, drift, dispersion, logistic, pmf, _ = ss_models.benes_bernoulli(N)
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    if normal:
        dev = moments.sde_cond_moments_tme_normal(drift, dispersion, dt, tme_order, N)
        ora = tme_sympy.sde_cond_moments_tme_normal_1d(odrift, odisp, odt, tme_order, N)
    else:
        dev = moments.sde_cond_moments_tme(drift, dispersion, dt, tme_order)
        ora = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, tme_order, 2 * N)
    return dt, ic, pmf, dev, oic, opmf, ora


@pytest.mark.parametrize('N', [2, 3, 5, 8, 13, 16, 20])
def test_quadrature_matches_oracle(N):
    """Cholesky / triangular solves / eigensolve in isolation (mfs/one_dim/quadtures.py:83-133), incl. odd N.

    Inputs are standardised (scaled central) moments of random Gaussian mixtures, as the filters feed the
    quadrature.  The Hankel matrix has cond ~ 1e8 (N = 8) ... 1e16 (N = 20), so node-by-node agreement between two
    correct fp64 implementations degrades with N; the size-independent property checked at every N is that the rule
    reproduces all 2N input moments (reference tests/test_one_dim_quadrature.py:86-113)."""
    rng = np.random.default_rng(N)
    B = 64
    ms, means, scales = np.empty((B, 2 * N)), np.empty(B), np.empty(B)
    for b in range(B):
        ws = rng.dirichlet(np.ones(3))
        mus, vs = rng.normal(scale=0.8, size=3), rng.uniform(0.05, 0.6, size=3)
        rms = np.array([sum(w * float(o.raw_moment_of_normal(m, v, p)) for w, m, v in zip(ws, mus, vs))
                        for p in range(2 * N)])
        means[b], scales[b] = rms[1], math.sqrt(rms[2] - rms[1] ** 2)
        cen = np.array([sum(w * float(o.raw_moment_of_normal(m - rms[1], v, p)) for w, m, v in zip(ws, mus, vs))
                        for p in range(2 * N)])
        ms[b] = cen / scales[b] ** np.arange(2 * N)
    w, x = quadtures.moment_quadrature(ms, means, scales)
    node_tol = 1e-9 if N <= 5 else 1e-7 if N <= 8 else 1e-5 if N <= 13 else 1e-3
    for b in range(B):
        wr, xr = o.moment_quadrature(ms[b], means[b], scales[b])
        assert np.all(np.isfinite(w[b])) and np.all(np.isfinite(x[b]))
        ia, ib = np.argsort(x[b]), np.argsort(xr)
        npt.assert_allclose(x[b][ia], xr[ib], rtol=node_tol, atol=node_tol)
        z = (x[b] - means[b]) / scales[b]
        for p in range(2 * N):
            npt.assert_allclose(np.dot(w[b], z ** p), ms[b][p], rtol=1e-6, atol=1e-7 * max(1., abs(ms[b][p])))
        if N <= 8:
            npt.assert_allclose(w[b][ia], wr[ib], rtol=1e-6, atol=1e-10)


def test_quadrature_nan_poisoning_and_ldl():
    bad = np.array([[1., 0., -1., 0.], [1., 0., 1., 0.]])
    w, x = quadtures.moment_quadrature(bad)
    assert np.all(np.isnan(w[0])) and np.all(np.isnan(x[0]))
    npt.assert_allclose(np.sort(x[1]), [-1., 1.], atol=1e-14)
    npt.assert_allclose(w[1], [0.5, 0.5], atol=1e-14)
    # stable=True (LDL completion, mfs/utils.py:495-538) equals Cholesky on PD input and completes an indefinite one
    N = 5
    ms = np.array([[float(o.raw_moment_of_normal(0.2, 0.7, p)) for p in range(2 * N)]])
    w1, x1 = quadtures.moment_quadrature(ms)
    w2, x2 = quadtures.moment_quadrature(ms, ldl=True)
    npt.assert_allclose(np.sort(x1[0]), np.sort(x2[0]), rtol=1e-9)
    ind = np.array([1., 0., 1., 0., 0.5, 0.])  # G is indefinite: 0.5 < 1
    wl, xl = quadtures.moment_quadrature(ind[None, :], ldl=True)
    wo, xo = o.moment_quadrature(ind, ldl=True)
    assert np.all(np.isfinite(wl))
    npt.assert_allclose(np.sort(xl[0]), np.sort(xo), rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('mode', ['raw', 'central', 'scaled'])
@pytest.mark.parametrize('N,tme_order,normal', [(7, 3, False), (4, 2, False), (7, 3, True), (10, 2, True)])
def test_benes_bernoulli_config1(mode, N, tme_order, normal):
    """BASELINE config 1 (N = 7, T = 100, TME-3) and neighbours, all three modes, B = 3 replicates."""
    dt, ic, pmf, dev, oic, opmf, ora = _benes(N, tme_order, normal)
    T, B = 100, 3
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=N)
    if mode == 'raw':
        m, nell = filtering.moment_filter_rms(dev[0], pmf, ic.rms, ys)
        ref = [o.moment_filter_rms(ora[0], opmf, oic.rms, ys[b]) for b in range(B)]
        extra = []
    elif mode == 'central':
        m, means, nell = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys)
        ref = [o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[b]) for b in range(B)]
        extra = [means]
    else:
        s0 = math.sqrt(ic.variance)
        m, means, scales, nell = filtering.moment_filter_scms(dev[2], dev[4], pmf, ic.scms, ic.mean, s0, ys)
        ref = [o.moment_filter_scms(ora[2], ora[4], opmf, oic.scms, oic.mean, s0, ys[b]) for b in range(B)]
        extra = [means, scales]
    for b in range(B):
        _assert_moments(m[b], ref[b][0])
        npt.assert_allclose(nell[b], ref[b][-1], rtol=RTOL)
        for e, r in zip(extra, ref[b][1:-1]):
            npt.assert_allclose(e[b], r, rtol=RTOL, atol=1e-9)


@pytest.mark.parametrize('N', [5, 10, 15, 20, 25])
def test_ou_convergence_config3(N):
    """BASELINE config 3 (dardel/convergence/convergence_mf.py): central mode vs the oracle AND vs the exact KF."""
    m = om.ou_gaussian(N)
    T, B = 200, 4
    ys, _ = synth.ou_gaussian_batch(B, T, seed=3)
    F, Sigma = m['F'], m['Sigma']
    _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_normal(lambda x: F * x, lambda x: Sigma)
    from mfs_amd import stats
    cmss, means, nell = filtering.moment_filter_cms(cond_cms, lambda x: F * x,
                                                    lambda y, x: stats.norm_pdf(y, x, 1.), m['cms0'], m['mean0'], ys)
    alive = 0
    for b in range(B):
        r_cmss, r_means, r_nell = o.moment_filter_cms(m['cond_cms'], m['cond_mean'], m['pdf'], m['cms0'], m['mean0'],
                                                      ys[b])
        if N >= 25 and (np.isnan(nell[b]) or np.isnan(r_nell)):
            # cond(Hankel) ~ 1e20 at N = 25: whether a Cholesky pivot rounds to <= 0 is summation-order luck, so a
            # replicate may NaN-poison in one fp64 implementation and not in another (hard part 3)
            continue
        alive += 1
        npt.assert_allclose(nell[b], r_nell, rtol=RTOL)
        npt.assert_allclose(means[b], r_means, rtol=RTOL, atol=1e-8)
        npt.assert_allclose(cmss[b][:, 2], r_cmss[:, 2], rtol=RTOL)
        if N <= 15:
            _assert_moments(cmss[b], r_cmss, rtol=1e-5 if N > 10 else RTOL)
        kf_m, kf_v, kf_nell = m['kf'](ys[b])
        tol = {5: 5e-3, 10: 1e-5, 15: 1e-7, 20: 1e-8, 25: 1e-8}[N]
        assert np.abs(means[b] - kf_m).max() < tol
        npt.assert_allclose(nell[b], kf_nell, rtol=max(tol, 1e-9))
    assert alive >= B - 1


def test_well_poisson_per_replicate_parameters_config4():
    """BASELINE config 4 in miniature: central, TME-normal-2 and Euler, theta per replicate, NLL parity."""
    N, T = 7, 150
    dt, _, ts, ic, drift, dispersion, emission, pmf, _ = ss_models.well_poisson(3., N)
    odt, _, oic, odrift, odisp, oemis, opmf = om.well_poisson(N)
    p1 = np.array([0.5, 2., 3., 4.5, 6.])
    p2 = np.array([1., 3., 2.5, 0.5, 6.])
    B = len(p1)
    ys, _ = synth.well_poisson_batch(B, T, p1=3., p2=3., dt=dt, seed=11)
    for euler in (False, True):
        if euler:
            _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_euler(lambda x: drift(x, p1), dispersion, dt, N)
        else:
            _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, p1), dispersion,
                                                                               dt, 2, N)
        cmss, means, nell = filtering.moment_filter_cms(cond_cms, cond_mean, lambda y, x: pmf(y, x, p2),
                                                        ic.cms, ic.mean, ys)
        for b in range(B):
            d = (lambda p: (lambda x: odrift(x, p)))(p1[b])
            if euler:
                ora = tme_sympy.sde_cond_moments_euler_1d(d, odisp, odt, N)
            else:
                ora = tme_sympy.sde_cond_moments_tme_normal_1d(d, odisp, odt, 2, N)
            r_cmss, r_means, r_nell = o.moment_filter_cms(ora[1], ora[3], (lambda p: lambda y, x: opmf(y, x, p))(p2[b]),
                                                          oic.cms, oic.mean, ys[b])
            if np.isnan(r_nell):
                assert np.isnan(nell[b])
                continue
            npt.assert_allclose(nell[b], r_nell, rtol=RTOL)
            npt.assert_allclose(means[b], r_means, rtol=RTOL, atol=1e-9)
            _assert_moments(cmss[b], r_cmss)


def test_batch_shapes_empty_and_ragged_blocks():
    """Edge cases: B = 0, T = 0, B not a multiple of the filters-per-block, (2N,) vs (B, 2N) initial moments."""
    N = 4
    dt, ic, pmf, dev, oic, opmf, ora = _benes(N, 2)
    ys, _ = synth.benes_bernoulli_batch(19, 30, dt, seed=5)
    m_shared, means_s, nell_s = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys)
    m_b, means_b, nell_b = filtering.moment_filter_cms(dev[1], dev[3], pmf, np.tile(ic.cms, (19, 1)),
                                                       np.full(19, ic.mean), ys)
    npt.assert_array_equal(m_shared, m_b)
    npt.assert_array_equal(nell_s, nell_b)
    m1, means1, nell1 = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys[7])
    npt.assert_array_equal(m1, m_shared[7])
    assert np.ndim(nell1) == 0 and nell1 == nell_s[7]
    m0, means0, nell0 = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, np.zeros((0, 30)))
    assert m0.shape == (0, 30, 2 * N) and nell0.shape == (0,)
    mt, meanst, nellt = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, np.zeros((3, 0)))
    assert mt.shape == (3, 0, 2 * N) and np.all(nellt == 0.)


def test_nan_poisoning_is_per_replicate():
    """A replicate whose moment matrix loses positive definiteness emits NaN from that step on and never disturbs
    its neighbours; first_nan agrees with the oracle within a few steps."""
    N = 7
    dt, ic, pmf, dev, oic, opmf, ora = _benes(N, 3)
    T, B = 60, 6
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=9)
    cms0 = np.tile(ic.cms, (B, 1))
    cms0[2, 4] = -1.  # negative 4th central moment: G is indefinite from the first step
    m, means, nell, first_nan = filtering.moment_filter_cms(dev[1], dev[3], pmf, cms0, np.full(B, ic.mean), ys,
                                                            return_first_nan=True)
    assert first_nan[2] == 0 and np.all(np.isnan(m[2])) and np.isnan(nell[2])
    good = [b for b in range(B) if b != 2]
    assert np.all(first_nan[good] == -1) and np.all(np.isfinite(nell[good]))
    r = o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[3])
    npt.assert_allclose(nell[3], r[2], rtol=RTOL)
    r_bad = o.moment_filter_cms(ora[1], ora[3], opmf, cms0[2], ic.mean, ys[2])
    assert np.all(np.isnan(r_bad[0])) and np.isnan(r_bad[2])


def test_routines_equivalence_on_device():
    """reference tests/test_filtering.py:113-164 run on the HIP path: rms vs cms vs scms agree."""
    N = 4
    rs = np.random.RandomState(666)
    dtt, T, ell, sigma = 1e-2, 100, 1., 0.5
    ts = np.linspace(dtt, dtt * T, T)
    cov = np.exp(-np.abs(ts[None, :] - ts[:, None]) / ell) * sigma ** 2
    ys = np.linalg.cholesky(cov) @ rs.randn(T) + rs.randn(T)
    from mfs_amd import stats
    b = math.sqrt(2) * sigma / math.sqrt(ell)
    r, c, s, mu, mv = moments.sde_cond_moments_tme(lambda x: -x / ell, lambda _: b, dtt, 2)
    rms0 = np.array([moments.raw_moment_of_normal(0., 0.5, p) for p in range(2 * N)])
    cms0, scms0 = moments.raw_to_central(rms0), moments.raw_to_scaled(rms0)

    def pdf(y, x):
        return stats.norm_pdf(y, x, 1.)

    rmss, nell_r = filtering.moment_filter_rms(lambda x, n: s(x, n, 0., 1.), pdf, rms0, ys)
    cmss, means_c, nell_c = filtering.moment_filter_cms(lambda x, n, m: s(x, n, m, 1.), mu, pdf, cms0, 0., ys)
    scmss, means, scales, nell_s = filtering.moment_filter_scms(s, mv, pdf, scms0, 0., math.sqrt(0.5), ys)
    npt.assert_array_almost_equal(cmss, moments.raw_to_central(rmss), decimal=10)
    npt.assert_array_almost_equal(scmss, moments.raw_to_scaled(rmss), decimal=9)
    npt.assert_array_almost_equal(means_c, means, decimal=13)
    npt.assert_array_almost_equal(rmss[:, 2] - rmss[:, 1] ** 2, scales ** 2, decimal=11)
    for nell in (nell_s, nell_c):
        npt.assert_array_almost_equal(nell_r, nell, decimal=10)


@pytest.mark.parametrize('N', [3, 7, 12, 15])
def test_fast_and_dense_device_paths_agree(N, monkeypatch):
    """The default register-resident path (Jacobi matrix from the Cholesky pivots + tridiagonal Laguerre eigensolve) and
    the dense LDS path (both triangular solves in full + cyclic Jacobi on the dense K) are two independent device
    implementations of mfs/one_dim/quadtures.py:122-133; they must agree far inside the 1e-6 parity bar."""
    dt, ic, pmf, dev, oic, opmf, ora = _benes(N, 3)
    T, B = 120, 6
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=40 + N)
    monkeypatch.delenv('MFS_SOLVER', raising=False)
    m_f, means_f, nell_f, fn_f = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys,
                                                             return_first_nan=True)
    monkeypatch.setenv('MFS_SOLVER', 'dense')
    m_d, means_d, nell_d, fn_d = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys,
                                                             return_first_nan=True)
    both = (fn_f < 0) & (fn_d < 0)
    assert both.sum() >= B - 2
    npt.assert_allclose(nell_f[both], nell_d[both], rtol=1e-8)
    npt.assert_allclose(means_f[both], means_d[both], rtol=1e-7, atol=1e-10)
    for b in np.where(both)[0]:
        _assert_moments(m_f[b], m_d[b], rtol=1e-6)
    # quadrature level, incl. the N = 16..32 instantiations of both paths
    for Nq in (N, N + 16):
        rms = np.array([[float(o.raw_moment_of_normal(0.1, 0.9, p)) for p in range(2 * Nq)]])
        cms = o.raw_to_central(rms[0])[None, :]
        monkeypatch.delenv('MFS_SOLVER', raising=False)
        w1, x1 = quadtures.moment_quadrature(cms, 0.1)
        monkeypatch.setenv('MFS_SOLVER', 'dense')
        w2, x2 = quadtures.moment_quadrature(cms, 0.1)
        i1, i2 = np.argsort(x1[0]), np.argsort(x2[0])
        tol = 1e-10 if Nq <= 12 else 1e-7 if Nq <= 20 else 1e-3
        npt.assert_allclose(x1[0][i1], x2[0][i2], rtol=tol, atol=tol)
        big = w2[0][i2] > 1e-8
        npt.assert_allclose(w1[0][i1][big], w2[0][i2][big], rtol=max(tol, 1e-9) * 10)


@pytest.mark.parametrize('N', [7, 8, 15, 16, 31, 32])
def test_lane_group_boundaries_and_maximum_order(N):
    """N = 7 | 8, 15 | 16, 31 | 32 straddle the 8 / 16 / 32 / 64 lanes-per-filter instantiations; N = 32 is MFS_MAX_N.
    Scaled central moments of a Gaussian: the rule must be the Gauss-Hermite rule (nodes to a conditioning-dependent
    tolerance) and a short OU filter must track the exact Kalman filter."""
    cms = np.array([[o.central_moment_of_normal(1., p) for p in range(2 * N)]])
    w, x = quadtures.moment_quadrature(cms, 0.25, 1.5)
    gh_x, gh_w = np.polynomial.hermite_e.hermegauss(N)
    gh_w = gh_w / math.sqrt(2 * math.pi)
    assert np.all(np.isfinite(w)) and np.all(np.isfinite(x))
    order = np.argsort(x[0])
    tol = 1e-10 if N <= 8 else 1e-7 if N <= 16 else 1e-2
    npt.assert_allclose(x[0][order], 0.25 + 1.5 * gh_x, rtol=tol, atol=tol)
    npt.assert_allclose(w[0].sum(), 1., rtol=1e-12)
    big = gh_w > 1e-6
    npt.assert_allclose(w[0][order][big], gh_w[big], rtol=max(tol * 100, 1e-8))
    # a short filter run at this order
    mdl = om.ou_gaussian(N)
    F, Sigma = mdl['F'], mdl['Sigma']
    ys, _ = synth.ou_gaussian_batch(3, 40, seed=N)
    from mfs_amd import stats
    _, cc, _, cm, _ = moments.sde_cond_moments_normal(lambda xx: F * xx, lambda xx: Sigma)
    cmss, means, nell = filtering.moment_filter_cms(cc, cm, lambda y, xx: stats.norm_pdf(y, xx, 1.), mdl['cms0'],
                                                    mdl['mean0'], ys)
    for b in range(3):
        if not np.isfinite(nell[b]):  # N >= 25: a pivot may round below zero (see test_ou_convergence_config3)
            assert N >= 25
            continue
        kf_m, kf_v, kf_nell = mdl['kf'](ys[b])
        assert np.abs(means[b] - kf_m).max() < (2e-3 if N <= 8 else 1e-6)
        npt.assert_allclose(nell[b], kf_nell, rtol=1e-4 if N <= 8 else 1e-8)


def test_kernel_elementary_functions_against_libm():
    """The fast kernel's in-line exp / tanh / log (mfs_elementary): a few ulp of the result, tanh in ABSOLUTE terms near
    zero (it feeds polynomials in tanh x), non-finite in -> non-finite out."""
    import ctypes as C
    from mfs_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(0)

    def run(which, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        _lib.check(L.mfs_elementary(which, x.size, _lib.ptr(x), _lib.ptr(out), 0))
        return out

    x = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(-2, 2, 20000), [0.0, -0.0, 1e-300, -745.0, 709.7]])
    got, ref = run(0, x), np.exp(x)
    assert np.max(np.abs(got - ref) / ref) < 4e-16
    assert np.all(run(0, np.array([709.9, 800.0, np.inf])) == np.inf) and np.all(run(0, np.array([-800.0, -np.inf])) == 0.0)
    x = np.concatenate([rng.uniform(-25, 25, 20000), rng.uniform(-1e-3, 1e-3, 20000), [0.0, 30.0, -30.0, 1e-200]])
    got, ref = run(1, x), np.tanh(x)
    assert np.max(np.abs(got - ref)) < 3e-16
    big = np.abs(x) > 1e-12
    assert np.all(np.sign(got[big]) == np.sign(x[big]))
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0.5, 2.0, 20000), [1.0, 5e-324, 1.7e308]])
    got, ref = run(2, x), np.log(x)
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)) < 4e-16
    bad = run(2, np.array([0.0, -1.0, np.nan, np.inf]))
    assert not np.any(np.isfinite(bad))
    assert np.all(np.isnan(run(0, np.array([np.nan])))) and np.all(np.isnan(run(1, np.array([np.nan]))))


@pytest.mark.parametrize('N', [14, 15, 16])
def test_register_budget_builds_agree_bitwise(N):
    """N = 14..16 exist in two builds of the same kernel (256 registers / 512 registers, csrc/filter1d_fast.hpp); the
    plan picks the wide one when the batch puts at most one wave on a SIMD.  Eight measurement sequences repeated over a
    batch large enough for the 256-register build must give exactly the numbers the small batch (wide build) gives."""
    dt, T, ts, ic, drift, dispersion, logistic, pmf, _ = ss_models.benes_bernoulli(N)
    _, c, _, mu, _ = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    ys8, _ = synth.benes_bernoulli_batch(8, 40, dt, seed=11)
    cm_s, me_s, nell_s = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys8)
    reps = 4200 // (64 // (16 if N < 16 else 32)) + 8    # more single-wave workgroups than the 1024 SIMDs of an MI355X
    ys = np.tile(ys8, (reps, 1))
    cm_l, me_l, nell_l = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
    assert np.isfinite(nell_s).any()
    for r in (0, reps // 2, reps - 1):
        sl = slice(8 * r, 8 * r + 8)
        assert np.array_equal(nell_l[sl], nell_s, equal_nan=True)
        assert np.array_equal(me_l[sl], me_s, equal_nan=True)
        assert np.array_equal(cm_l[sl], cm_s, equal_nan=True)


@pytest.mark.parametrize('mode', ['raw', 'central', 'scaled'])
def test_odd_moment_count_proceeds_like_the_reference(mode):
    """An odd number of moments: the reference warns and proceeds (mfs/one_dim/filtering.py:65-66) with
    N = floor(M / 2) (mfs/one_dim/quadtures.py:122).  The first 2N columns are the even-count filter's, the last one is
    the N-node rule's order-2N moment, as in the oracle."""
    N, T, B = 4, 40, 3
    dt, ic9, pmf, dev, oic9, opmf, ora = _benes(N, 2)
    # 2N + 1 = 9 initial moments of the same mixture
    from mfs_amd.utils import GaussianSum1D
    big = GaussianSum1D.new(means=[-0.5, 0.5], variances=[0.05, 0.05], weights=[0.5, 0.5], N=N + 1)
    ora9 = tme_sympy.sde_cond_moments_tme_1d(om.benes_bernoulli(N)[3], om.benes_bernoulli(N)[4], dt, 2, 2 * N + 1)
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=31)
    s0 = math.sqrt(big.variance)
    with pytest.warns(UserWarning, match='not odd'):
        if mode == 'raw':
            got = filtering.moment_filter_rms(dev[0], pmf, big.rms[:9], ys)
            even = filtering.moment_filter_rms(dev[0], pmf, big.rms[:8], ys)
            ref = [o.moment_filter_rms(ora9[0], opmf, big.rms[:9], y) for y in ys]
        elif mode == 'central':
            got = filtering.moment_filter_cms(dev[1], dev[3], pmf, big.cms[:9], big.mean, ys)
            even = filtering.moment_filter_cms(dev[1], dev[3], pmf, big.cms[:8], big.mean, ys)
            ref = [o.moment_filter_cms(ora9[1], ora9[3], opmf, big.cms[:9], big.mean, y) for y in ys]
        else:
            got = filtering.moment_filter_scms(dev[2], dev[4], pmf, big.scms[:9], big.mean, s0, ys)
            even = filtering.moment_filter_scms(dev[2], dev[4], pmf, big.scms[:8], big.mean, s0, ys)
            ref = [o.moment_filter_scms(ora9[2], ora9[4], opmf, big.scms[:9], big.mean, s0, y) for y in ys]
    assert got[0].shape == (B, T, 9)
    npt.assert_allclose(got[-1], even[-1], rtol=1e-9)                      # the extra entry never feeds back
    npt.assert_allclose(got[0][..., :8], even[0], rtol=1e-7, atol=1e-12)
    for b in range(B):
        npt.assert_allclose(got[-1][b], ref[b][-1], rtol=RTOL)
        _assert_moments(got[0][b], ref[b][0])


def test_poisson_counts_beyond_the_log_factorial_table():
    """The Poisson likelihood looks log(y!) up for y <= 32 and computes it for larger counts: a trajectory with counts of 40
    and 75 among ordinary ones, against the oracle (scipy's pmf)."""
    N, T = 5, 40
    dt, _, ts, ic, drift, dispersion, emission, pmf, _ = ss_models.well_poisson(3., N)
    odt, _, oic, odrift, odisp, oemis, opmf = om.well_poisson(N)
    ys, _ = synth.well_poisson_batch(2, T, p1=3., p2=3., dt=dt, seed=4)
    ys = ys.astype(np.float64)
    ys[0, 7], ys[0, 19], ys[1, 3] = 40., 75., 33.
    _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, 3.), dispersion, dt, 2, N)
    cmss, means, nell = filtering.moment_filter_cms(cond_cms, cond_mean, lambda y, x: pmf(y, x, 3.), ic.cms, ic.mean, ys)
    ora = tme_sympy.sde_cond_moments_tme_normal_1d(lambda x: odrift(x, 3.), odisp, odt, 2, N)
    for b in range(2):
        r_cmss, r_means, r_nell = o.moment_filter_cms(ora[1], ora[3], lambda y, x: opmf(y, x, 3.), oic.cms, oic.mean, ys[b])
        # (an absurd count puts the whole posterior on the outermost node: the next Hankel matrix is numerically singular and
        #  whether its Cholesky survives is rounding luck on either side -- the step of the large count itself must agree)
        both = np.isfinite(r_means) & np.isfinite(means[b])
        first_large = 7 if b == 0 else 3
        assert both[:first_large + 1].all()
        npt.assert_allclose(means[b][both], r_means[both], rtol=RTOL, atol=1e-9)
        _assert_moments(cmss[b][both], r_cmss[both])


def test_non_finite_measurement_poisons_from_that_step():
    """A NaN measurement (a missing value handed through) makes the likelihood NaN: the replicate is NaN from that step on, as
    in the reference's scan, its neighbours are untouched, and first_nan reports the step.  Gaussian measurements, OU model
    (the Bernoulli pmf `y ? p : 1 - p` would silently read NaN as 0)."""
    m = om.ou_gaussian(5)
    T, B = 30, 3
    ys, _ = synth.ou_gaussian_batch(B, T, seed=2)
    ys[1, 11] = np.nan
    from mfs_amd import stats
    F, Sigma = m['F'], m['Sigma']
    _, cond_cms, _, cond_mean, _ = moments.sde_cond_moments_normal(lambda x: F * x, lambda x: Sigma)
    mm, means, nell, fn = filtering.moment_filter_cms(cond_cms, lambda x: F * x, lambda y, x: stats.norm_pdf(y, x, 1.),
                                                      m['cms0'], m['mean0'], ys, return_first_nan=True)
    assert list(fn) == [-1, 11, -1]
    assert np.all(np.isfinite(mm[1, :11])) and np.all(np.isnan(mm[1, 11:])) and np.isnan(nell[1])
    for b in (0, 2):
        r = o.moment_filter_cms(m['cond_cms'], m['cond_mean'], m['pdf'], m['cms0'], m['mean0'], ys[b])
        npt.assert_allclose(nell[b], r[2], rtol=RTOL)
    r = o.moment_filter_cms(m['cond_cms'], m['cond_mean'], m['pdf'], m['cms0'], m['mean0'], ys[1])
    assert np.all(np.isfinite(r[1][:11])) and np.all(np.isnan(r[1][11:]))
