"""The C port of the oracle (bench.py's cpu_baseline) against the NumPy oracle."""
import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth
from oracle import c_oracle, tme_sympy, one_dim as o, models as om


@pytest.mark.parametrize('N', [3, 6, 11])
def test_c_quadrature_matches_numpy(N):
    rng = np.random.default_rng(N)
    ms = np.array([[float(o.raw_moment_of_normal(m, v, p)) for p in range(2 * N)]
                   for m, v in zip(rng.normal(scale=0.3, size=5), rng.uniform(0.5, 1.5, size=5))])
    w, x = c_oracle.quadrature_1d(ms)
    for b in range(5):
        wr, xr = o.moment_quadrature(ms[b])
        npt.assert_allclose(np.sort(x[b]), np.sort(xr), rtol=1e-7, atol=1e-9)
        npt.assert_allclose(w[b][np.argsort(x[b])], wr[np.argsort(xr)], rtol=1e-6, atol=1e-12)
    w, x = c_oracle.quadrature_1d(np.array([[1., 0., -1., 0.]]))
    assert np.all(np.isnan(w)) and np.all(np.isnan(x))


@pytest.mark.parametrize('mode', [0, 1, 2])
def test_c_filter_matches_numpy_benes(mode):
    N, T, B = 7, 100, 3
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    tab = tme_sympy.operator_tables_1d(odrift, odisp, odt, 3, 'tanh')
    ys, _ = synth.benes_bernoulli_batch(B, T, odt, seed=2)
    fns = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 3, 2 * N)
    m0 = [oic.rms, oic.cms, oic.scms][mode]
    s0 = np.sqrt(oic.variance)
    m, means, scales, nell = c_oracle.filter_1d(mode, N, ys, m0, oic.mean, s0, 0, 1, 6, tab, 1.0, 0,
                                                np.array([0., 0., 0., 0.2]))
    for b in range(B):
        if mode == 0:
            r = o.moment_filter_rms(fns[0], opmf, oic.rms, ys[b])
        elif mode == 1:
            r = o.moment_filter_cms(fns[1], fns[3], opmf, oic.cms, oic.mean, ys[b])
            npt.assert_allclose(means[b], r[1], rtol=1e-7, atol=1e-10)
        else:
            r = o.moment_filter_scms(fns[2], fns[4], opmf, oic.scms, oic.mean, s0, ys[b])
            npt.assert_allclose(scales[b], r[2], rtol=1e-7)
        npt.assert_allclose(nell[b], r[-1], rtol=1e-8)
        npt.assert_allclose(m[b][:, 2], r[0][:, 2], rtol=1e-6)


def test_c_filter_gaussian_closure_poisson():
    N, T = 5, 80
    odt, _, oic, odrift, odisp, _, opmf = om.well_poisson(N)
    p1, p2 = 2.5, 1.5
    tab = tme_sympy.operator_tables_1d(lambda x: odrift(x, p1), odisp, odt, 2, 'x')
    coef = np.zeros((2, tab.shape[1]))
    coef[0] = tab[0]
    coef[0, 1] += 1.  # mean = x + Q_1
    coef[1] = tab[-1]
    ys, _ = synth.well_poisson_batch(2, T, p1=p1, p2=p2, dt=odt, seed=4)
    fns = tme_sympy.sde_cond_moments_tme_normal_1d(lambda x: odrift(x, p1), odisp, odt, 2, N)
    m, means, _, nell = c_oracle.filter_1d(1, N, ys, oic.cms, oic.mean, None, 1, 0, 0, coef, 0.0, 1, np.array([p2]))
    for b in range(2):
        r = o.moment_filter_cms(fns[1], fns[3], lambda y, x: opmf(y, x, p2), oic.cms, oic.mean, ys[b])
        npt.assert_allclose(nell[b], r[2], rtol=1e-8)
        npt.assert_allclose(means[b], r[1], rtol=1e-7, atol=1e-10)
