"""Pin the N-D oracle: golden multi-index tables from the reference module + the reference's analytic tests.

reference tests/test_multi_indices.py, tests/test_multi_dim_quadrature.py:71-79,86-216,
tests/test_filtering.py:169-329, tests/test_multi_dim_moments.py.
"""
import math
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle import multi_dims as md
from oracle import one_dim as o
from oracle import models, tme_sympy


@pytest.fixture(scope='module')
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'multi_indices.npz'))


@pytest.mark.parametrize('N,d', [(3, 1), (3, 2), (4, 2), (5, 2), (6, 2), (2, 3), (3, 3)])
def test_multi_index_tables_match_reference(golden, N, d):
    npt.assert_array_equal(md.generate_graded_lexico_multi_indices(d, 2 * N - 1, 0), golden[f'mi_N{N}_d{d}'])
    npt.assert_array_equal(md.gram_and_hankel_indices_graded_lexico(N, d), golden[f'inds_N{N}_d{d}'])


def test_multi_index_probes_match_reference(golden):
    npt.assert_array_equal([md.graded_lexico_indexof_multi_index(list(p)) for p in golden['probe_mi']],
                           golden['probe_index'])
    npt.assert_array_equal([md.graded_lexico_indexof_multi_index(list(p), lower_sum=2)
                            for p in golden['probe_mi'][1:]], golden['probe_index_lower2'])
    npt.assert_array_equal([md.sizeof_multi_indices(*a) for a in golden['size_args']], golden['size_vals'])
    npt.assert_array_equal(md.generate_graded_lexico_multi_indices(3, 4, 2), golden['mi_lower_d3'])


def test_kan_moments_closed_forms():
    """reference tests/test_multi_dim_moments.py: Kan formula vs known Gaussian moments."""
    mean = np.array([0.3, -0.2])
    cov = np.array([[0.5, 0.1], [0.1, 0.3]])
    npt.assert_allclose(md.raw_moments_mvn_kan(mean, cov, [0, 0]), 1.)
    npt.assert_allclose(md.raw_moments_mvn_kan(mean, cov, [1, 0]), mean[0])
    npt.assert_allclose(md.raw_moments_mvn_kan(mean, cov, [1, 1]), cov[0, 1] + mean[0] * mean[1])
    npt.assert_allclose(md.raw_moments_mvn_kan(mean, cov, [2, 0]), cov[0, 0] + mean[0] ** 2)
    npt.assert_allclose(md.raw_moments_mvn_kan(np.zeros(2), cov, [2, 2]),
                        cov[0, 0] * cov[1, 1] + 2 * cov[0, 1] ** 2)
    npt.assert_allclose(md.raw_moments_mvn_kan(np.zeros(2), cov, [4, 0]), 3 * cov[0, 0] ** 2)
    # 1-D agreement with raw_moment_of_normal
    for p in range(8):
        npt.assert_allclose(md.raw_moments_mvn_kan(np.array([0.4]), np.array([[0.7]]), [p]),
                            float(o.raw_moment_of_normal(0.4, 0.7, p)), rtol=1e-12)


@pytest.mark.parametrize('N', [2, 3, 4])
def test_nd_quadrature_reproduces_input_moments(N):
    """reference tests/test_multi_dim_quadrature.py:86-216: the rule integrates every input monomial exactly."""
    d = 2
    mi = md.generate_graded_lexico_multi_indices(d, 2 * N - 1)
    inds = md.gram_and_hankel_indices_graded_lexico(N, d)
    mean = np.array([0.2, -0.1])
    cov = np.array([[0.6, 0.2], [0.2, 0.4]])
    rms = np.array([md.raw_moments_mvn_kan(mean, cov, m) for m in mi])
    w, x = md.moment_quadrature_nd(rms, inds)
    approx = np.einsum('i,ij->j', w, np.prod(x[:, None, :] ** mi[None, :, :], axis=-1))
    npt.assert_allclose(approx, rms, rtol=1e-9, atol=1e-11)
    # central mode gives the same integrals
    cms = np.array([md.raw_moments_mvn_kan(np.zeros(2), cov, m) for m in mi])
    wc, xc = md.moment_quadrature_nd(cms, inds, mean)
    approx_c = np.einsum('i,ij->j', wc, np.prod(xc[:, None, :] ** mi[None, :, :], axis=-1))
    npt.assert_allclose(approx_c, rms, rtol=1e-9, atol=1e-11)


def test_nd_quadrature_equals_1d_at_d1():
    """reference tests/test_multi_dim_quadrature.py:71-79."""
    N = 5
    rms = np.array([float(o.raw_moment_of_normal(0.3, 0.8, p)) for p in range(2 * N)])
    inds = md.gram_and_hankel_indices_graded_lexico(N, 1)
    w1, x1 = o.moment_quadrature(rms)
    wn, xn = md.moment_quadrature_nd(rms, inds)
    npt.assert_allclose(wn, w1, rtol=1e-12, atol=1e-14)
    npt.assert_allclose(xn[:, 0], x1, rtol=1e-12, atol=1e-14)


def _ou_setup():
    rs = np.random.RandomState(666)
    dt, T, ell, sigma = 1e-2, 60, 1., 0.5
    ts = np.linspace(dt, dt * T, T)
    cov = np.exp(-np.abs(ts[None, :] - ts[:, None]) / ell) * sigma ** 2
    ys = np.linalg.cholesky(cov) @ rs.randn(T) + rs.randn(T)
    return dt, T, ell, sigma, ys


def test_nd_filter_reduces_to_1d():
    """reference tests/test_filtering.py:304-329: the d = 1 N-D path equals the 1-D path."""
    dt, T, ell, sigma, ys = _ou_setup()
    N = 4
    b = math.sqrt(2) * sigma / math.sqrt(ell)
    mi = md.generate_graded_lexico_multi_indices(1, 2 * N - 1)
    inds = md.gram_and_hankel_indices_graded_lexico(N, 1)
    rms1, cms1, _, mean1, _ = tme_sympy.sde_cond_moments_tme_1d(lambda x: -x / ell, lambda _: b, dt, 2, 2 * N)
    rmsn, cmsn, meann, _ = tme_sympy.sde_cond_moments_tme_nd(lambda x: [-x[0] / ell], lambda x: [[b]], 1, dt, 2, mi)
    rms0 = np.array([float(o.raw_moment_of_normal(0.1, 0.3, p)) for p in range(2 * N)])
    cms0 = o.raw_to_central(rms0)

    def pdf1(y, x):
        return models.norm_pdf(y, x, 1.)

    def pdfn(y, x):
        return models.norm_pdf(y, x[0], 1.)

    r1, n1 = o.moment_filter_rms(rms1, pdf1, rms0, ys)
    rn, nn = md.moment_filter_nd_rms((rmsn, 'multi-index'), pdfn, ys, (mi, inds), rms0)
    npt.assert_allclose(rn, r1, rtol=1e-7)
    npt.assert_allclose(nn, n1, rtol=1e-10)
    c1, m1, nc1 = o.moment_filter_cms(cms1, mean1, pdf1, cms0, 0.1, ys)
    cn, mn, ncn = md.moment_filter_nd_cms((cmsn, 'multi-index'), meann, pdfn, ys, (mi, inds), cms0, np.array([0.1]))
    npt.assert_allclose(cn, c1, rtol=1e-7, atol=1e-12)
    npt.assert_allclose(mn[:, 0], m1, rtol=1e-9)
    npt.assert_allclose(ncn, nc1, rtol=1e-10)


def test_nd_modes_equivalence():
    """reference tests/test_filtering.py:169-242: nd_rms and nd_cms agree (means, nell)."""
    dt, T, ell, sigma, ys = _ou_setup()
    N, d = 3, 2
    b = math.sqrt(2) * sigma / math.sqrt(ell)
    mi = md.generate_graded_lexico_multi_indices(d, 2 * N - 1)
    inds = md.gram_and_hankel_indices_graded_lexico(N, d)
    rmsn, cmsn, meann, _ = tme_sympy.sde_cond_moments_tme_nd(
        lambda x: [-x[0] / ell, -x[1] / ell], lambda x: [[b, 0], [0, b]], d, dt, 2, mi)
    mean0, cov0 = np.array([0.1, 0.1]), 0.3 * np.eye(2)
    rms0 = np.array([md.raw_moments_mvn_kan(mean0, cov0, m) for m in mi])
    cms0 = np.array([md.raw_moments_mvn_kan(np.zeros(2), cov0, m) for m in mi])
    ys2 = np.stack([ys, ys], axis=-1)

    def pdf2(y, x):
        return float(np.prod(models.norm_pdf(y, x, 1.)))

    rn, nr = md.moment_filter_nd_rms((rmsn, 'multi-index'), pdf2, ys2, (mi, inds), rms0)
    cn, mn, nc = md.moment_filter_nd_cms((cmsn, 'multi-index'), meann, pdf2, ys2, (mi, inds), cms0, mean0)
    npt.assert_allclose(mn[:, 0], rn[:, 2], rtol=1e-8)  # graded-lex: index 2 is (1, 0), index 1 is (0, 1)
    npt.assert_allclose(mn[:, 1], rn[:, 1], rtol=1e-8)
    npt.assert_allclose(nr, nc, rtol=1e-10)


def test_independent_2d_reduces_to_two_1d_filters():
    """reference tests/test_filtering.py:244-302 restated with ITS OWN inputs and thresholds: np.random.seed(666)
    measurements (:17-31), TME-3, N = 3, m0 = 0.1, var0 = 0.2; marginals equal (default rtol 1e-7), 1-D moments
    to rtol 1e-3, and nell_2d == 2 nell_1d to the default rtol 1e-7 (:302)."""
    np.random.seed(666)
    dt, T, ell, sigma = 1e-2, 100, 1., 0.5
    ts = np.linspace(dt, dt * T, T)
    cov = np.exp(-np.abs(ts[None, :] - ts[:, None]) / ell) * sigma ** 2
    ys = np.linalg.cholesky(cov) @ np.random.randn(T) + np.random.randn(T)
    ys2 = np.stack([ys, ys], axis=-1)
    b = math.sqrt(2) * sigma / math.sqrt(ell)
    N, d, m0, var0 = 3, 2, 0.1, 0.2
    rms1, *_ = tme_sympy.sde_cond_moments_tme_1d(lambda x: -x / ell, lambda _: b, dt, 3, 2 * N)
    rms0_1 = np.array([float(o.raw_moment_of_normal(m0, var0, p)) for p in range(2 * N)])
    r1, n1 = o.moment_filter_rms(rms1, lambda y, x: models.norm_pdf(y, x, 1.), rms0_1, ys)

    mi = md.generate_graded_lexico_multi_indices(d, 2 * N - 1)
    inds = md.gram_and_hankel_indices_graded_lexico(N, d)
    rmsn, *_ = tme_sympy.sde_cond_moments_tme_nd(
        lambda x: [-x[0] / ell, -x[1] / ell], lambda x: [[b, 0], [0, b]], d, dt, 3, mi)
    rms0 = np.array([md.raw_moments_mvn_kan(m0 * np.ones(2), var0 * np.eye(2), m) for m in mi])
    rn, nn = md.moment_filter_nd_rms((rmsn, 'multi-index'),
                                     lambda y, x: float(np.prod(models.norm_pdf(y, x, 1.))), ys2, (mi, inds), rms0)
    marg0 = rn[:, md.find_indices(np.stack([np.arange(2 * N), np.zeros(2 * N, int)], -1))]
    marg1 = rn[:, md.find_indices(np.stack([np.zeros(2 * N, int), np.arange(2 * N)], -1))]
    npt.assert_allclose(rms0_1, rms0[md.find_indices(np.stack([np.arange(2 * N), np.zeros(2 * N, int)], -1))])
    npt.assert_allclose(marg0, marg1)
    npt.assert_allclose(r1, marg0, rtol=1e-3)
    npt.assert_allclose(n1 * 2, nn)


def test_nd_length_check_raises():
    mi = md.generate_graded_lexico_multi_indices(2, 3)
    inds = md.gram_and_hankel_indices_graded_lexico(2, 2)
    with pytest.raises(ValueError):
        md.moment_filter_nd_rms((None, 'index'), None, np.zeros(0), (mi, inds), np.zeros(3))
