"""CPU-only checks of the N-D host side: multi-index tables against the golden tables generated from the reference,
the N-D polynomial-ring TME generator against the oracle's SymPy differentiation, Kan moments, model tracing."""
import os

import math

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import _lib, sym, tme_poly_nd
from mfs_amd.multi_dims import filtering, moments, multi_indices as mid, ss_models
from mfs_amd.utils import GaussianSumND
from oracle import multi_dims as omd, tme_sympy


@pytest.mark.parametrize('N,d', [(3, 1), (3, 2), (4, 2), (5, 2), (6, 2), (2, 3), (3, 3)])
def test_multi_index_tables_match_reference_golden(golden_dir, N, d):
    g = np.load(os.path.join(golden_dir, 'multi_indices.npz'))
    npt.assert_array_equal(mid.generate_graded_lexico_multi_indices(d, 2 * N - 1, 0), g[f'mi_N{N}_d{d}'])
    npt.assert_array_equal(mid.gram_and_hankel_indices_graded_lexico(N, d), g[f'inds_N{N}_d{d}'])


def test_multi_index_probes_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'multi_indices.npz'))
    npt.assert_array_equal([mid.graded_lexico_indexof_multi_index(list(p)) for p in g['probe_mi']], g['probe_index'])
    npt.assert_array_equal([mid.graded_lexico_indexof_multi_index(list(p), lower_sum=2) for p in g['probe_mi'][1:]],
                           g['probe_index_lower2'])
    npt.assert_array_equal([mid.sizeof_multi_indices(*a) for a in g['size_args']], g['size_vals'])
    npt.assert_array_equal(mid.generate_graded_lexico_multi_indices(3, 4, 2), g['mi_lower_d3'])
    npt.assert_array_equal(mid.find_indices(g['mi_N4_d2']), np.arange(g['mi_N4_d2'].shape[0]))


@pytest.mark.parametrize('order', [1, 2])
def test_nd_tme_tables_match_sympy_on_prey_predator(order):
    mi = mid.generate_graded_lexico_multi_indices(2, 5)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    r, c, s, mu, mv = moments.sde_cond_moments_tme(drift, disp, dt, order)
    _, _, ogs, odrift, odisp, _, _ = omd.prey_predator(mi)
    orms, ocms, omean, omv = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, order, mi)
    x = np.array([[1.0, 1.1], [0.9, 1.2], [1.3, 0.7], [0.5, 1.6]])
    npt.assert_allclose(c(x, mi, np.array([1.01, 0.98])), ocms(x, mi, np.array([1.01, 0.98])), rtol=1e-12, atol=1e-18)
    npt.assert_allclose(r(x, mi), orms(x, mi), rtol=1e-12)
    npt.assert_allclose(mu(x), omean(x), rtol=1e-14)
    npt.assert_allclose(mv(x)[1], omv(x)[1], rtol=1e-12, atol=1e-20)
    npt.assert_allclose(gs.rms, ogs.rms, rtol=1e-13)
    npt.assert_allclose(gs.cms, ogs.cms, rtol=1e-12, atol=1e-20)


def test_nd_tracing_and_model_struct():
    mi = mid.generate_graded_lexico_multi_indices(2, 5)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    r, c, s, mu, mv = moments.sde_cond_moments_tme(drift, disp, dt, 2)
    tables = filtering._trace_transition((c, 'multi-index'), 'central')
    lik = filtering._trace_likelihood(pmf, 2)
    assert len(lik) == 1 and lik[0].kind == 'bernoulli_logistic' and (lik[0].component, lik[0].ycol) == (0, 0)
    npt.assert_allclose(lik[0].params, [-1., 0., 0., 1.])  # 1 / (1 + exp(-x^3 + 1))
    m, keep = filtering._model_struct(tables, lik)
    assert (m.d, m.extent, m.n_factors, m.ny) == (2, 5, 1, 1) and m.n_terms <= _lib.ND_TERMS
    assert (m.fac_kind[0], m.fac_component[0], m.fac_ycol[0], m.fac_n_par[0]) == (0, 0, 0, 4)
    assert (m.coef_batched, m.lik_batched) == (0, 0)
    coef = keep[0]
    # row order is the fixed graded-lex kappa list the kernel unrolls over
    for t, kap in enumerate(tables.kappas):
        row = _lib.ND_KAPPAS.index(tuple(int(v) for v in kap))
        npt.assert_array_equal(coef[row][:tables.Q[t].coef.shape[0], :tables.Q[t].coef.shape[1]], tables.Q[t].coef)
    with pytest.raises(sym.NotDeviceDescribable):  # likelihood on both components
        filtering._trace_likelihood(lambda y, x: pmf(y, [x[0] + x[1], x[1]]), 2)
    with pytest.raises(sym.NotDeviceDescribable):  # tme_4 needs |kappa| up to 8: beyond the device's tables
        r4, c4, *_ = moments.sde_cond_moments_tme(drift, disp, dt, 4)
        filtering._model_struct(filtering._trace_transition((c4, 'multi-index'), 'central'), lik)
    with pytest.raises(sym.NotDeviceDescribable):
        moments.sde_cond_moments_tme(lambda x: np.array([x[0] / x[1], x[1]], dtype=object), disp, dt, 2)


def test_kan_and_extractors():
    mean, cov = np.array([0.3, -0.2]), np.array([[0.5, 0.1], [0.1, 0.3]])
    for n in ([0, 0], [1, 0], [1, 1], [2, 0], [2, 2], [3, 1], [0, 5]):
        npt.assert_allclose(moments.raw_moments_mvn_kan(mean, cov, n), omd.raw_moments_mvn_kan(mean, cov, n), rtol=1e-13)
    npt.assert_allclose(moments.central_moments_mvn_kan(cov, [2, 2]), cov[0, 0] * cov[1, 1] + 2 * cov[0, 1] ** 2)
    assert moments.central_moments_mvn_kan(cov, [2, 1]) == 0.
    mi = mid.generate_graded_lexico_multi_indices(2, 3)
    rms = np.array([moments.raw_moments_mvn_kan(mean, cov, m) for m in mi])
    npt.assert_allclose(moments.extract_mean(rms, 2), mean)
    npt.assert_allclose(moments.extract_cov(rms, 2), cov + np.outer(mean, mean))
    npt.assert_allclose(moments.marginalise_moments(rms, 2, 2, 0),
                        [moments.raw_moments_mvn_kan(mean, cov, [p, 0]) for p in range(4)])
    gs = GaussianSumND.new(np.array([[1., 1.], [1., 1.]]), np.array([np.eye(2), 2 * np.eye(2)]) * 1e-3,
                           np.array([0.5, 0.5]), mi)
    npt.assert_allclose(gs.mean, [1., 1.])
    npt.assert_allclose(gs.cov, 1.5e-3 * np.eye(2))


@pytest.mark.parametrize('order', [1, 2, 3, 'euler'])
def test_nd_normal_closure_tables_match_kan_oracle(order):
    """mean / covariance polynomials + Stein recursion == tme.mean_and_cov + Kan's formula
    (mfs/multi_dims/moments.py:257-411, :110-154)."""
    mi = mid.generate_graded_lexico_multi_indices(2, 5)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    _, _, ogs, odrift, odisp, _, _ = omd.prey_predator(mi)
    fns = (moments.sde_cond_moments_euler_maruyama(drift, disp, dt, mi) if order == 'euler'
           else moments.sde_cond_moments_tme_normal(drift, disp, dt, order, mi))
    orms, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, order, mi)
    x = np.array([[1.0, 1.1], [0.9, 1.2], [1.3, 0.7]])
    idx = np.arange(mi.shape[0])
    c = np.array([1.01, 0.98])
    npt.assert_allclose(fns[1](x, idx, c), ocms(x, idx, c), rtol=1e-11, atol=1e-300)
    npt.assert_allclose(fns[0](x, idx[3:7]), orms(x, idx[3:7]), rtol=1e-11)
    npt.assert_allclose(fns[3](x), omean(x), rtol=1e-14)
    dense, D = fns[1].tables.dense_table()
    assert dense.shape == (5, D, D)
    # traced: an 'index' closure describes itself and is tied to its table
    tables = filtering._trace_transition((fns[1], 'index'), 'central', (mi, None))
    assert tables is fns[1].tables
    with pytest.raises(sym.NotDeviceDescribable):
        filtering._trace_transition((fns[1], 'multi-index'), 'central', (mi, None))
    with pytest.raises(ValueError):
        filtering._trace_transition((fns[1], 'index'), 'central', (mi[:-1], None))
    m, keep = filtering._model_struct(tables, filtering._trace_likelihood(pmf, 2))
    assert m.trans_kind == 1 and m.n_terms == 5


def test_product_likelihoods_vector_measurements_and_batching():
    """measurement_cond_pdf_2d of the reference (tests/test_filtering.py:44-46): prod(norm.pdf(y, x, sd)) on vector y, x
    traces to two Gaussian factors, one per component and measurement column; ys of shape (T, 2) / (B, T, 2)."""
    import math
    from mfs_amd import stats
    factors = filtering._trace_likelihood(lambda y, x: math.prod(stats.norm_pdf(y, x, 1.5)), 2)
    assert [(f.kind, f.component, f.ycol) for f in factors] == [('gaussian', 0, 0), ('gaussian', 1, 1)]
    npt.assert_allclose(factors[1].params, [1., 0., 2.25])
    # explicit columns, a second factor on the same component, per-replicate parameters
    sd = np.array([0.5, 1.0, 2.0])
    factors = filtering._trace_likelihood(
        lambda y, x: stats.norm_pdf(y[1], 2. * x[0] + 1., sd) * stats.bernoulli_pmf(y[0], 1. / (1. + sym.exp(-x[0]))), 2)
    assert [(f.kind, f.component, f.ycol) for f in factors] == [('gaussian', 0, 1), ('bernoulli_logistic', 0, 0)]
    mi = mid.generate_graded_lexico_multi_indices(2, 5)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    c = moments.sde_cond_moments_tme(drift, disp, dt, 2)[1]
    tables = filtering._trace_transition((c, 'multi-index'), 'central')
    m, (coef, lp) = filtering._model_struct(tables, factors, B=3)
    assert (m.n_factors, m.ny, m.lik_batched, m.coef_batched) == (2, 2, 1, 0) and lp.shape == (3, 2, _lib.MAX_LIK)
    npt.assert_allclose(lp[:, 0, :3], np.stack([np.full(3, 2.), np.ones(3), sd ** 2], axis=-1))
    npt.assert_allclose(lp[:, 1], np.tile([0., 1., 0., 0.], (3, 1)))
    with pytest.raises(ValueError):
        filtering._model_struct(tables, factors, B=4)
    with pytest.raises(sym.NotDeviceDescribable):   # a factor may read one component only
        filtering._trace_likelihood(lambda y, x: stats.norm_pdf(y, x[0] + x[1], 1.), 2)
    with pytest.raises(sym.NotDeviceDescribable):   # one value per component must be reduced with prod
        filtering._trace_likelihood(lambda y, x: stats.norm_pdf(y, x, 1.), 2)
    # ys layouts
    ys3, sq = filtering._split_ys(np.zeros((7, 2)), 2)
    assert ys3.shape == (1, 7, 2) and sq
    ys3, sq = filtering._split_ys(np.zeros((4, 7)), 1)
    assert ys3.shape == (4, 7, 1) and not sq
    with pytest.raises(ValueError):
        filtering._split_ys(np.zeros((7,)), 2)
    # per-replicate transition tables: B closure tuples stacked
    per = []
    for sig in (0.1, 0.2, 0.3):
        per.append(moments.sde_cond_moments_tme(drift, lambda x, s=sig: np.array([[s * x[0], 0.], [0., s * x[1]]], dtype=object), dt, 2))
    fns = moments.batch_closures(per)
    bt = filtering._trace_transition((fns[1], 'multi-index'), 'central')
    m, (coef, _) = filtering._model_struct(bt, filtering._trace_likelihood(pmf, 2), B=3)
    assert m.coef_batched == 1 and coef.shape[0] == 3 and coef.shape[1] == _lib.ND_ROWS
    _, (single, _) = filtering._model_struct(per[1][1].tables, filtering._trace_likelihood(pmf, 2))
    npt.assert_array_equal(coef[1][:, :single.shape[1], :single.shape[2]], single)


def test_d1_family_has_the_same_tables_as_the_1d_factory():
    """d = 1: the N-D closure family reduces to the 1-D tables the 1-D kernels run (reference
    tests/test_filtering.py:304-329, tests/test_one_dim_moments.py:68-88: 1-D vs N-D factories bit-equal)."""
    from mfs_amd.one_dim import moments as m1
    ell, sigma, dt = 1., 0.5, 1e-2
    nd = moments.sde_cond_moments_tme(lambda x: -x / ell, lambda _: math.sqrt(2) * sigma / math.sqrt(ell), dt, 3, d=1)
    one = m1.sde_cond_moments_tme(lambda x: -x / ell, lambda _: math.sqrt(2) * sigma / math.sqrt(ell), dt, 3)
    a, _ = nd[1].tables.as_one_dim().table()
    b, _ = one[1].tables.table()
    npt.assert_allclose(a, b, rtol=1e-15, atol=1e-300)
    mi1 = mid.generate_graded_lexico_multi_indices(1, 5)
    ndn = moments.sde_cond_moments_tme_normal(lambda x: -x / ell, lambda _: math.sqrt(2) * sigma / math.sqrt(ell), dt, 2, mi1)
    onen = m1.sde_cond_moments_tme_normal(lambda x: -x / ell, lambda _: math.sqrt(2) * sigma / math.sqrt(ell), dt, 2, 3)
    t = ndn[1].tables.as_one_dim()
    xs = np.linspace(-1., 1., 5)
    npt.assert_allclose(t.cond_mean(xs), onen[1].tables.cond_mean(xs), rtol=1e-15)
    npt.assert_allclose(t.cond_var(xs), onen[1].tables.cond_var(xs), rtol=1e-15)


def test_tme_order_3_uses_the_long_table_layout():
    """TME order 3 has derivative terms up to |kappa| = 6: 27 operator rows + 2 variance rows, extent up to 7
    (include/mfs_hip.h: MFS_ND_TABLE_ROWS); order <= 2 keeps the 16-row layout."""
    from mfs_amd import _lib
    from mfs_amd.multi_dims import filtering, moments, ss_models
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices
    mi = generate_graded_lexico_multi_indices(2, 5)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    lik = filtering._trace_likelihood(pmf, 2)
    for order, rows in ((2, _lib.ND_ROWS), (3, _lib.ND_ROWS_MAX)):
        fns = moments.sde_cond_moments_tme(drift, disp, dt, order)
        tables = filtering._trace_transition((fns[1], 'multi-index'), 'central')
        m, keep = filtering._model_struct(tables, lik)
        assert keep[0].shape[0] == rows == _lib.nd_table_rows(m.n_terms)
        assert int(tables.kappas.sum(axis=1).max()) == 2 * order
        # the variance rows are the last two of either layout
        np.testing.assert_array_equal(keep[0][rows - 2:], tables.var_blocks(m.extent))
