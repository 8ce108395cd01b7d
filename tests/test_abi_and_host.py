"""CPU-only checks of the drop-in boundary and the host logic: the C-ABI library loads and exports every symbol
include/mfs_hip.h declares (no compute calls without a GPU), the tracer reduces reference-style callables to device
tables, and the polynomial-ring TME generator agrees with the oracle's SymPy derivation."""
import ctypes as C
import math
import os
import re

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import _lib, sym, stats
from mfs_amd.one_dim import filtering, moments, ss_models
from mfs_amd.utils import GaussianSum1D
from oracle import tme_sympy, one_dim as o, models as om

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'mfs_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mfs_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _header_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(L, name), f'{name} is declared in include/mfs_hip.h but not exported by libmfs_hip.so'
    assert sorted(_lib.DECLARED_SYMBOLS) == declared, 'ctypes signature table out of sync with the header'
    assert L.mfs_version() == _lib.ABI_VERSION == 2
    assert L.mfs_last_error() == b''


def test_model_struct_layout_matches_header():
    assert C.sizeof(_lib.MfsModel1d) == 10 * 4 + 8 + 2 * 8
    assert _lib.MfsModel1d.mean_x_coef.offset == 40 and _lib.MfsModel1d.coef.offset == 48


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmfs_hip.so')
    with pytest.raises(_lib.MfsError, match='no CPU fallback'):
        _lib.lib()


def test_argument_errors_are_codes_not_crashes():
    """Error convention: negative return code + message; mirrors the reference raising only on argument errors."""
    L = _lib.lib()
    plan = C.c_void_p()
    m = _lib.MfsModel1d()
    m.trans_kind = 7
    assert L.mfs_plan_1d_create(C.byref(plan), C.byref(m), 1, 5, 10, 4, 0, 0, 0) == -1
    assert b'trans_kind' in L.mfs_last_error()
    assert L.mfs_quadrature_1d(99, 1, None, None, None, 0, None, None, 0, None) == -2
    assert L.mfs_plan_1d_run(None, None, 0, None, None, None, None, None, None, None, None, None) == -1


# -- tracing ------------------------------------------------------------------------------------------------------------
def test_trace_benes_bernoulli_tables_match_sympy():
    N = 7
    dt, T, ts, ic, drift, dispersion, logistic, pmf, _ = ss_models.benes_bernoulli(N)
    r, c, s, mu, mv = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    tables, lik = filtering.trace_model('central', c, mu, pmf)
    assert tables.kind == 'operator' and tables.umap == 'tanh' and tables.n_terms == 6
    coef, J = tables.table()
    odt, _, _, odrift, odisp, _, _ = om.benes_bernoulli(N)
    ref = tme_sympy.operator_tables_1d(odrift, odisp, odt, 3, 'tanh')
    npt.assert_allclose(coef, ref[:, :J + 1], rtol=1e-13, atol=1e-18)
    assert np.all(np.abs(ref[:, J + 1:]) < 1e-18)
    assert lik.kind == 'bernoulli_logistic'
    npt.assert_allclose(lik.params, [0., 0., 0., 0.2])
    # the numeric side of the same closures agrees with the oracle's direct differentiation
    _, oc, osc, omu, omv = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 3, 2 * N)
    x, n = np.linspace(-2, 2, 9), np.arange(2 * N)
    npt.assert_allclose(c(x, n, 0.3), oc(x, n, 0.3), rtol=1e-12, atol=1e-15)
    npt.assert_allclose(s(x, n, 0.3, 0.7), osc(x, n, 0.3, 0.7), rtol=1e-12, atol=1e-15)
    npt.assert_allclose(mv(x)[1], omv(x)[1], rtol=1e-13)


@pytest.mark.parametrize('order', [1, 2, 3])
def test_tme_poly_cubic_drift_matches_sympy(order):
    p1 = 2.5
    r, c, s, mu, mv = moments.sde_cond_moments_tme(lambda x: x * (1 - p1 * x ** 2), lambda _: 1., 1e-2, order)
    t = c(sym.X, sym.ORDER, sym.MEAN).tables
    ref = tme_sympy.operator_tables_1d(lambda x: x * (1 - p1 * x ** 2), lambda _: 1., 1e-2, order, 'x')
    coef, J = t.table()
    npt.assert_allclose(coef, ref[:, :J + 1], rtol=1e-12, atol=1e-18)
    n1, n2 = moments.sde_cond_moments_tme_normal(lambda x: x * (1 - p1 * x ** 2), lambda _: 1., 1e-2, order, 5)[3:]
    _, _, _, omu, omv = tme_sympy.sde_cond_moments_tme_normal_1d(lambda x: x * (1 - p1 * x ** 2), lambda _: 1., 1e-2,
                                                                  order, 5)
    x = np.linspace(-1.5, 1.5, 7)
    npt.assert_allclose(n1(x), omu(x), rtol=1e-13)
    npt.assert_allclose(n2(x)[1], omv(x)[1], rtol=1e-12, atol=1e-18)


def test_state_dependent_dispersion_matches_sympy():
    """b(x) = 0.3 x (geometric noise): the generator recursion handles polynomial diffusion too."""
    a, b = (lambda x: 0.5 * x - x ** 3), (lambda x: 0.3 * x)
    t = moments.sde_cond_moments_tme(a, b, 5e-3, 2)[0](sym.X, sym.ORDER).tables
    ref = tme_sympy.operator_tables_1d(a, b, 5e-3, 2, 'x')
    coef, J = t.table()
    npt.assert_allclose(coef, ref[:, :J + 1], rtol=1e-12, atol=1e-20)


def test_trace_user_closures_with_per_replicate_parameters():
    """dardel/parameter_estimation/mf.py:41-53: closures over `drift(x, p1)` / `pmf(y, x, p2)` with arrays."""
    N = 7
    dt, T, ts, ic, drift, dispersion, emission, pmf, _ = ss_models.well_poisson(3., N)
    p1, p2 = np.array([1., 2., 3.]), np.array([0.5, 1.5, 2.5])
    _, c, _, mu, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, p1), dispersion, dt, 2, N)
    tables, lik = filtering.trace_model('central', c, mu, lambda y, x: pmf(y, x, p2))
    coef, J = tables.table(3)
    assert coef.shape[0] == 3 and tables.kind == 'gaussian' and tables.umap == 'x' and tables.mean_x_coef == 0.
    assert lik.kind == 'poisson_softplus' and lik.params.shape == (3, 1)
    m, _ = filtering.build_model_struct(tables, lik, 3)
    assert (m.coef_batched, m.lik_batched, m.n_rows) == (1, 1, 2)
    for b in range(3):
        _, _, _, omu, omv = tme_sympy.sde_cond_moments_tme_normal_1d(lambda x: x * (1 - p1[b] * x ** 2),
                                                                      lambda _: 1., dt, 2, N)
        x = np.linspace(-1, 1, 5)
        npt.assert_allclose(sym.Poly(coef[b, 0], 'x')(x), omu(x), rtol=1e-13)
        npt.assert_allclose(sym.Poly(coef[b, 1], 'x')(x), omv(x)[1], rtol=1e-12)
    with pytest.raises(ValueError, match='batched'):
        filtering.build_model_struct(tables, lik, 5)


def test_untraceable_callables_raise_not_fall_back():
    N = 3
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    r, c, s, mu, mv = moments.sde_cond_moments_tme(drift, dispersion, dt, 2)
    with pytest.raises(sym.NotDeviceDescribable):
        filtering.trace_model('raw', lambda x, n: np.ones((3, 6)), None, pmf)
    with pytest.raises(sym.NotDeviceDescribable):
        filtering.trace_model('raw', r, None, lambda y, x: 0.5)
    with pytest.raises(sym.NotDeviceDescribable):
        moments.sde_cond_moments_tme(lambda x: sym.tanh(x) * x, dispersion, dt, 2)
    with pytest.raises(TypeError):  # NumPy refuses ufuncs on the tracer; NotDeviceDescribable is a TypeError too
        moments.sde_cond_moments_tme(lambda x: np.sin(x), dispersion, dt, 2)
    with pytest.raises(sym.NotDeviceDescribable, match='disagrees'):
        filtering.trace_model('central', c, lambda x: 0.5 * x, pmf)
    # a wrapper that forwards (mean, scale) wrongly is caught
    with pytest.raises(sym.NotDeviceDescribable):
        filtering.trace_model('central', lambda x, n, m: s(x, n, 0., 1.), mu, pmf)


def test_wrappers_of_scaled_closure_are_accepted():
    """reference tests/test_filtering.py:139-143 defines raw / central closures by wrapping the scaled one."""
    r, c, s, mu, mv = moments.sde_cond_moments_tme(lambda x: -x, lambda _: 0.7, 1e-2, 2)
    t1, _ = filtering.trace_model('raw', lambda x, n: s(x, n, 0., 1.), None, lambda y, x: stats.norm_pdf(y, x, 1.))
    t2, lk = filtering.trace_model('central', lambda x, n, m: s(x, n, m, 1.), mu,
                                   lambda y, x: stats.norm_pdf(y, 2 * x + 1, 0.5))
    assert t1 is t2
    npt.assert_allclose(lk.params, [2., 1., 0.25])


def test_host_moment_utilities_match_oracle():
    rms = np.array([moments.raw_moment_of_normal(0.3, 0.8, p) for p in range(10)])
    npt.assert_allclose(rms, [float(o.raw_moment_of_normal(0.3, 0.8, p)) for p in range(10)], rtol=1e-14)
    npt.assert_allclose(moments.raw_to_central(rms), o.raw_to_central(rms), rtol=1e-12, atol=1e-12)
    npt.assert_allclose(moments.raw_to_scaled(rms), o.raw_to_scaled(rms), rtol=1e-12, atol=1e-12)
    cms = moments.raw_to_central(rms)
    npt.assert_allclose(moments.central_to_raw(cms, 0.3), rms, rtol=1e-12)
    npt.assert_allclose(moments.raw_to_central(np.stack([rms, rms]))[1], cms)
    npt.assert_allclose([moments.central_moment_of_normal(0.8, p) for p in range(8)],
                        [o.central_moment_of_normal(0.8, p) for p in range(8)], rtol=1e-14)
    a = GaussianSum1D.new([-0.5, 0.5], [0.05, 0.05], [0.5, 0.5], N=6)
    b = o.GaussianSum1D.new([-0.5, 0.5], [0.05, 0.05], [0.5, 0.5], N=6)
    for f in ('rms', 'cms', 'scms'):
        npt.assert_allclose(getattr(a, f), getattr(b, f), rtol=1e-14, atol=1e-16)
    assert a.mean == b.mean and math.isclose(a.variance, b.variance)


def test_odd_moment_count_warns_like_the_reference():
    """mfs/one_dim/filtering.py:65-66 warns and proceeds; so does the host side (the run itself needs a GPU: the device
    half of this is tests/test_gpu_parity_1d.py::test_odd_moment_count_proceeds_like_the_reference)."""
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(3)
    r, *_ = moments.sde_cond_moments_tme(drift, dispersion, dt, 2)
    tables, lik = filtering.trace_model('raw', r, None, pmf)
    with pytest.warns(UserWarning, match='not odd'):
        ys2, squeeze, B, T, ms0, N, batched, mean0, scale0 = filtering._prep_inputs(ic.rms[:5], None, None, np.zeros(4),
                                                                                    tables, lik)
    assert N == 2 and ms0.shape == (5,) and squeeze and (B, T) == (1, 4)


def test_header_constants_match_the_python_mirror():
    """The numeric #defines a binding needs (modes, table layouts, limits) have the same values in include/mfs_hip.h and in
    mfs_amd/_lib.py, and mfs_model_nd has the layout the header describes."""
    text = open(os.path.join(ROOT, 'include', 'mfs_hip.h')).read()
    defs = {k: v for k, v in re.findall(r'^#define\s+(MFS_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))\b', text, flags=re.M)}
    val = lambda name: int(defs[name], 0)   # noqa: E731
    assert val('MFS_ABI_VERSION') == _lib.ABI_VERSION
    assert val('MFS_MODE_ODD_TAIL') == _lib.MODE_ODD_TAIL
    assert (val('MFS_ND_TERMS'), val('MFS_ND_ROWS'), val('MFS_ND_TERMS_MAX'), val('MFS_ND_ROWS_MAX')) == \
           (_lib.ND_TERMS, _lib.ND_ROWS, _lib.ND_TERMS_MAX, _lib.ND_ROWS_MAX)
    assert (val('MFS_ND_MAX_EXTENT'), val('MFS_ND_MAX_EXTENT_HI'), val('MFS_ND_MAX_FACTORS')) == \
           (_lib.ND_MAX_EXTENT, _lib.ND_MAX_EXTENT_HI, _lib.ND_MAX_FACTORS)
    assert (val('MFS_ND_TRANS_OPERATOR'), val('MFS_ND_TRANS_GAUSSIAN')) == (_lib.ND_TRANS_OPERATOR, _lib.ND_TRANS_GAUSSIAN)
    assert _lib.nd_table_rows(_lib.ND_TERMS) == _lib.ND_ROWS and _lib.nd_table_rows(_lib.ND_TERMS + 1) == _lib.ND_ROWS_MAX
    assert len(_lib.ND_KAPPAS) == _lib.ND_TERMS_MAX and _lib.ND_KAPPAS[:_lib.ND_TERMS] == [(a, s - a) for s in range(1, 5) for a in range(s + 1)]
    # mfs_model_nd: 6 ints, 4 int[2] arrays, 2 ints, 2 pointers
    assert C.sizeof(_lib.MfsModelNd) == (6 + 8 + 2) * 4 + 2 * 8
