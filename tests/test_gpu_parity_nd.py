"""GPU parity of the N-D (d = 2) path against the oracle (BASELINE config 5 in miniature).

Individual nodes / weights of the N-D rule are not unique when the K_k have repeated eigenvalues (SURVEY section 7,
hard part 4), so only filter outputs -- moments, means, NLL -- are compared."""
import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth, sym
from mfs_amd.multi_dims import filtering, moments, ss_models
from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, gram_and_hankel_indices_graded_lexico
from oracle import multi_dims as omd, tme_sympy

pytestmark = pytest.mark.gpu


def _assert_moments(got, ref, mi, rtol):
    """Relative error per moment with the natural magnitude prod_k sigma_k^{n_k} as the floor of the denominator:
    first-order central moments and odd moments of near-symmetric laws are rounding noise around zero."""
    got, ref, mi = np.asarray(got), np.asarray(ref), np.asarray(mi)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    d = mi.shape[1]
    second = [int(np.where((mi == 2 * np.eye(d, dtype=int)[k]).all(axis=1))[0][0]) for k in range(d)]
    m2 = np.stack([np.abs(ref[:, second[k]]) for k in range(d)], axis=-1)          # (T, d): E[(x_k - c_k)^2] or E[x_k^2]
    natural = np.prod(np.sqrt(m2)[:, None, :] ** mi[None, :, :], axis=-1)          # (T, z)
    scale = np.maximum(np.abs(ref), natural * 1e-2 + 1e-300)
    err = np.abs(got - ref) / scale
    assert np.nanmax(err) <= rtol, f'max scaled error {np.nanmax(err):.3e} > {rtol}'


def _setup(N, tme_order=2):
    d = 2
    mi = generate_graded_lexico_multi_indices(d, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, d)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    fns = moments.sde_cond_moments_tme(drift, disp, dt, tme_order)
    _, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    ofns = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, tme_order, mi)
    return mi, inds, dt, gs, fns, pmf, ogs, ofns, opmf


@pytest.mark.parametrize('N,T,tme_order', [(2, 80, 2), (3, 60, 2), (4, 40, 1), (6, 12, 2), (3, 40, 3), (4, 20, 3)])
def test_prey_predator_central_and_raw(N, T, tme_order):
    mi, inds, dt, gs, fns, pmf, ogs, ofns, opmf = _setup(N, tme_order)
    B = 3
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=N)
    cmss, means, nell = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms,
                                                       gs.mean)
    rmss, nell_r = filtering.moment_filter_nd_rms((fns[0], 'multi-index'), pmf, ys, (mi, inds), gs.rms)
    assert cmss.shape == (B, T, mi.shape[0]) and means.shape == (B, T, 2) and nell.shape == (B,)
    for b in range(B if N < 6 else 1):
        rc = omd.moment_filter_nd_cms((ofns[1], 'multi-index'), ofns[2], opmf, ys[b], (mi, inds), ogs.cms, ogs.mean)
        npt.assert_allclose(nell[b], rc[2], rtol=1e-6)
        npt.assert_allclose(means[b], rc[1], rtol=1e-6)
        _assert_moments(cmss[b], rc[0], mi, rtol=1e-6)
        if N <= 4:
            rr = omd.moment_filter_nd_rms((ofns[0], 'multi-index'), opmf, ys[b], (mi, inds), ogs.rms)
            npt.assert_allclose(nell_r[b], rr[1], rtol=1e-6)
            npt.assert_allclose(rmss[b], rr[0], rtol=1e-6)
    # raw and central filters agree with each other on the device (reference tests/test_filtering.py:229-242)
    # (raw monomial Gram matrices around (1, 1) with variance 1e-3 are numerically singular at N = 6: raw mode
    #  NaN-poisons there in any fp64 implementation, so the cross-check stops at N = 4)
    if N <= 4:
        npt.assert_allclose(means[:, :, 0], rmss[:, :, 2], rtol=1e-6)
        npt.assert_allclose(means[:, :, 1], rmss[:, :, 1], rtol=1e-6)
    else:
        assert np.all(np.isfinite(nell))


def test_nd_shapes_errors_and_single_trajectory():
    mi, inds, dt, gs, fns, pmf, *_ = _setup(3)
    ys, _ = synth.prey_predator_batch(2, 20, dt, seed=1)
    m1, means1, nell1 = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys[1], (mi, inds), gs.cms,
                                                       gs.mean)
    mB, meansB, nellB = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms,
                                                       gs.mean)
    npt.assert_array_equal(m1, mB[1])
    assert np.ndim(nell1) == 0 and nell1 == nellB[1]
    with pytest.raises(ValueError, match='must match'):  # the reference's only raise, multi_dims/filtering.py:238
        filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms[:-1], gs.mean)
    with pytest.raises(sym.NotDeviceDescribable):
        filtering.moment_filter_nd_cms((fns[1], 'index'), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
    # NaN poisoning is per replicate
    cms0 = np.tile(gs.cms, (2, 1))
    cms0[0, 5] = -1.  # E[(x0 - m0)^2] < 0
    m, means, nell, fn = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), cms0,
                                                        np.tile(gs.mean, (2, 1)), return_first_nan=True)
    assert fn[0] == 0 and np.isnan(nell[0]) and np.all(np.isnan(m[0]))
    assert fn[1] == -1 and nell[1] == nellB[1]


# (grid sizes of the Chebyshev-grid prediction, NCP = g (2N - 1) + 1: 7, 11, 6, 22, 19, 12, 28 -- and 34 > 28 at N = 6 with
#  TME-normal-3, which takes the eigen-node route)
@pytest.mark.parametrize('N,T,order', [(2, 60, 2), (3, 40, 2), (3, 40, 'euler'), (4, 25, 3), (5, 10, 2), (6, 8, 'euler'), (5, 6, 3),
                                       (6, 5, 3)])
def test_prey_predator_normal_closures(N, T, order):
    """'index'-signature Normal closures (mfs/multi_dims/moments.py:257-411; used by dardel/prey_predator/mf.py with
    --trans=tme_normal_2 / euler; reference tests/test_filtering.py:182-222) against the oracle's Kan-formula path."""
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    _, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    if order == 'euler':
        fns = moments.sde_cond_moments_euler_maruyama(drift, disp, dt, mi)
    else:
        fns = moments.sde_cond_moments_tme_normal(drift, disp, dt, order, mi)
    orms, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, order, mi)
    B = 2
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=10 + N)
    cmss, means, nell = filtering.moment_filter_nd_cms((fns[1], 'index'), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
    rc = omd.moment_filter_nd_cms((ocms, 'index'), omean, opmf, ys[0], (mi, inds), ogs.cms, ogs.mean)
    npt.assert_allclose(nell[0], rc[2], rtol=1e-6)
    npt.assert_allclose(means[0], rc[1], rtol=1e-6)
    _assert_moments(cmss[0], rc[0], mi, rtol=1e-6)   # (N = 6 with TME-normal-3 needs the 34 x 34 Chebyshev grid: the largest the tables hold)
    if N <= 3:
        rmss, nell_r = filtering.moment_filter_nd_rms((fns[0], 'index'), pmf, ys, (mi, inds), gs.rms)
        rr = omd.moment_filter_nd_rms((orms, 'index'), opmf, ys[0], (mi, inds), ogs.rms)
        npt.assert_allclose(nell_r[0], rr[1], rtol=1e-6)
        npt.assert_allclose(rmss[0], rr[0], rtol=1e-6)
    # a closure declared with the wrong signature, or built for another table, is refused
    with pytest.raises(sym.NotDeviceDescribable):
        filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)


@pytest.mark.parametrize('N,T,family', [(3, 50, 'tme_2'), (4, 30, 'tme_2'), (3, 50, 'tme_normal_2'), (6, 10, 'tme_2')])
def test_prey_predator_scaled_mode(N, T, family):
    """moment_filter_nd_scms (mfs/multi_dims/filtering.py:33-207) against the oracle, and the reference's own
    three-mode equivalence (tests/test_filtering.py:168-242: means of the scaled and central filters agree)."""
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    _, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    if family == 'tme_2':
        fns, sig = moments.sde_cond_moments_tme(drift, disp, dt, 2), 'multi-index'
        _, ocms, _, omean_var = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, 2, mi)
    else:
        fns, sig = moments.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi), 'index'
        _, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, 2, mi)
        tab = fns[4].tables

        def omean_var(x):
            return omean(x), tab.cond_var(x)   # diag of the same polynomial covariance (checked on the CPU suite)

    def oscms(x, idx, mean, scale):
        sel = mi[np.asarray(idx)] if sig == 'index' else np.asarray(idx)
        return ocms(x, idx, mean) / np.prod(np.asarray(scale) ** sel, axis=-1)

    scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))           # (2, 0) and (0, 2)
    scms0 = gs.cms / np.prod(scale0 ** mi, axis=-1)
    B = 2
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=20 + N)
    scmss, means, scales, nell = filtering.moment_filter_nd_scms((fns[2], sig), fns[4], pmf, ys, (mi, inds), scms0,
                                                                 gs.mean, scale0)
    assert scmss.shape == (B, T, mi.shape[0]) and means.shape == (B, T, 2) and scales.shape == (B, T, 2)
    cmss, means_c, nell_c = filtering.moment_filter_nd_cms((fns[1], sig), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
    npt.assert_allclose(means, means_c, rtol=1e-7 if N <= 4 else 1e-5)
    npt.assert_allclose(nell, nell_c, rtol=1e-7 if N <= 4 else 1e-5)
    npt.assert_allclose(scales ** 2, np.stack([cmss[:, :, 5], cmss[:, :, 3]], axis=-1), rtol=1e-6)
    npt.assert_allclose(scmss[:, :, 5], 1., rtol=1e-9)           # posterior scaled second moments are 1 by construction
    rs = omd.moment_filter_nd_scms((oscms, sig), omean_var, opmf, ys[0], (mi, inds), scms0, ogs.mean, scale0)
    npt.assert_allclose(nell[0], rs[3], rtol=1e-6)
    npt.assert_allclose(means[0], rs[1], rtol=1e-6)
    npt.assert_allclose(scales[0], rs[2], rtol=1e-6)
    _assert_moments(scmss[0], rs[0], mi, rtol=1e-6)
    with pytest.raises(sym.NotDeviceDescribable):   # the mean-only closure is not the mean-and-variance closure
        filtering.moment_filter_nd_scms((fns[2], sig), fns[3], pmf, ys, (mi, inds), scms0, gs.mean, scale0)


@pytest.mark.parametrize('N,T,family', [(4, 40, 'tme_2'), (6, 30, 'tme_2'), (4, 30, 'tme_normal_2')])
def test_update_by_eigen_decomposition_matches_chebyshev_route(N, T, family, monkeypatch):
    """The update half evaluates lik_k(X_k) e_0 by a checked Chebyshev interpolant and falls back to diagonalising K_k
    (quadratures.py:163 as the reference does it) when the coefficient tail has not converged.  MFS_ND_UPDATE=eigen
    makes the fallback the only route: both must give the reference's numbers."""
    mi, inds, dt, gs, fns, pmf, ogs, ofns, opmf = _setup(N, 2)
    if family == 'tme_normal_2':
        _, _, _, _, drift, disp, _, _, _ = ss_models.prey_predator(mi)
        nf = moments.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi)
        trans, mean_fn = (nf[1], 'index'), nf[3]
    else:
        trans, mean_fn = (fns[1], 'multi-index'), fns[3]
    B = 2
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=40 + N)
    cheb = filtering.moment_filter_nd_cms(trans, mean_fn, pmf, ys, (mi, inds), gs.cms, gs.mean)
    monkeypatch.setenv('MFS_ND_UPDATE', 'eigen')
    eig = filtering.moment_filter_nd_cms(trans, mean_fn, pmf, ys, (mi, inds), gs.cms, gs.mean)
    monkeypatch.delenv('MFS_ND_UPDATE')
    assert np.all(np.isfinite(cheb[2])) and np.all(np.isfinite(eig[2]))
    npt.assert_allclose(eig[2], cheb[2], rtol=1e-9)
    npt.assert_allclose(eig[1], cheb[1], rtol=1e-9)
    for b in range(B):
        _assert_moments(eig[0][b], cheb[0][b], mi, rtol=1e-6)
    if family == 'tme_2':
        rc = omd.moment_filter_nd_cms((ofns[1], 'multi-index'), ofns[2], opmf, ys[0], (mi, inds), ogs.cms, ogs.mean)
        npt.assert_allclose(eig[2][0], rc[2], rtol=1e-6)
        npt.assert_allclose(eig[1][0], rc[1], rtol=1e-6)
        _assert_moments(eig[0][0], rc[0], mi, rtol=1e-6)


def test_tme_order_3_scaled_mode_and_wide_table():
    """TME order 3 on the operator path: derivative terms up to |kappa| = 6 (27 rows instead of 14) and coefficient blocks of
    extent 7 -- the long table layout and its kernel instantiation: the central filter against the oracle, and the scaled filter
    against the central one (reference tests/test_filtering.py:169-242 restated at order 3)."""
    N, T, B = 3, 30, 2
    mi, inds, dt, gs, fns, pmf, ogs, ofns, opmf = _setup(N, 3)
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=33)
    tables = filtering._trace_transition((fns[1], 'multi-index'), 'central')
    assert int(tables.kappas.sum(axis=1).max()) == 6 and tables.dense_table()[1] == 7
    scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))
    scms0 = gs.cms / np.prod(scale0 ** mi, axis=-1)
    scmss, means_s, scales, nell_s = filtering.moment_filter_nd_scms((fns[2], 'multi-index'), fns[4], pmf, ys, (mi, inds),
                                                                     scms0, gs.mean, scale0)
    cmss, means_c, nell_c = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms,
                                                           gs.mean)
    rc = omd.moment_filter_nd_cms((ofns[1], 'multi-index'), ofns[2], opmf, ys[0], (mi, inds), ogs.cms, ogs.mean)
    npt.assert_allclose(nell_c[0], rc[2], rtol=1e-6)
    npt.assert_allclose(means_c[0], rc[1], rtol=1e-6)
    _assert_moments(cmss[0], rc[0], mi, rtol=1e-6)
    npt.assert_allclose(nell_s, nell_c, rtol=1e-6)
    npt.assert_allclose(means_s, means_c, rtol=1e-6)
    npt.assert_allclose(scales ** 2, np.stack([cmss[:, :, 5], cmss[:, :, 3]], axis=-1), rtol=1e-6)


@pytest.mark.parametrize('family', ['tme_2', 'tme_normal_2'])
def test_maximum_order_N7(family):
    """The largest order the d = 2 kernels are built for: N = 7 (s = 28 Gram size, z = 105 moments, 784 tensor nodes) -- its
    own code paths: two elimination streams in the front end (3 s > 64 lanes), strided Jacobi rounds without the index table,
    gather indices from global memory.  Against the oracle, a few steps."""
    N, T, B = 7, 6, 2
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    assert mi.shape[0] == 105 and inds.shape[1] == 28
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    _, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=70)
    if family == 'tme_2':
        fns, sig = moments.sde_cond_moments_tme(drift, disp, dt, 2), 'multi-index'
        ofns = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, 2, mi)
        ocms, omean = ofns[1], ofns[2]
    else:
        fns, sig = moments.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi), 'index'
        _, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, 2, mi)
    cmss, means, nell = filtering.moment_filter_nd_cms((fns[1], sig), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
    assert cmss.shape == (B, T, 105) and np.all(np.isfinite(nell))
    rc = omd.moment_filter_nd_cms((ocms, sig), omean, opmf, ys[0], (mi, inds), ogs.cms, ogs.mean)
    npt.assert_allclose(nell[0], rc[2], rtol=1e-6)
    npt.assert_allclose(means[0], rc[1], rtol=1e-6)
    _assert_moments(cmss[0], rc[0], mi, rtol=1e-6)


def test_nd_empty_batch_and_zero_steps():
    """Empty inputs behave like the 1-D entry points': B = 0 gives empty outputs, T = 0 gives the initial state's shapes and a
    zero NLL (a scan over no steps)."""
    mi, inds, dt, gs, fns, pmf, *_ = _setup(3)
    ys, _ = synth.prey_predator_batch(3, 10, dt, seed=2)
    m, means, nell = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys[:0], (mi, inds), gs.cms, gs.mean)
    assert m.shape == (0, 10, mi.shape[0]) and means.shape == (0, 10, 2) and nell.shape == (0,)
    m, means, nell = filtering.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys[:, :0], (mi, inds), gs.cms, gs.mean)
    assert m.shape == (3, 0, mi.shape[0]) and means.shape == (3, 0, 2)
    npt.assert_array_equal(nell, np.zeros(3))


def test_normal_closure_poisoning_is_per_replicate():
    """An indefinite start (negative variance) NaN-poisons that replicate at step 0 on the Normal-closure path too (its
    prediction runs on the Chebyshev grid: the rule's matrices are NaN, so are the grid weights), the neighbour is untouched."""
    N = 4
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    fns = moments.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi)
    ys, _ = synth.prey_predator_batch(2, 20, dt, seed=1)
    cms0 = np.tile(gs.cms, (2, 1))
    cms0[0, 5] = -1.
    m, means, nell, fn = filtering.moment_filter_nd_cms((fns[1], 'index'), fns[3], pmf, ys, (mi, inds), cms0,
                                                        np.tile(gs.mean, (2, 1)), return_first_nan=True)
    assert list(fn) == [0, -1] and np.isnan(nell[0]) and np.all(np.isnan(m[0])) and np.all(np.isfinite(m[1]))
    alone = filtering.moment_filter_nd_cms((fns[1], 'index'), fns[3], pmf, ys[1], (mi, inds), gs.cms, gs.mean)
    npt.assert_array_equal(m[1], alone[0])
