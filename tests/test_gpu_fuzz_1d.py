"""Randomised GPU parity sweep of the 1-D filters against the NumPy oracle: modes x transition families x models x
N in 3..20, seeded.  Complements the hand-picked cases of test_gpu_parity_1d.py; the bar is the same 1e-6 on NLL and
means, and the NaN pattern (which replicates poison) must agree wherever the problem is well-posed."""
import numpy as np
import pytest

from mfs_amd import synth
from mfs_amd.one_dim import filtering, moments, ss_models
from oracle import one_dim as o, models as om, tme_sympy

pytestmark = pytest.mark.gpu


def _case(rng):
    N = int(rng.choice([3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15, 16, 18, 20]))
    mode = str(rng.choice(['raw', 'central', 'scaled']))
    model = str(rng.choice(['benes', 'well']))
    trans = str(rng.choice(['tme_2', 'tme_3', 'tme_normal_2', 'tme_normal_3', 'euler']))
    if model == 'benes' and trans == 'euler':
        trans = 'tme_3'
    T = int(rng.integers(20, 50)) if N <= 10 else int(rng.integers(10, 25))
    return N, mode, model, trans, T, int(rng.integers(1 << 30))


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_cases_match_oracle(seed):
    rng = np.random.default_rng(seed)
    B = 2
    for _ in range(8):
        N, mode, model, trans, T, dseed = _case(rng)
        if model == 'benes':
            dt, _, _, ic, drift, disp, _, pmf, _ = ss_models.benes_bernoulli(N)
            odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
            ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=dseed)
        else:
            p1 = float(rng.uniform(1, 5))
            dt, _, _, ic, drift0, disp, _, pmf0, _ = ss_models.well_poisson(p1, N)
            odt, _, oic, odrift0, odisp, _, opmf0 = om.well_poisson(N)
            drift, pmf = (lambda x, p=p1: drift0(x, p)), (lambda y, x: pmf0(y, x, 3.))
            odrift, opmf = (lambda x, p=p1: odrift0(x, p)), (lambda y, x: opmf0(y, x, 3.))
            ys, _ = synth.well_poisson_batch(B, T, p1=p1, p2=3., dt=dt, seed=dseed)
        if trans == 'euler':
            dev = moments.sde_cond_moments_euler(drift, disp, dt, N)
            ora = tme_sympy.sde_cond_moments_euler_1d(odrift, odisp, odt, N)
        elif trans.startswith('tme_normal'):
            order = int(trans[-1])
            dev = moments.sde_cond_moments_tme_normal(drift, disp, dt, order, N)
            ora = tme_sympy.sde_cond_moments_tme_normal_1d(odrift, odisp, odt, order, N)
        else:
            order = int(trans[-1])
            dev = moments.sde_cond_moments_tme(drift, disp, dt, order)
            ora = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, order, 2 * N)
        tag = f'N={N} {mode} {model} {trans} T={T}'
        if mode == 'raw':
            _, gn = filtering.moment_filter_rms(dev[0], pmf, ic.rms, ys)
            ref = [o.moment_filter_rms(ora[0], opmf, oic.rms, ys[b]) for b in range(B)]
            rn, gmean, rmean = np.array([r[1] for r in ref]), None, None
        elif mode == 'central':
            _, gmean, gn = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys)
            ref = [o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[b]) for b in range(B)]
            rn, rmean = np.array([r[2] for r in ref]), np.stack([r[1] for r in ref])
        else:
            sc0 = np.sqrt(ic.variance)
            _, gmean, _, gn = filtering.moment_filter_scms(dev[2], dev[4], pmf, ic.scms, ic.mean, sc0, ys)
            ref = [o.moment_filter_scms(ora[2], ora[4], opmf, oic.scms, oic.mean, sc0, ys[b]) for b in range(B)]
            rn, rmean = np.array([r[3] for r in ref]), np.stack([r[1] for r in ref])
        # which replicate loses positive definiteness, and when, is decided by the last bits once the Hankel matrix
        # reaches cond ~ 1e16 (N >= 14 without scaling): there two correct fp64 implementations already disagree
        # (DESIGN.md section 4), so the pattern is only required to match below that
        if N <= 12:
            assert np.array_equal(np.isfinite(gn), np.isfinite(rn)), f'{tag}: NaN pattern differs'
        ok = np.isfinite(rn) & np.isfinite(gn)
        if ok.any():
            assert np.max(np.abs(gn[ok] - rn[ok]) / np.abs(rn[ok])) <= 1e-6, tag
            if gmean is not None:
                err = np.abs(gmean[ok] - rmean[ok]) / np.maximum(np.abs(rmean[ok]), 1e-2)
                assert np.nanmax(err) <= 1e-6, tag
