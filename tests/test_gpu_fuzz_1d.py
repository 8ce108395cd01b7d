"""Randomised GPU parity sweep of the 1-D filters against the NumPy oracle: modes x transition families x models x
N in 3..20, seeded.  Complements the hand-picked cases of test_gpu_parity_1d.py; the bar is the same 1e-6 on NLL and
means.

NaN pattern -- the criterion, decided once and independent of the seeds chosen: in exact arithmetic no replicate poisons
(tests/test_gpu_envelope.py), so a poisoning is the event "a Hankel pivot rounded to <= 0", and two fp64 implementations
may only differ on it where the matrix is numerically singular.  A replicate that poisons on ONE side only is therefore
accepted iff (i) both runs agree to 1e-6 on every step before the poisoning and (ii) the surviving side's own posterior
Hankel matrix around that step has cond >= 1e13; anything else fails.  The two known one-sided cases found by off-line
sweeps (N = 9 scaled and N = 12 raw, well--Poisson) are explicit test cases below."""
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


def _hankel_cond(ms, N):
    G = ms[np.add.outer(np.arange(N), np.arange(N))]
    ev = np.linalg.eigvalsh(G)
    return np.inf if ev.min() <= 0. else ev.max() / ev.min()


def _check(tag, N, mode, gm, gmean, gn, fn_dev, rm, rmean, rn):
    """gm / rm (B, T, 2N) moments, gmean / rmean (B, T) or None, gn / rn (B,) NLL, fn_dev (B,) device first-NaN step."""
    B, T = gm.shape[:2]
    for b in range(B):
        dev_ok, ora_ok = np.isfinite(gn[b]), np.isfinite(rn[b])
        ora_bad = ~np.isfinite(rm[b]).all(axis=1)
        f_ora = int(np.argmax(ora_bad)) if ora_bad.any() else T
        f_dev = int(fn_dev[b]) if fn_dev[b] >= 0 else T
        upto = min(f_dev, f_ora)
        if dev_ok and ora_ok:
            assert abs(gn[b] - rn[b]) / abs(rn[b]) <= 1e-6, tag
        if gmean is not None and upto > 0:
            err = np.abs(gmean[b, :upto] - rmean[b, :upto]) / np.maximum(np.abs(rmean[b, :upto]), 1e-2)
            assert err.max() <= 1e-6, f'{tag}: means differ before any poisoning'
        if dev_ok != ora_ok:
            # one-sided poisoning: only where the survivor's Hankel matrix is numerically singular around that step
            surv = rm[b] if ora_ok else gm[b]
            conds = [_hankel_cond(surv[t], N) for t in range(max(upto - 2, 0), min(upto + 1, T))]
            assert max(conds) >= 1e13, f'{tag}: replicate {b} poisons on one side only at step {upto} with cond {max(conds):.1e}'
        elif not dev_ok and abs(f_dev - f_ora) > 2:
            # both poison, at different steps: the earlier event must sit on a numerically singular matrix of the other run
            surv = rm[b] if f_ora > f_dev else gm[b]
            conds = [_hankel_cond(surv[t], N) for t in range(max(upto - 2, 0), min(upto + 1, T))]
            assert max(conds) >= 1e13, f'{tag}: replicate {b} poisons at steps {f_dev} / {f_ora}, cond {max(conds):.1e}'


def _run_case(N, mode, model, trans, T, dseed, p1, B):
    if model == 'benes':
        dt, _, _, ic, drift, disp, _, pmf, _ = ss_models.benes_bernoulli(N)
        odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
        ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=dseed)
    else:
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
        gm, gn, fn_dev = filtering.moment_filter_rms(dev[0], pmf, ic.rms, ys, return_first_nan=True)
        ref = [o.moment_filter_rms(ora[0], opmf, oic.rms, ys[b]) for b in range(B)]
        rn, gmean, rmean = np.array([r[1] for r in ref]), None, None
        rm = np.stack([r[0] for r in ref])
    elif mode == 'central':
        gm, gmean, gn, fn_dev = filtering.moment_filter_cms(dev[1], dev[3], pmf, ic.cms, ic.mean, ys, return_first_nan=True)
        ref = [o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[b]) for b in range(B)]
        rn, rmean = np.array([r[2] for r in ref]), np.stack([r[1] for r in ref])
        rm = np.stack([r[0] for r in ref])
    else:
        sc0 = np.sqrt(ic.variance)
        gm, gmean, _, gn, fn_dev = filtering.moment_filter_scms(dev[2], dev[4], pmf, ic.scms, ic.mean, sc0, ys,
                                                                return_first_nan=True)
        ref = [o.moment_filter_scms(ora[2], ora[4], opmf, oic.scms, oic.mean, sc0, ys[b]) for b in range(B)]
        rn, rmean = np.array([r[3] for r in ref]), np.stack([r[1] for r in ref])
        rm = np.stack([r[0] for r in ref])
    _check(tag, N, mode, gm, gmean, gn, fn_dev, rm, rmean, rn)


@pytest.mark.parametrize('seed', [11, 12, 13])
def test_random_cases_match_oracle(seed):
    rng = np.random.default_rng(seed)
    for _ in range(8):
        N, mode, model, trans, T, dseed = _case(rng)
        p1 = float(rng.uniform(1, 5)) if model == 'well' else None
        _run_case(N, mode, model, trans, T, dseed, p1, B=2)


@pytest.mark.parametrize('case', [
    dict(N=9, mode='scaled', model='well', trans='tme_3', T=54, dseed=94653828, p1=1.6148513167085983, B=3),
    dict(N=12, mode='raw', model='well', trans='tme_normal_2', T=21, dseed=28076647, p1=4.028574127363186, B=3),
])
def test_known_one_sided_poisonings(case):
    """Replicate 2 of each: the device's pivot rounds below zero at a step where the oracle's own posterior Hankel matrix
    has cond 4e16 (N = 9, step 46) / 1e17 (N = 12, step 8-9); means, scales and even moments agree to 1e-13 up to there."""
    _run_case(**case)


