"""GPU parity against FROZEN oracle outputs (tests/golden/filter_cfg*.npz, written by tests/golden/make_filter_golden.py):
the HIP path on the fixture's measurements, compared with the numbers the NumPy/LAPACK restatement produced in the
build container -- nothing is recomputed by the oracle on the GPU box.  One test per BASELINE.json configuration.

Bars: 1e-6 relative on NLL / means / variances (BASELINE north_star); moments per order with the floor of
oracle/parity.py (odd central moments of near-symmetric laws are rounding noise around zero).  Where the problem is
ill-posed in fp64 (config 2: N = 15, cond(Hankel) up to 1e16) the 1e-6 bar is asserted on the replicates that survive in
both implementations, and the divergence statistics are asserted separately; tests/test_gpu_envelope.py scores both
against exact arithmetic.
"""
import math
import os

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth, stats
from mfs_amd.one_dim import filtering, moments, ss_models
from oracle import parity

pytestmark = pytest.mark.gpu
RTOL = 1e-6


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.fail(f'{name} is missing: run tests/golden/make_filter_golden.py')
    return np.load(path)


def _unpack(bits, T):
    return np.unpackbits(bits, axis=1)[:, :T].astype(np.float64)


def _scaled_moment_error(got, ref):
    return parity.rel_err(got, ref, parity.moment_floor(ref))


# ---------------------------------------------------------------------------------------------------------------------
def test_config1_benes_N7_three_modes(golden_dir):
    """BASELINE configs[0]: N = 7, T = 100, TME-3, raw / central / scaled."""
    g = _load(golden_dir, 'filter_cfg1.npz')
    N, T, B = int(g['N']), int(g['T']), int(g['B'])
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    ys = _unpack(g['ys_bits'], T)
    npt.assert_array_equal(ys, synth.benes_bernoulli_batch(B, T, dt, seed=int(g['seed']))[0])   # fixture == generator
    f = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    m, nell = filtering.moment_filter_rms(f[0], pmf, ic.rms, ys)
    npt.assert_allclose(nell, g['raw_nell'], rtol=RTOL)
    assert _scaled_moment_error(m, g['raw_moments']).max() <= RTOL
    m, means, nell = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys)
    npt.assert_allclose(nell, g['central_nell'], rtol=RTOL)
    npt.assert_allclose(means, g['central_means'], rtol=RTOL, atol=1e-9)
    assert _scaled_moment_error(m, g['central_moments']).max() <= RTOL
    m, means, scales, nell = filtering.moment_filter_scms(f[2], f[4], pmf, ic.scms, ic.mean, math.sqrt(ic.variance), ys)
    npt.assert_allclose(nell, g['scaled_nell'], rtol=RTOL)
    npt.assert_allclose(means, g['scaled_means'], rtol=RTOL, atol=1e-9)
    npt.assert_allclose(scales, g['scaled_scales'], rtol=RTOL)
    assert _scaled_moment_error(m, g['scaled_moments']).max() <= RTOL


@pytest.mark.parametrize('mode', ['central', 'scaled'])
def test_config2_headline_N15_T300_B64(golden_dir, mode):
    """BASELINE configs[1] at its own order: Benes--Bernoulli N = 15, TME-3, the first 64 replicates x 300 steps of the
    benchmark batch, against the frozen NumPy/LAPACK outputs.

    Every replicate finite in both implementations over all 300 steps: 1e-6 on NLL, mean, variance; moments within the
    per-order bound below (central mode; for the scaled mode see the comment in the body).  The poisoning statistics must
    agree.  Which of two disagreeing fp64 implementations is closer to the truth is settled in
    tests/test_gpu_envelope.py against exact arithmetic."""
    g = _load(golden_dir, 'filter_cfg2.npz')
    N, T, B = int(g['N']), int(g['T']), int(g['B'])
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    ys = _unpack(g['ys_bits'], T)
    full = synth.benes_bernoulli_batch(int(g['batch_B']), int(g['batch_T']), dt, seed=int(g['seed']))[0]
    npt.assert_array_equal(ys, full[:B, :T])
    f = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    if mode == 'central':
        m, means, nell, fn = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys, return_first_nan=True)
        second, ref_second = m[..., 2], g['central_variances']
    else:
        m, means, second, nell, fn = filtering.moment_filter_scms(f[2], f[4], pmf, ic.scms, ic.mean,
                                                                  math.sqrt(ic.variance), ys, return_first_nan=True)
        ref_second = g['scaled_scales']
    gfn = g[f'{mode}_first_nan']
    steps = g['moment_steps']
    rm, rmeans, rnell = g[f'{mode}_moments'], g[f'{mode}_means'], g[f'{mode}_nell']
    both = (fn < 0) & (gfn < 0)
    assert both.sum() >= B // 2, 'most replicates survive 300 steps in both implementations'
    if mode == 'central':
        # Where device and NumPy/LAPACK differ by more than the bar, 80-digit arithmetic decides who is off
        # (tests/golden/filter_cfg2_exact_B64.npz: the same 64 replicates): the device must be within the bar of the exact
        # result and the frozen NumPy number further from it.
        ex = _load(golden_dir, 'filter_cfg2_exact_B64.npz')
        npt.assert_array_equal(ex['ys_bits'], g['ys_bits'])

        def arbitrated(dev, ref, exact, scale=None):
            den = np.maximum(np.abs(ref), 1e-300) if scale is None else np.maximum(np.abs(ref), scale)
            err = np.abs(dev - ref) / den
            off = err > RTOL
            if off.any():
                d_ex, r_ex = np.abs(dev - exact)[off] / den[off], np.abs(ref - exact)[off] / den[off]
                assert np.all(np.isfinite(d_ex)) and np.all(d_ex <= RTOL) and np.all(r_ex > d_ex), (err[off], d_ex, r_ex)
            return int(off.sum())

        n_arb = arbitrated(nell[both], rnell[both], ex['central_nell'][both])
        print(f'central: {n_arb} NLL value(s) of {both.sum()} settled by exact arithmetic in favour of the device')
    else:
        npt.assert_allclose(nell[both], rnell[both], rtol=RTOL)
    sd = np.sqrt(g['central_variances'][both]) if mode == 'central' else ref_second[both]
    e_mean = np.abs(means[both] - rmeans[both]) / np.maximum(np.abs(rmeans[both]), sd)
    e_second = parity.rel_err(second[both], ref_second[both])
    e = _scaled_moment_error(m[both][:, steps], rm[both])
    per_order = e.reshape(-1, 2 * N).max(axis=0)
    print(f'{mode}: survivors {both.sum()}, mean {e_mean.max():.1e}, second {e_second.max():.1e}, per-order max scaled moment '
          f'error', np.array2string(per_order, precision=1))
    if mode == 'central':
        # ---- survivors in both: the north-star bar, every step (arbitrated as above where the two differ by more)
        arbitrated(means[both], rmeans[both], ex['central_means'][both], sd)
        arbitrated(second[both], ref_second[both], ex['central_variances'][both])
        # stated per-order bound: 1e-6 from order 8 up; the lowest orders carry the conditioning of the whole Hankel system
        bound = np.where(np.arange(2 * N) >= 8, 1e-6, 1e-5)
        if not np.all(per_order <= bound):
            e_ex = _scaled_moment_error(m[both][:, steps], ex['central_moments'][both])
            per_order_ex = e_ex.reshape(-1, 2 * N).max(axis=0)
            print('central: per-order max scaled moment error against exact arithmetic', np.array2string(per_order_ex, precision=1))
            assert np.all(per_order_ex <= bound), per_order_ex
    else:
        # The NumPy/LAPACK restatement of the SCALED filter is itself off by up to 4e-3 on moments and 4e-5 on scales for
        # these replicates (its distance from exact arithmetic, tests/test_gpu_envelope.py; the device's is 3e-7 / 7e-9),
        # so here the frozen numbers can only bound the bulk: medians at rounding level, 90 % within 1e-7.
        for err in (e_mean, e_second, e):
            assert np.quantile(err, 0.5) <= 1e-9 and np.quantile(err, 0.9) <= 1e-7
        assert e_mean.max() <= 1e-4 and e_second.max() <= 1e-3
    # ---- poisoning statistics (in exact arithmetic nobody poisons: which replicates do is rounding luck on both sides)
    agree = parity.first_nan_agreement(np.where(fn >= 0, fn, T), np.where(gfn >= 0, gfn, T), T)
    print(mode, agree)
    # (two CPU implementations of the same algorithm, NumPy/LAPACK and the C port, agree on the onset for 71 % of 1024
    #  replicates over 1000 steps: tests/test_gpu_envelope.py)
    assert agree['exact_match_fraction'] >= 0.70
    assert agree['poisoned_in_first_only'] <= agree['poisoned_in_second_only'] + 4   # the device is not the fragile one


@pytest.mark.parametrize('N', [5, 10, 15, 20, 25])
def test_config3_ou_convergence(golden_dir, N):
    """BASELINE configs[2] (dardel/convergence/convergence_mf.py): OU / Gaussian, central mode, T = 200, B = 4."""
    g = _load(golden_dir, 'filter_cfg3.npz')
    T, B = int(g['T']), int(g['B'])
    ys = g['ys']
    npt.assert_array_equal(ys, synth.ou_gaussian_batch(B, T, seed=int(g['seed']))[0])
    dt_ou, ell, sigma = 0.1, 1., 0.5
    F, Sigma = math.exp(-dt_ou / ell), sigma ** 2 * (1 - math.exp(-2 * dt_ou / ell))
    from mfs_amd.utils import GaussianSum1D
    ic = GaussianSum1D.new(means=[0.], variances=[sigma ** 2], weights=[1.], N=N)
    f = moments.sde_cond_moments_normal(lambda x: F * x, lambda x: Sigma)
    m, means, nell = filtering.moment_filter_cms(f[1], f[3], lambda y, x: stats.norm_pdf(y, x, 1.), ic.cms, ic.mean, ys)
    ok = np.isfinite(g[f'N{N}_nell'])
    assert ok.all() or N >= 20          # (raw mode would diverge at N >= 20; central survives these 200 steps)
    npt.assert_allclose(nell[ok], g[f'N{N}_nell'][ok], rtol=RTOL)
    npt.assert_allclose(means[ok], g[f'N{N}_means'][ok], rtol=RTOL, atol=1e-9)
    npt.assert_allclose(m[ok][..., 2], g[f'N{N}_variances'][ok], rtol=RTOL)
    # (cond(Hankel) ~ 1e16 at N = 20 and ~ 1e20 at N = 25: the highest moments carry it -- 1.3e-4 on the order-49 moment)
    assert _scaled_moment_error(m[ok][:, -1], g[f'N{N}_moments_last'][ok]).max() <= {20: 1e-5, 25: 1e-3}.get(N, RTOL)
    # and the analytic pin the reference itself uses: the exact Kalman filter's NLL (tests/test_filtering.py:82-111)
    npt.assert_allclose(nell[ok], g[f'N{N}_kf_nell'][ok], rtol={5: 1e-3, 10: 1e-5}.get(N, 1e-7))


def test_config4_well_poisson_theta_grid(golden_dir):
    """BASELINE configs[3], slice: well--Poisson N = 7, T = 1000, TME-normal-2, 32 theta points x 2 data sets, one theta
    per replicate."""
    g = _load(golden_dir, 'filter_cfg4.npz')
    N, T, keys = int(g['N']), int(g['T']), int(g['keys'])
    p1, p2 = g['p1'], g['p2']
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.well_poisson(3., N)
    ys_k = g['ys_keys'].astype(np.float64)
    npt.assert_array_equal(ys_k, synth.well_poisson_batch(keys, T, p1=3., p2=3., dt=dt, seed=int(g['seed']))[0])
    ys = np.repeat(ys_k, p1.shape[0] // keys, axis=0)
    f = moments.sde_cond_moments_tme_normal(lambda x: drift(x, p1), dispersion, dt, 2, N)
    m, means, nell, fn = filtering.moment_filter_cms(f[1], f[3], lambda y, x: pmf(y, x, p2), ic.cms, ic.mean, ys,
                                                     return_first_nan=True)
    gfn = g['central_first_nan']
    npt.assert_array_equal(fn < 0, gfn < 0)                   # N = 7: well-posed, the same replicates survive
    ok = gfn < 0
    assert ok.sum() >= 0.8 * ok.size
    steps = g['moment_steps']
    npt.assert_allclose(nell[ok], g['central_nell'][ok], rtol=RTOL)
    npt.assert_allclose(means[ok][:, steps], g['central_means'][ok], rtol=RTOL, atol=1e-9)
    assert _scaled_moment_error(m[ok][:, steps], g['central_moments'][ok]).max() <= RTOL


def test_config5_prey_predator_N6_T500(golden_dir):
    """BASELINE configs[4] at its own order and length: d = 2, N = 6 (z = 78, s = 21), T = 500, B = 4, central, TME-2
    ('multi-index'), and the TME-normal-2 closure ('index') on T = 100, B = 2.  1e-6 on NLL, means, variances and on
    EVERY moment (scaled by its natural magnitude prod_k sd_k^n_k where it is rounding noise around zero)."""
    from mfs_amd.multi_dims import filtering as fnd, moments as mnd, ss_models as snd
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    g = _load(golden_dir, 'filter_cfg5.npz')
    N, T, B = int(g['N']), int(g['T']), int(g['B'])
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, disp, _, pmf, _ = snd.prey_predator(mi)
    ys = _unpack(g['ys_bits'], T)
    npt.assert_array_equal(ys, synth.prey_predator_batch(B, T, dt, seed=int(g['seed']))[0])
    deg = mi.sum(axis=1)

    def moment_error(got, ref):
        sd = np.sqrt(np.stack([np.abs(ref[..., 5]), np.abs(ref[..., 3])], axis=-1))          # (2, 0) and (0, 2)
        natural = np.prod(sd[..., None, :] ** mi, axis=-1)
        return (np.abs(got - ref) / np.maximum(np.abs(ref), 1e-2 * natural + 1e-300))[..., deg >= 2]

    f = mnd.sde_cond_moments_tme(drift, disp, dt, 2)
    m, means, nell, fn = fnd.moment_filter_nd_cms((f[1], 'multi-index'), f[3], pmf, ys, (mi, inds), gs.cms, gs.mean,
                                                  return_first_nan=True)
    assert np.all(fn == -1)
    steps = g['moment_steps']
    npt.assert_allclose(nell, g['tme2_nell'], rtol=RTOL)
    npt.assert_allclose(means, g['tme2_means'], rtol=RTOL)
    npt.assert_allclose(m[..., 5], g['tme2_var0'], rtol=RTOL)
    npt.assert_allclose(m[..., 3], g['tme2_var1'], rtol=RTOL)
    e = moment_error(m[:, steps], g['tme2_moments'])
    print('tme_2: max scaled moment error over T = 500:', e.max())
    assert e.max() <= RTOL
    Tn, Bn = int(g['normal_T']), int(g['normal_B'])
    fnn = mnd.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi)
    m, means, nell = fnd.moment_filter_nd_cms((fnn[1], 'index'), fnn[3], pmf, ys[:Bn, :Tn], (mi, inds), gs.cms, gs.mean)
    npt.assert_allclose(nell, g['normal2_nell'], rtol=RTOL)
    npt.assert_allclose(means, g['normal2_means'], rtol=RTOL)
    assert moment_error(m[:, np.arange(9, Tn, 10)], g['normal2_moments']).max() <= RTOL
