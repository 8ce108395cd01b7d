"""The headline configuration (Benes--Bernoulli N = 15, TME-3) scored against EXACT arithmetic, and the three-way
envelope device / NumPy-LAPACK oracle / C port at the benchmark's own size.

Why this file exists.  At N = 15 the Hankel systems reach cond ~ 1e12 ... 1e16: two correct fp64 implementations of
mfs/one_dim/filtering.py:140-158 part by far more than 1e-16 and disagree on when a replicate NaN-poisons, so "equal to
the oracle" cannot be the whole bar there.  The arbiter is tests/golden/filter_cfg2_exact.npz: the reference's algorithm
run in 80-digit arithmetic (oracle/exact_mp.py), in which NO replicate poisons -- poisoning is a rounding artefact of
fp64, upstream included.  The tests assert

  1. the device is within 1e-6 of the exact trajectories on NLL, means, variances / scales and every moment, wherever
     it is finite (north-star tolerance, against the truth rather than against another fp64 implementation);
  2. the device is no further from the truth than the NumPy/LAPACK restatement (what XLA's LAPACK calls do upstream)
     and the C port, and poisons no more replicates than they do;
  3. at the benchmark size (first 1024 replicates x 1000 steps of the 4096-replicate batch) the device sits inside the
     envelope of the two CPU implementations: its first-NaN agreement and survivors' NLL differences against either are
     no worse than theirs against each other.
"""
import math
import os

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import synth
from mfs_amd.one_dim import filtering, moments, ss_models
from oracle import c_oracle, parity, tme_sympy, models as om

pytestmark = pytest.mark.gpu


def _load(golden_dir, name):
    path = os.path.join(golden_dir, name)
    if not os.path.exists(path):
        pytest.fail(f'{name} is missing: see tests/golden/make_exact_golden.py / make_filter_golden.py')
    return np.load(path)


def _setup():
    N = 15
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    f = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    ys = synth.benes_bernoulli_batch(4096, 1000, dt, seed=100)[0]
    odt, _, oic, odrift, odisp, _, _ = om.benes_bernoulli(N)
    tab = tme_sympy.operator_tables_1d(odrift, odisp, odt, 3, 'tanh')      # SymPy derivation: independent of the product's tables
    return N, ic, f, pmf, ys, oic, tab


def _device(mode, ic, f, pmf, ys):
    if mode == 'central':
        m, means, nell, fn = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys, return_first_nan=True)
        return m, means, m[..., 2], nell, fn
    m, means, scales, nell, fn = filtering.moment_filter_scms(f[2], f[4], pmf, ic.scms, ic.mean, math.sqrt(ic.variance), ys,
                                                              return_first_nan=True)
    return m, means, scales, nell, fn


def _cport(mode, N, oic, tab, ys):
    lik = np.array([0., 0., 0., 0.2])
    if mode == 'central':
        cm, cmeans, _, cnell = c_oracle.filter_1d(1, N, ys, oic.cms, oic.mean, None, 0, 1, 6, tab, 1.0, 0, lik)
        return cm, cmeans, cm[..., 2], cnell
    cm, cmeans, cscales, cnell = c_oracle.filter_1d(2, N, ys, oic.scms, oic.mean, math.sqrt(oic.variance), 0, 1, 6, tab, 1.0,
                                                    0, lik)
    return cm, cmeans, cscales, cnell


def _errors(mom_at_steps, means, second, nell, e, mode):
    sec = 'variances' if mode == 'central' else 'scales'
    sd = np.sqrt(e['central_variances']) if mode == 'central' else e['scaled_scales']
    with np.errstate(all='ignore'):
        out = {'nll': parity.rel_err(nell, e[f'{mode}_nell']),
               'mean': np.abs(means - e[f'{mode}_means']) / np.maximum(np.abs(e[f'{mode}_means']), sd),
               'second': parity.rel_err(second, e[f'{mode}_{sec}']),
               'moments': parity.rel_err(mom_at_steps, e[f'{mode}_moments'], parity.moment_floor(e[f'{mode}_moments']))}
    return {k: v[np.isfinite(v)] for k, v in out.items()}


@pytest.mark.parametrize('mode', ['central', 'scaled'])
def test_device_against_exact_arithmetic(golden_dir, mode):
    e = _load(golden_dir, 'filter_cfg2_exact.npz')
    g = _load(golden_dir, 'filter_cfg2.npz')
    N, ic, f, pmf, ys_full, oic, tab = _setup()
    T, B = int(e['T']), int(e['B'])
    ys = ys_full[:B, :T]
    npt.assert_array_equal(np.packbits(ys.astype(np.uint8), axis=1), e['ys_bits'])
    assert np.all(e[f'{mode}_first_nan'] == -1), 'in exact arithmetic no replicate poisons'
    # 80 digits are enough: the 120-digit re-run of replicate 0 rounds to the same doubles
    if mode == 'central':
        assert e['check120_nell'] == e['central_nell'][0]
        npt.assert_array_equal(e['check120_means'], e['central_means'][0])
    steps = e['moment_steps']
    sec = 'variances' if mode == 'central' else 'scales'
    m, means, second, nell, fn = _device(mode, ic, f, pmf, ys)
    dev = _errors(m[:, steps], means, second, nell, e, mode)
    cm, cmeans, csecond, cnell = _cport(mode, N, oic, tab, ys)
    cpo = _errors(cm[:, steps], cmeans, csecond, cnell, e, mode)
    ora = _errors(g[f'{mode}_moments'][:B], g[f'{mode}_means'][:B], g[f'{mode}_{sec}'][:B], g[f'{mode}_nell'][:B], e, mode)
    for name, err in (('device', dev), ('numpy/lapack', ora), ('c port', cpo)):
        print(f'{mode:8s} {name:13s}' + '  '.join(f'{k}: max {v.max():.1e} p99 {np.quantile(v, .99):.1e} p50 {np.quantile(v, .5):.1e}'
                                                 for k, v in err.items()))
    # 1. north-star tolerance against the truth, on every finite step of every replicate
    for k in ('nll', 'mean', 'second', 'moments'):
        assert dev[k].max() <= 1e-6, (k, dev[k].max())
    # 2. no further from the truth than the CPU implementations (quantiles; an absolute floor well under the bar keeps
    #    the comparison meaningful where everybody is at rounding level)
    for k in ('nll', 'mean', 'second', 'moments'):
        for qq in (0.5, 0.99):
            best_cpu = min(np.quantile(ora[k], qq), np.quantile(cpo[k], qq))
            assert np.quantile(dev[k], qq) <= 3. * best_cpu + 1e-10, (k, qq)
        assert dev[k].max() <= 3. * max(ora[k].max(), cpo[k].max()) + 1e-10
    survivors = {'device': int((fn < 0).sum()), 'numpy/lapack': int((g[f'{mode}_first_nan'][:B] < 0).sum()),
                 'c port': int(np.isfinite(cnell).sum())}
    print(mode, 'survivors of', B, survivors)
    assert survivors['device'] >= min(survivors['numpy/lapack'], survivors['c port']) - 1


def test_device_against_exact_arithmetic_full_length(golden_dir):
    """The same at the benchmark's own length T = 1000 (8 replicates, central mode)."""
    path = os.path.join(golden_dir, 'filter_cfg2_exact_T1000.npz')
    if not os.path.exists(path):
        pytest.skip('filter_cfg2_exact_T1000.npz not generated yet (tests/golden/make_exact_golden.py --T 1000 ...)')
    e = np.load(path)
    N, ic, f, pmf, ys_full, oic, tab = _setup()
    T, B = int(e['T']), int(e['B'])
    ys = ys_full[:B, :T]
    npt.assert_array_equal(np.packbits(ys.astype(np.uint8), axis=1), e['ys_bits'])
    # (over 1000 steps cond(Hankel) outgrows even 80 digits for some replicates -- 1e80 where fp64 gave up at 1e16 hundreds
    #  of steps earlier; the arbiter's own horizon is recorded in the fixture and every fp64 run ends well before it)
    horizon = np.where(e['central_first_nan'] >= 0, e['central_first_nan'], T)
    steps = e['moment_steps']
    m, means, second, nell, fn = _device('central', ic, f, pmf, ys)
    assert np.all(np.where(fn >= 0, fn, T) <= horizon)
    dev = _errors(m[:, steps], means, second, nell, e, 'central')
    cm, cmeans, csecond, cnell = _cport('central', N, oic, tab, ys)
    cpo = _errors(cm[:, steps], cmeans, csecond, cnell, e, 'central')
    for name, err in (('device', dev), ('c port', cpo)):
        print(f'T=1000 {name:8s}' + '  '.join(f'{k}: max {v.max():.1e} p99 {np.quantile(v, .99):.1e}' for k, v in err.items() if v.size))
    print('first non-finite step: device', np.where(fn >= 0, fn, T).tolist(), ' c port',
          parity.first_nan_steps(cmeans[..., None], T).tolist())
    for k in ('mean', 'second', 'moments'):
        assert dev[k].max() <= 1e-6, (k, dev[k].max())
    if dev['nll'].size:
        assert dev['nll'].max() <= 1e-6
    assert (fn < 0).sum() >= np.isfinite(cnell).sum() - 1


def test_three_way_envelope_at_benchmark_size(golden_dir):
    """Device, NumPy/LAPACK (frozen: tests/golden/filter_cfg2env.npz) and the C port on the same 1024 replicates x 1000
    steps of the benchmark batch.  Pairwise: which replicates survive, where the others poison, NLL of common survivors."""
    g = _load(golden_dir, 'filter_cfg2env.npz')
    N, ic, f, pmf, ys_full, oic, tab = _setup()
    T, B = int(g['T']), int(g['B'])
    ys = ys_full[:B, :T]
    npt.assert_array_equal(np.packbits(ys.astype(np.uint8), axis=1), g['ys_bits'])
    m, means, second, nell, fn = _device('central', ic, f, pmf, ys)
    _, cmeans, _, cnell = c_oracle.filter_1d(1, N, ys, oic.cms, oic.mean, None, 0, 1, 6, tab, 1.0, 0,
                                             np.array([0., 0., 0., 0.2]), want_moments=False)
    first = {'device': np.where(fn >= 0, fn, T), 'numpy': np.where(g['central_first_nan'] >= 0, g['central_first_nan'], T),
             'cport': parity.first_nan_steps(cmeans[..., None], T)}
    nll = {'device': nell, 'numpy': g['central_nell'], 'cport': cnell}
    stats = {}
    for a, b in (('device', 'numpy'), ('device', 'cport'), ('numpy', 'cport')):
        ag = parity.first_nan_agreement(first[a], first[b], T)
        both = np.isfinite(nll[a]) & np.isfinite(nll[b])
        rel = parity.rel_err(nll[a][both], nll[b][both])
        stats[a, b] = dict(ag, nll_p50=float(np.quantile(rel, .5)), nll_p90=float(np.quantile(rel, .9)),
                           nll_p99=float(np.quantile(rel, .99)), nll_max=float(rel.max()), common_survivors=int(both.sum()))
        print(f'{a} vs {b}: exact first-NaN match {ag["exact_match_fraction"]:.3f}, alive in both {ag["alive_in_both"]}, '
              f'only first poisons {ag["poisoned_in_first_only"]}, only second {ag["poisoned_in_second_only"]}; NLL of common '
              f'survivors p50 {stats[a, b]["nll_p50"]:.1e} p90 {stats[a, b]["nll_p90"]:.1e} p99 {stats[a, b]["nll_p99"]:.1e} '
              f'max {stats[a, b]["nll_max"]:.1e}')
    cpu = stats['numpy', 'cport']
    for other in ('numpy', 'cport'):
        d = stats['device', other]
        # the device agrees with either CPU implementation at least as well as they agree with each other
        assert d['exact_match_fraction'] >= cpu['exact_match_fraction'] - 0.05
        assert d['nll_p50'] <= 3. * cpu['nll_p50'] + 1e-12 and d['nll_p90'] <= 3. * cpu['nll_p90'] + 1e-11
        assert d['nll_p99'] <= 5. * cpu['nll_p99'] + 1e-9
        # and is not the fragile one: it does not poison replicates the other keeps more often than the reverse, by much
        assert d['poisoned_in_first_only'] <= d['poisoned_in_second_only'] + 0.03 * B
    alive = {k: int((v >= T).sum()) for k, v in first.items()}
    print('survivors at T = 1000 of', B, alive)
    assert alive['device'] >= min(alive['numpy'], alive['cport']) - 0.02 * B
    # frozen NumPy/LAPACK means / variances at every 100th step, for the replicates alive in both
    both = (first['device'] >= T) & (first['numpy'] >= T)
    st = g['check_steps']
    sd = np.sqrt(g['central_variances'][both])
    err = np.abs(means[both][:, st] - g['central_means'][both]) / np.maximum(np.abs(g['central_means'][both]), sd)
    print('common survivors', int(both.sum()), 'mean error vs numpy: max %.1e p99 %.1e' % (err.max(), np.quantile(err, .99)))
    assert np.quantile(err, .99) <= 1e-6


def test_predict_rule_from_posterior_atoms_vs_recomputed(golden_dir, monkeypatch):
    """The posterior of an update is an N-atom measure, and the N-node Gauss rule of an N-atom measure is that measure: the
    kernel takes the next predict-half rule from the atoms it already holds (the Cholesky of the posterior moments still
    decides the poisoning) instead of reconstructing it by the eigen-decomposition of quadtures.py:128-133.
    MFS_PREDICT_RULE=recompute restores the reconstruction.  Checked here: (i) where the reconstruction is well
    conditioned (N = 7, 10) the two routes give the same filter to 1e-9; (ii) at N = 15, where they differ by the
    conditioning of the Hankel matrix, BOTH stay within the same bounds against 80-digit arithmetic (measured: NLL 3.2e-11 /
    3.2e-11, mean 1.2e-9 / 2.2e-9, variance 7.8e-9 / 1.8e-8, moments 1.0e-7 / 6.5e-8 for atoms / recomputed, with 17 / 15 of the 24
    replicates surviving 300 steps; in exact arithmetic all do)."""
    # (i)
    for N, T in ((7, 100), (10, 100)):
        dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
        f = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
        ys = synth.benes_bernoulli_batch(16, T, dt, seed=5 + N)[0]
        a = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys)
        monkeypatch.setenv('MFS_PREDICT_RULE', 'recompute')
        r = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys)
        monkeypatch.delenv('MFS_PREDICT_RULE')
        assert np.all(np.isfinite(a[2])) and np.all(np.isfinite(r[2]))
        npt.assert_allclose(a[2], r[2], rtol=1e-9)
        npt.assert_allclose(a[1], r[1], rtol=1e-9, atol=1e-12)
        err = parity.rel_err(a[0], r[0], parity.moment_floor(r[0]))
        assert np.nanmax(err) <= (1e-9 if N == 7 else 1e-7), np.nanmax(err)
    # (ii)
    e = _load(golden_dir, 'filter_cfg2_exact.npz')
    N, ic, f, pmf, ys_full, oic, tab = _setup()
    T, B = int(e['T']), int(e['B'])
    ys, steps = ys_full[:B, :T], e['moment_steps']
    worst = {}
    for route in ('atoms', 'recompute'):
        if route == 'recompute':
            monkeypatch.setenv('MFS_PREDICT_RULE', 'recompute')
        m, means, second, nell, fn = _device('central', ic, f, pmf, ys)
        if route == 'recompute':
            monkeypatch.delenv('MFS_PREDICT_RULE')
        survivors = int((fn < 0).sum())                     # (in exact arithmetic all 24 survive; in fp64 about two thirds)
        assert survivors >= B // 2
        err = _errors(m[:, steps], means, second, nell, e, 'central')
        worst[route] = {k: float(v.max()) for k, v in err.items()}
        worst[route]['survivors'] = survivors
        assert worst[route]['nll'] <= 1e-9 and worst[route]['mean'] <= 1e-7 and worst[route]['second'] <= 1e-6
        assert worst[route]['moments'] <= 1e-6
    print(worst)
    assert worst['atoms']['survivors'] >= worst['recompute']['survivors'] - 1     # fewer fp64 casualties, if anything
