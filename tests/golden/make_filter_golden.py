"""Freeze oracle outputs for BASELINE.json's configurations as golden fixtures (inputs + expected outputs).

The reference itself cannot run here (JAX / tme absent), so these vectors are produced by the
NumPy/LAPACK restatement under oracle/ -- pinned by the reference's analytic tests, tests/test_oracle_*.py -- on seeded
synthetic measurements, and committed so that the GPU parity tests compare the HIP path with FROZEN numbers instead
of recomputing the oracle on the GPU box.  Run in the build container:

    python tests/golden/make_filter_golden.py [cfg1 cfg2 cfg2stable cfg2env cfg3 cfg4 cfg5 ...] [--procs P]

Files (tests/golden/):
  filter_cfg1.npz     Benes--Bernoulli N = 7, T = 100, B = 3, raw / central / scaled, TME-3 (BASELINE configs[0])
  filter_cfg2.npz     Benes--Bernoulli N = 15, TME-3, central and scaled: the first 64 replicates x first 300 steps of
                      the benchmark batch (synth seed 100, B = 4096, T = 1000): NLL, means, variances / scales for
                      every step, all 2N moments at every 10th step, first-NaN step (BASELINE configs[1], slice)
  filter_cfg2stable.npz  the same 64 replicates x 300 steps with stable=True (LDL^T completion), central
  filter_cfg2env.npz  the same batch, first 1024 replicates x all 1000 steps, central: NLL, first-NaN step, means and
                      variances at every 100th step -- the NumPy/LAPACK leg of the three-way envelope test
  filter_cfg3.npz     OU / Gaussian convergence model N in {5, 10, 15, 20, 25}, T = 200, B = 4, central (configs[2])
  filter_cfg4.npz     well--Poisson N = 7, T = 1000, TME-normal-2, 32 theta points x 2 data sets (configs[3], slice)
  filter_cfg5.npz     prey--predator d = 2, N = 6, central, TME-2 ('multi-index'), T = 500, B = 4; and TME-normal-2
                      ('index'), T = 100, B = 2 (configs[4], slice)
Measurements are stored bit-packed (Bernoulli) or as small integers next to the outputs, so a fixture is
self-contained; the tests also regenerate them from mfs_amd.synth and check that the two agree.
"""
import argparse
import math
import multiprocessing as mp
import os
import sys
import time

for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):   # forked workers + threaded BLAS deadlock
    os.environ.setdefault(_v, '1')

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mfs_amd import synth  # noqa: E402  (seeded input generators only)
from oracle import one_dim as o, models as om, multi_dims as omd, tme_sympy  # noqa: E402

_G = {}   # closures shared with forked workers (lambdified functions do not pickle)


def first_nan_of(means):
    """Index of the first step whose outputs are non-finite, -1 if none (the device's out_first_nan)."""
    bad = ~np.isfinite(np.asarray(means).reshape(means.shape[0], -1)).all(axis=1)
    return int(np.argmax(bad)) if bad.any() else -1


def _pool_map(fn, items, procs):
    if procs <= 1:
        return [fn(i) for i in items]
    with mp.get_context('fork').Pool(procs) as pool:
        return pool.map(fn, items, chunksize=1)


# ---------------------------------------------------------------------------------------------------------------------
def cfg1(procs):
    N, T, B = 7, 100, 3
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    ora = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 3, 2 * N)
    ys, _ = synth.benes_bernoulli_batch(B, T, odt, seed=7)
    s0 = math.sqrt(oic.variance)
    out = {'N': N, 'T': T, 'B': B, 'seed': 7, 'tme_order': 3, 'ys_bits': np.packbits(ys.astype(np.uint8), axis=1)}
    raw = [o.moment_filter_rms(ora[0], opmf, oic.rms, y) for y in ys]
    cen = [o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, y) for y in ys]
    sca = [o.moment_filter_scms(ora[2], ora[4], opmf, oic.scms, oic.mean, s0, y) for y in ys]
    out.update(raw_moments=np.stack([r[0] for r in raw]), raw_nell=np.array([r[1] for r in raw]),
               central_moments=np.stack([r[0] for r in cen]), central_means=np.stack([r[1] for r in cen]),
               central_nell=np.array([r[2] for r in cen]),
               scaled_moments=np.stack([r[0] for r in sca]), scaled_means=np.stack([r[1] for r in sca]),
               scaled_scales=np.stack([r[2] for r in sca]), scaled_nell=np.array([r[3] for r in sca]))
    return out


def _cfg2_one(args):
    mode, b, T = args
    ora, opmf, oic, ys = _G['ora'], _G['opmf'], _G['oic'], _G['ys']
    if mode == 'central':
        m, means, nell = o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[b, :T])
        return m, means, m[:, 2].copy(), nell
    m, means, scales, nell = o.moment_filter_scms(ora[2], ora[4], opmf, oic.scms, oic.mean, math.sqrt(oic.variance),
                                                  ys[b, :T])
    return m, means, scales, nell


def _cfg2_setup():
    N = 15
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    _G.update(ora=tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 3, 2 * N), opmf=opmf, oic=oic,
              ys=synth.benes_bernoulli_batch(4096, 1000, odt, seed=100)[0])
    return N


def cfg2(procs):
    N = _cfg2_setup()
    B, T, every = 64, 300, 10
    out = {'N': N, 'T': T, 'B': B, 'seed': 100, 'batch_B': 4096, 'batch_T': 1000, 'tme_order': 3,
           'moment_steps': np.arange(every - 1, T, every), 'ys_bits': np.packbits(_G['ys'][:B, :T].astype(np.uint8), axis=1)}
    for mode in ('central', 'scaled'):
        res = _pool_map(_cfg2_one, [(mode, b, T) for b in range(B)], procs)
        out[f'{mode}_moments'] = np.stack([r[0][every - 1::every] for r in res])
        out[f'{mode}_means'] = np.stack([r[1] for r in res])
        out[f'{mode}_{"variances" if mode == "central" else "scales"}'] = np.stack([r[2] for r in res])
        out[f'{mode}_nell'] = np.array([r[3] for r in res])
        out[f'{mode}_first_nan'] = np.array([first_nan_of(np.column_stack([r[1], r[0]])) for r in res], dtype=np.int32)
    return out


def _cfg2_stable_one(args):
    b, T = args
    ora, opmf, oic, ys = _G['ora'], _G['opmf'], _G['oic'], _G['ys']
    # which rule (2 per step) is the first whose LDL^T has a pivot that is not > 0: from there on the run is no longer the
    # plain filter (the reference's explicit LDL^T loop rounds differently from LAPACK's potrf, so this is not always the
    # step at which the plain oracle run poisons)
    calls, first = [0], [-1]
    ldl0 = o.ldl

    def spy(mat):
        l, d = ldl0(mat)
        if first[0] < 0 and not np.all(d > 0):
            first[0] = calls[0]
        calls[0] += 1
        return l, d
    o.ldl = spy
    try:
        m, means, nell = o.moment_filter_cms(ora[1], ora[3], opmf, oic.cms, oic.mean, ys[b, :T], stable=True)
    finally:
        o.ldl = ldl0
    return m, means, m[:, 2].copy(), nell, (first[0] // 2 if first[0] >= 0 else -1)


def cfg2stable(procs):
    """The 64 replicates of filter_cfg2.npz with stable=True (LDL^T completion, mfs/utils.py:525-538), central mode."""
    N = _cfg2_setup()
    B, T, every = 64, 300, 10
    res = _pool_map(_cfg2_stable_one, [(b, T) for b in range(B)], procs)
    return {'N': N, 'T': T, 'B': B, 'seed': 100, 'batch_B': 4096, 'batch_T': 1000, 'tme_order': 3,
            'moment_steps': np.arange(every - 1, T, every), 'ys_bits': np.packbits(_G['ys'][:B, :T].astype(np.uint8), axis=1),
            'central_moments': np.stack([r[0][every - 1::every] for r in res]),
            'central_means': np.stack([r[1] for r in res]), 'central_variances': np.stack([r[2] for r in res]),
            'central_nell': np.array([r[3] for r in res]),
            'central_first_completion': np.array([r[4] for r in res], dtype=np.int32),   # step of the first completed rule, -1: none
            'central_first_nan': np.array([first_nan_of(np.column_stack([r[1], r[0]])) for r in res], dtype=np.int32)}


def cfg2env(procs):
    N = _cfg2_setup()
    B, T, every = 1024, 1000, 100
    res = _pool_map(_cfg2_one, [('central', b, T) for b in range(B)], procs)
    return {'N': N, 'T': T, 'B': B, 'seed': 100, 'batch_B': 4096, 'batch_T': 1000, 'tme_order': 3,
            'ys_bits': np.packbits(_G['ys'][:B].astype(np.uint8), axis=1), 'check_steps': np.arange(every - 1, T, every),
            'central_means': np.stack([r[1][every - 1::every] for r in res]),
            'central_variances': np.stack([r[2][every - 1::every] for r in res]),
            'central_nell': np.array([r[3] for r in res]),
            'central_first_nan': np.array([first_nan_of(np.column_stack([r[1], r[0]])) for r in res], dtype=np.int32)}


def cfg3(procs):
    T, B = 200, 4
    ys, _ = synth.ou_gaussian_batch(B, T, seed=3)
    out = {'T': T, 'B': B, 'seed': 3, 'ys': ys, 'Ns': np.array([5, 10, 15, 20, 25])}
    for N in out['Ns']:
        m = om.ou_gaussian(int(N))
        res = [o.moment_filter_cms(m['cond_cms'], m['cond_mean'], m['pdf'], m['cms0'], m['mean0'], y) for y in ys]
        out[f'N{N}_moments_last'] = np.stack([r[0][-1] for r in res])
        out[f'N{N}_means'] = np.stack([r[1] for r in res])
        out[f'N{N}_variances'] = np.stack([r[0][:, 2] for r in res])
        out[f'N{N}_nell'] = np.array([r[2] for r in res])
        kf = [m['kf'](y) for y in ys]
        out[f'N{N}_kf_nell'] = np.array([k[2] for k in kf])
    return out


def _cfg4_one(b):
    ys, p1, p2, N, dt, oic, drift, disp, pmf = (_G[k] for k in ('ys', 'p1', 'p2', 'N', 'dt', 'oic', 'drift', 'disp', 'pmf'))
    fns = tme_sympy.sde_cond_moments_tme_normal_1d(lambda x: drift(x, float(p1[b])), disp, dt, 2, N)
    m, means, nell = o.moment_filter_cms(fns[1], fns[3], lambda y, x: pmf(y, x, float(p2[b])), oic.cms, oic.mean, ys[b])
    return m, means, nell


def cfg4(procs):
    N, T, keys = 7, 1000, 2
    dt, _, oic, drift, disp, _, pmf = om.well_poisson(N)
    g1, g2 = np.meshgrid(np.linspace(0.5, 6., 8), np.linspace(0.5, 6., 4), indexing='ij')
    p1, p2 = np.tile(g1.ravel(), keys), np.tile(g2.ravel(), keys)
    ys_k, _ = synth.well_poisson_batch(keys, T, p1=3., p2=3., dt=dt, seed=5)
    ys = np.repeat(ys_k, g1.size, axis=0)
    _G.update(ys=ys, p1=p1, p2=p2, N=N, dt=dt, oic=oic, drift=drift, disp=disp, pmf=pmf)
    res = _pool_map(_cfg4_one, list(range(p1.shape[0])), procs)
    every = 10
    return {'N': N, 'T': T, 'keys': keys, 'seed': 5, 'p1': p1, 'p2': p2, 'ys_keys': ys_k.astype(np.int16),
            'moment_steps': np.arange(every - 1, T, every),
            'central_moments': np.stack([r[0][every - 1::every] for r in res]),
            'central_means': np.stack([r[1][every - 1::every] for r in res]),
            'central_nell': np.array([r[2] for r in res]),
            'central_first_nan': np.array([first_nan_of(np.column_stack([r[1], r[0]])) for r in res], dtype=np.int32)}


def _cfg5_one(args):
    which, b, T = args
    mi, inds, gs, opmf, ys = (_G[k] for k in ('mi', 'inds', 'gs', 'opmf', 'ys'))
    fn, sig, mean_fn = _G[which]
    return omd.moment_filter_nd_cms((fn, sig), mean_fn, opmf, ys[b, :T], (mi, inds), gs.cms, gs.mean)


def cfg5(procs):
    N, T, B = 6, 500, 4
    mi = omd.generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = omd.gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, gs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    _, ocms, omean, _ = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, 2, mi)
    _, ncms, nmean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, 2, mi)
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=21)
    _G.update(mi=mi, inds=inds, gs=gs, opmf=opmf, ys=ys, tme=(ocms, 'multi-index', omean), normal=(ncms, 'index', nmean))
    every = 10
    res = _pool_map(_cfg5_one, [('tme', b, T) for b in range(B)], procs)
    Tn, Bn = 100, 2
    resn = _pool_map(_cfg5_one, [('normal', b, Tn) for b in range(Bn)], procs)
    return {'N': N, 'T': T, 'B': B, 'seed': 21, 'ys_bits': np.packbits(ys.astype(np.uint8), axis=1),
            'moment_steps': np.arange(every - 1, T, every),
            'tme2_moments': np.stack([r[0][every - 1::every] for r in res]),
            'tme2_means': np.stack([r[1] for r in res]), 'tme2_nell': np.array([r[2] for r in res]),
            'tme2_var0': np.stack([r[0][:, 5] for r in res]), 'tme2_var1': np.stack([r[0][:, 3] for r in res]),
            'normal_T': Tn, 'normal_B': Bn, 'normal2_moments': np.stack([r[0][every - 1::every] for r in resn]),
            'normal2_means': np.stack([r[1] for r in resn]), 'normal2_nell': np.array([r[2] for r in resn])}


CONFIGS = {'cfg1': cfg1, 'cfg2': cfg2, 'cfg2stable': cfg2stable, 'cfg2env': cfg2env, 'cfg3': cfg3, 'cfg4': cfg4, 'cfg5': cfg5}

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('configs', nargs='*', default=list(CONFIGS))
    ap.add_argument('--procs', type=int, default=max(1, (os.cpu_count() or 2) - 1))
    a = ap.parse_args()
    for name in a.configs:
        t0 = time.time()
        out = CONFIGS[name](a.procs)
        path = os.path.join(HERE, f'filter_{name}.npz')
        np.savez_compressed(path, **out)
        print(f'{name}: {os.path.getsize(path) / 1024:.0f} KiB in {time.time() - t0:.0f} s', flush=True)
