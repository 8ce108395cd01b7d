"""Exact-arithmetic trajectories of the headline filter (Benes--Bernoulli N = 15, TME-3) as a golden fixture.

oracle/exact_mp.py runs the reference's algorithm in 80-digit mpmath arithmetic, where rounding is invisible; the
result is what mfs.one_dim.filtering.moment_filter_cms / _scms compute "in exact arithmetic" on the same inputs.  Every
fp64 implementation -- XLA/LAPACK upstream, the NumPy/LAPACK oracle, the C port, the HIP kernel -- can then be scored by
its distance from that, instead of by its distance from another fp64 implementation.

    python tests/golden/make_exact_golden.py [--B 32] [--T 300] [--procs P] [--dps 80]

Writes tests/golden/filter_cfg2_exact.npz (B = 24, T = 300, both modes) and, with `--B 8 --T 1000 --modes central --every 50
--out filter_cfg2_exact_T1000.npz`, the full-length companion: for the first B replicates x first T steps of the benchmark batch (synth
seed 100, B = 4096, T = 1000), central and scaled mode: NLL, first-NaN step, means, variances / scales, moments at
every 10th step (rounded to fp64 from the 80-digit values), plus a 120-digit re-run of replicate 0 as the check that 80
digits are enough.
"""
import argparse
import math
import multiprocessing as mp_
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

from mfs_amd import synth  # noqa: E402
from oracle import exact_mp, models as om  # noqa: E402

_G = {}


def _one(args):
    mode, b, T, dps = args
    oic, ys = _G['oic'], _G['ys']
    scaled = mode == 'scaled'
    r = exact_mp.benes_bernoulli_cms(oic.scms if scaled else oic.cms, oic.mean, ys[b, :T], dps=dps, scaled=scaled,
                                     scale0=math.sqrt(oic.variance))
    M2 = len(oic.cms)
    f = lambda v: float(v) if v is not None else float('nan')   # noqa: E731
    moments = np.array([[f(v) for v in row] if row is not None else [float('nan')] * M2 for row in r['moments']])
    return (moments, np.array([f(v) for v in r['means']]), np.array([f(v) for v in r['scales']]), f(r['nell']),
            r['first_nan'])


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--B', type=int, default=32)
    ap.add_argument('--T', type=int, default=300)
    ap.add_argument('--dps', type=int, default=80)
    ap.add_argument('--procs', type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument('--modes', type=str, default='central,scaled')
    ap.add_argument('--out', type=str, default='filter_cfg2_exact.npz')
    ap.add_argument('--every', type=int, default=10, help='keep the moments of every n-th step')
    a = ap.parse_args()
    N = 15
    odt, _, oic, *_ = om.benes_bernoulli(N)
    _G.update(oic=oic, ys=synth.benes_bernoulli_batch(4096, 1000, odt, seed=100)[0])
    every = a.every
    modes = tuple(a.modes.split(','))
    out = {'N': N, 'T': a.T, 'B': a.B, 'seed': 100, 'batch_B': 4096, 'batch_T': 1000, 'tme_order': 3, 'dps': a.dps,
           'moment_steps': np.arange(every - 1, a.T, every),
           'ys_bits': np.packbits(_G['ys'][:a.B, :a.T].astype(np.uint8), axis=1)}
    t0 = time.time()
    jobs = [(mode, b, a.T, a.dps) for mode in modes for b in range(a.B)] + [(modes[0], 0, a.T, 120)]
    with mp_.get_context('fork').Pool(a.procs) as pool:
        res = pool.map(_one, jobs, chunksize=1)
    for k, mode in enumerate(modes):
        rs = res[k * a.B:(k + 1) * a.B]
        out[f'{mode}_moments'] = np.stack([r[0][every - 1::every] for r in rs])
        out[f'{mode}_means'] = np.stack([r[1] for r in rs])
        out[f'{mode}_{"variances" if mode == "central" else "scales"}'] = \
            np.stack([r[0][:, 2] if mode == 'central' else r[2] for r in rs])
        out[f'{mode}_nell'] = np.array([r[3] for r in rs])
        out[f'{mode}_first_nan'] = np.array([r[4] for r in rs], dtype=np.int32)
    hi = res[-1]
    out['check120_nell'] = hi[3]
    out['check120_means'] = hi[1]
    out['check120_moments'] = hi[0][every - 1::every]
    path = os.path.join(HERE, a.out)
    np.savez_compressed(path, **out)
    print(f'{os.path.getsize(path) / 1024:.0f} KiB in {time.time() - t0:.0f} s; 80 vs 120 digits on replicate 0: '
          f'nell {abs(out[modes[0] + "_nell"][0] - hi[3])}, means {np.nanmax(np.abs(out[modes[0] + "_means"][0] - hi[1]))}')
