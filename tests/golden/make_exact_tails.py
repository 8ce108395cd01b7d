"""Exact-arithmetic trajectories of the WORST replicates of the headline batch (Benes--Bernoulli N = 15, TME-3).

tools/select_tails.py (run on the GPU box) picks, per mode, the replicates of the 4096 x 1000 benchmark batch on which the
device and the C port disagree most -- largest variance / scale, NLL and mean deviation, largest first-NaN gap -- i.e. the
replicates that set `cpu_baseline.max_rel_err_vs_device` in the bench line.  This script runs oracle/exact_mp.py (the
reference's algorithm, mfs/one_dim/filtering.py:140-158 / :217-237, without rounding: 200 digits, escalated to 500 when
200 give out) on exactly those replicates, over the whole T = 1000 or up to the step where even exact arithmetic loses
positive definiteness (the TME-truncated prediction need not be a realisable moment sequence: that event belongs to the
algorithm, not to rounding, and is recorded as `exact_first_nan`).

    python tests/golden/make_exact_tails.py [--select gpurun_out/tails_select.npz] [--procs 6]

Writes tests/golden/filter_cfg2_exact_tails.npz: per mode the replicate indices, why each was selected, the measurement bits,
exact means / variances (scales) / running NLL at every step, moments at every 10th step, the first-NaN steps the two fp64
implementations had when the selection was made, and the batch maxima the selection was made from.
"""
import argparse
import math
import multiprocessing as mp_
import os
import sys
import time

for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
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
    mode, b, T, dps_list = args
    oic, ys = _G['oic'], _G['ys']
    scaled = mode == 'scaled'
    r, used, first = None, None, []
    for dps in dps_list:
        r = exact_mp.benes_bernoulli_cms(oic.scms if scaled else oic.cms, oic.mean, ys[b, :T], dps=dps, scaled=scaled,
                                         scale0=math.sqrt(oic.variance))
        used = dps
        first.append(r['first_nan'])
        if r['first_nan'] < 0:
            break
    M2 = len(oic.cms)
    f = lambda v: float(v) if v is not None else float('nan')   # noqa: E731
    moments = np.array([[f(v) for v in row] if row is not None else [float('nan')] * M2 for row in r['moments']])
    return (moments, np.array([f(v) for v in r['means']]), np.array([f(v) for v in r['scales']]),
            np.array([f(v) for v in r['nell_cum']]), r['first_nan'], used, first)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--select', type=str, default=os.path.join(ROOT, 'gpurun_out', 'tails_select.npz'))
    ap.add_argument('--procs', type=int, default=6)
    ap.add_argument('--dps', type=str, default='200,500')
    ap.add_argument('--out', type=str, default='filter_cfg2_exact_tails.npz')
    a = ap.parse_args()
    sel = np.load(a.select)
    N, T, B, seed = int(sel['N']), int(sel['T']), int(sel['batch_B']), int(sel['seed'])
    steps = sel['moment_steps']
    odt, _, oic, *_ = om.benes_bernoulli(N)
    ys = synth.benes_bernoulli_batch(B, T, odt, seed=seed)[0]
    _G.update(oic=oic, ys=ys)
    dps_list = tuple(int(v) for v in a.dps.split(','))
    modes = ('central', 'scaled')
    jobs = [(mode, int(b), T, dps_list) for mode in modes for b in sel[f'{mode}_idx']]
    t0 = time.time()
    with mp_.get_context('fork').Pool(a.procs) as pool:
        res = pool.map(_one, jobs, chunksize=1)
    out = {'N': N, 'T': T, 'batch_B': B, 'seed': seed, 'tme_order': 3, 'moment_steps': steps, 'dps': np.array(dps_list)}
    k = 0
    for mode in modes:
        idx = sel[f'{mode}_idx']
        rs = res[k:k + len(idx)]
        k += len(idx)
        out[f'{mode}_idx'] = idx
        out[f'{mode}_why'] = sel[f'{mode}_why']
        out[f'{mode}_ys_bits'] = np.packbits(ys[idx].astype(np.uint8), axis=1)
        out[f'{mode}_moments'] = np.stack([r[0][steps] for r in rs])
        out[f'{mode}_means'] = np.stack([r[1] for r in rs])
        out[f'{mode}_{"variances" if mode == "central" else "scales"}'] = \
            np.stack([r[0][:, 2] if mode == 'central' else r[2] for r in rs])
        out[f'{mode}_nell_cum'] = np.stack([r[3] for r in rs])
        out[f'{mode}_exact_first_nan'] = np.array([r[4] for r in rs], dtype=np.int32)
        out[f'{mode}_dps_used'] = np.array([r[5] for r in rs], dtype=np.int32)
        out[f'{mode}_first_nan_by_dps'] = np.array([list(r[6]) + [-2] * (len(dps_list) - len(r[6])) for r in rs], dtype=np.int32)
        # what the two fp64 implementations did when the selection was made (bench batch, round 3 start)
        out[f'{mode}_sel_dev_first'] = sel[f'{mode}_dev_first']
        out[f'{mode}_sel_c_first'] = sel[f'{mode}_c_first']
        out[f'{mode}_sel_batch_max'] = sel[f'{mode}_batch_max']      # second, mean, nll, first-NaN gap over all 4096
    path = os.path.join(HERE, a.out)
    np.savez_compressed(path, **out)
    print(f'{os.path.getsize(path) / 1024:.0f} KiB in {time.time() - t0:.0f} s')
    for mode in modes:
        print(mode, 'exact first-NaN', out[f'{mode}_exact_first_nan'].tolist(), 'dps', out[f'{mode}_dps_used'].tolist())
