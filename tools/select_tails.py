"""Pick the replicates of the headline batch where the device and the C port disagree most (runs on the GPU box).

The bench line's `cpu_baseline.max_rel_err_vs_device` reports maxima over 4096 x 1000 filter-steps (variance 4e-2, mean
3.5e-4, NLL 6.9e-6 in round 2) set by a handful of replicates close to losing positive definiteness.  This tool finds
them -- per mode: the replicates with the largest device-vs-C-port variance (scale) / mean / NLL deviation and those whose
first-NaN steps lie furthest apart -- and writes their indices plus both implementations' trajectories to
gpurun_out/tails_select.npz; tests/golden/make_exact_golden.py --tails then runs oracle/exact_mp.py (80+ digits) on
exactly those replicates, so that each maximum can be attributed to a side.

    python tools/select_tails.py [--per-criterion 8] [--out gpurun_out/tails_select.npz]
"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mfs_amd import synth  # noqa: E402
from mfs_amd.one_dim import filtering, moments, ss_models  # noqa: E402
from oracle import c_oracle, parity, tme_sympy, models as om  # noqa: E402

N, T, B, SEED, EVERY = 15, 1000, 4096, 100, 10


def top(score, k, taken):
    """Indices of the k largest finite scores not yet taken."""
    order = np.argsort(-np.where(np.isfinite(score), score, -np.inf))
    out = []
    for b in order:
        if len(out) == k or not np.isfinite(score[b]) or score[b] <= 0:
            break
        if int(b) not in taken:
            out.append(int(b))
            taken.add(int(b))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--per-criterion', type=int, default=8)
    ap.add_argument('--out', type=str, default=os.path.join(ROOT, 'gpurun_out', 'tails_select.npz'))
    a = ap.parse_args()
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    f = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    ys = synth.benes_bernoulli_batch(B, T, dt, seed=SEED)[0]
    odt, _, oic, odrift, odisp, _, _ = om.benes_bernoulli(N)
    tab = tme_sympy.operator_tables_1d(odrift, odisp, odt, 3, 'tanh')
    lik = np.array([0., 0., 0., 0.2])
    steps = np.arange(EVERY - 1, T, EVERY)
    out = {'N': N, 'T': T, 'batch_B': B, 'seed': SEED, 'moment_steps': steps}
    for mode in ('central', 'scaled'):
        if mode == 'central':
            dm, dmean, dnell, dfn = filtering.moment_filter_cms(f[1], f[3], pmf, ic.cms, ic.mean, ys, return_first_nan=True)
            dsec = dm[..., 2].copy()
            cm, cmean, _, cnell = c_oracle.filter_1d(1, N, ys, oic.cms, oic.mean, None, 0, 1, 6, tab, 1.0, 0, lik)
            csec = cm[..., 2].copy()
        else:
            dm, dmean, dsec, dnell, dfn = filtering.moment_filter_scms(f[2], f[4], pmf, ic.scms, ic.mean,
                                                                       math.sqrt(ic.variance), ys, return_first_nan=True)
            cm, cmean, csec, cnell = c_oracle.filter_1d(2, N, ys, oic.scms, oic.mean, math.sqrt(oic.variance), 0, 1, 6, tab,
                                                        1.0, 0, lik)
        dfirst = np.where(dfn >= 0, dfn, T)
        cfirst = parity.first_nan_steps(np.concatenate([cmean[..., None], cm], axis=-1), T)
        with np.errstate(all='ignore'):
            e_sec = parity.rel_err(dsec, csec)
            e_mean = parity.rel_err(dmean, cmean, 1e-12)
            e_nll = parity.rel_err(dnell, cnell)
        both = np.isfinite(dsec) & np.isfinite(csec)
        s_sec = np.where(both, e_sec, 0.).max(axis=1)
        s_mean = np.where(np.isfinite(dmean) & np.isfinite(cmean), e_mean, 0.).max(axis=1)
        s_nll = np.where(np.isfinite(e_nll), e_nll, 0.)
        gap = np.abs(dfirst - cfirst).astype(np.float64)
        k = a.per_criterion
        taken = set()
        sel = {'second': top(s_sec, k, taken), 'nll': top(s_nll, k, taken), 'mean': top(s_mean, max(k // 2, 1), taken),
               'gap': top(gap, 2 * k if mode == 'central' else k, taken)}
        idx = np.array(sum(sel.values(), []), dtype=np.int64)
        why = np.array(sum(([c] * len(v) for c, v in sel.items()), []))
        print(mode, 'batch maxima: second', s_sec.max(), 'mean', s_mean.max(), 'nll', s_nll.max(), 'gap', gap.max())
        for c, v in sel.items():
            for b in v:
                print(f'  {mode} {c:6s} replicate {b:4d}: second {s_sec[b]:.2e} mean {s_mean[b]:.2e} nll {s_nll[b]:.2e} '
                      f'first-NaN device {dfirst[b]} / C port {cfirst[b]}')
        out.update({f'{mode}_idx': idx, f'{mode}_why': why,
                    f'{mode}_batch_max': np.array([s_sec.max(), s_mean.max(), s_nll.max(), gap.max()]),
                    f'{mode}_dev_first': dfirst[idx], f'{mode}_c_first': cfirst[idx],
                    f'{mode}_dev_means': dmean[idx], f'{mode}_c_means': cmean[idx],
                    f'{mode}_dev_second': dsec[idx], f'{mode}_c_second': csec[idx],
                    f'{mode}_dev_nell': dnell[idx], f'{mode}_c_nell': cnell[idx],
                    f'{mode}_dev_moments': dm[idx][:, steps], f'{mode}_c_moments': cm[idx][:, steps],
                    # whole-batch per-replicate summaries: the attribution needs the runner-up maxima too
                    f'{mode}_all_s_second': s_sec, f'{mode}_all_s_mean': s_mean, f'{mode}_all_s_nll': s_nll,
                    f'{mode}_all_dev_first': dfirst, f'{mode}_all_c_first': cfirst,
                    f'{mode}_all_dev_nell': dnell, f'{mode}_all_c_nell': cnell})
        del dm, cm
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    np.savez_compressed(a.out, **out)
    print('wrote', a.out, os.path.getsize(a.out) // 1024, 'KiB')


if __name__ == '__main__':
    main()
