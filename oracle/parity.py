"""Parity metrics shared by the GPU tests and the `cpu_baseline` leg of bench.py (TEST INFRASTRUCTURE, see
oracle/__init__.py): per-quantity relative errors between two runs of the same filter and first-NaN agreement.

Conventions
  * errors are taken over the filter-steps that are finite in BOTH runs;
  * a moment of order n is compared relative to max(|ref|, floor_n) where floor_n = 1e-3 x the largest magnitude the
    moments of orders n - 1, n and n + 1 reach over the run (odd central moments of near-symmetric laws and cms[1] are rounding
    noise around zero: a pure relative error is meaningless there) -- the same scaling the parity tests use;
  * first-NaN step = index of the first step with a non-finite output, T if the replicate survives.
"""
import numpy as np


def first_nan_steps(means, T=None):
    """(B,) first non-finite step per replicate from (B, T[, ...]) outputs; T (= no poisoning) if none."""
    means = np.asarray(means)
    bad = ~np.isfinite(means.reshape(means.shape[0], means.shape[1], -1)).all(axis=2)
    T = means.shape[1] if T is None else T
    return np.where(bad.any(axis=1), np.argmax(bad, axis=1), T)


def moment_floor(ref):
    """(2N,) per-order magnitude floor from a (..., 2N) array of reference moments."""
    flat = np.abs(np.asarray(ref).reshape(-1, ref.shape[-1]))
    colmax = np.nanmax(np.where(np.isfinite(flat), flat, 0.), axis=0)
    neighbour = np.maximum(colmax, np.maximum(np.concatenate([colmax[1:], colmax[-1:]]),
                                              np.concatenate([colmax[:1], colmax[:-1]])))
    return neighbour * 1e-3 + 1e-300


def natural_magnitude_nd(ref, multi_indices):
    """(..., z) floor for N-D central moments: 1e-2 x prod_k sd_k^{n_k} with sd_k from the reference's own second moments
    (graded-lex d = 2: (2, 0) is entry 5, (0, 2) entry 3).  First-order central moments and odd moments of near-symmetric
    laws are rounding noise around zero; a pure relative error is meaningless there."""
    ref, mi = np.asarray(ref), np.asarray(multi_indices)
    sd = np.sqrt(np.stack([np.abs(ref[..., 5]), np.abs(ref[..., 3])], axis=-1))
    return 1e-2 * np.prod(sd[..., None, :] ** mi, axis=-1) + 1e-300


def rel_err(got, ref, floor=0.):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    with np.errstate(all='ignore'):
        return np.abs(got - ref) / np.maximum(np.abs(ref), floor)


def quantity_errors(got, ref, floor=0., quantiles=(0.5, 0.99)):
    """max / quantiles of the relative error over entries finite in both; None when there is nothing to compare."""
    e = rel_err(got, ref, floor)
    ok = np.isfinite(np.asarray(got)) & np.isfinite(np.asarray(ref))
    e = e[ok]
    if e.size == 0:
        return None
    out = {'max': float(e.max()), 'n': int(e.size)}
    for q in quantiles:
        out[f'p{int(round(q * 100))}'] = float(np.quantile(e, q))
    return out


def moment_errors_by_order(got, ref, quantiles=(0.5, 0.99)):
    """Per-order max scaled error of (B, T, 2N) moments: list of 2N floats (None where nothing is comparable)."""
    floor = moment_floor(ref)
    e = rel_err(got, ref, floor)
    ok = np.isfinite(got) & np.isfinite(ref)
    e = np.where(ok, e, -1.)
    per_order = e.reshape(-1, e.shape[-1]).max(axis=0)
    return [None if v < 0 else float(v) for v in per_order]


def first_nan_agreement(fa, fb, T):
    """How two implementations agree on which step poisons a replicate (T = survives)."""
    fa, fb = np.asarray(fa), np.asarray(fb)
    d = np.abs(fa.astype(np.int64) - fb.astype(np.int64))
    both_dead = (fa < T) & (fb < T)
    out = {'replicates': int(fa.size), 'exact_match_fraction': float(np.mean(fa == fb)),
           'within_2_steps_fraction': float(np.mean(d <= 2)),
           'alive_in_both': int(np.sum((fa >= T) & (fb >= T))), 'poisoned_in_both': int(both_dead.sum()),
           'poisoned_in_first_only': int(np.sum((fa < T) & (fb >= T))),
           'poisoned_in_second_only': int(np.sum((fa >= T) & (fb < T))),
           'mean_first_nan': [float(fa.mean()), float(fb.mean())]}
    if both_dead.any():
        out['abs_step_difference_p50_p90_p99_when_both_poison'] = [float(v) for v in np.quantile(d[both_dead], [0.5, 0.9, 0.99])]
    return out


def score_against_exact_tails(fixture, mode, moments, means, second, nell, T=None):
    """Distance of one implementation from the exact-arithmetic trajectories of tests/golden/filter_cfg2_exact_tails.npz
    (the worst replicates of the headline batch, oracle/exact_mp.py at 200 / 500 digits).

    fixture  the loaded .npz;  mode 'central' | 'scaled'
    moments (R, T, 2N), means (R, T), second (R, T) = variance (central) / scale (scaled), nell (R,) of the fixture's replicates
    Compared at every step at which the implementation is finite, up to the step where exact arithmetic itself loses positive
    definiteness (`exact_first_nan`: an event of the algorithm, not of rounding).  Per replicate: the largest error of the mean
    (in standard deviations), of the variance / scale, of the moments of every order at every 10th step (scaled as
    `moment_floor`), and of the final NLL where both reach T.  Returns {'per_replicate': (R, 4) array, 'max': {...}, ...}."""
    e = fixture
    T = int(e['T']) if T is None else T
    steps = e['moment_steps']
    xf = e[f'{mode}_exact_first_nan']
    em, en, emom = e[f'{mode}_means'], e[f'{mode}_nell_cum'], e[f'{mode}_moments']
    es = e['central_variances'] if mode == 'central' else e['scaled_scales']
    sd = np.sqrt(es) if mode == 'central' else es
    floor = moment_floor(emom)
    R = len(xf)
    out = np.zeros((R, 4))
    finite_steps = np.zeros(R, dtype=np.int64)
    for b in range(R):
        hor = int(xf[b]) if xf[b] >= 0 else T
        fin = np.isfinite(means[b]) & np.isfinite(second[b]) & (np.arange(T) < hor)
        finite_steps[b] = int(fin.sum())
        if not fin.any():
            continue
        with np.errstate(all='ignore'):
            rm = np.abs(means[b] - em[b]) / np.maximum(sd[b], 1e-300)
            rs = rel_err(second[b], es[b])
            rmo = rel_err(moments[b][steps], emom[b], floor)
        out[b, 0], out[b, 1] = rm[fin].max(), rs[fin].max()
        fs = fin[steps]
        if fs.any():
            out[b, 2] = rmo[fs].max()
        if hor == T and fin[T - 1] and np.isfinite(nell[b]):
            out[b, 3] = abs(nell[b] - en[b, T - 1]) / abs(en[b, T - 1])
    names = ('mean_in_sd', 'variance' if mode == 'central' else 'scale', 'moments_all_orders', 'nll')
    return {'per_replicate': out, 'finite_steps': finite_steps, 'replicates': R,
            'max': {n: float(out[:, i].max()) for i, n in enumerate(names)},
            'replicates_over_1e-6': {n: int((out[:, i] > 1e-6).sum()) for i, n in enumerate(names)}}
