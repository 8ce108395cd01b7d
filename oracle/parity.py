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
