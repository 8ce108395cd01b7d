"""Graded-lexicographic multi-indices and the Gram / Hankel gather tables, mirroring `mfs.multi_dims.multi_indices`
(same function names and results; host side, NumPy, static index tables computed once per run).

Order (mfs/multi_dims/multi_indices.py:61-112): by total degree first, then lexicographically on the tuple, e.g. for
d = 2: (0,0), (0,1), (1,0), (0,2), (1,1), (2,0), ...  The position of a multi-index is therefore
#{|x| < |n|} + its rank among the degree-|n| tuples, which `graded_lexico_indexof_multi_index` evaluates in closed
form with stars-and-bars counts.
"""
import itertools
import math
from typing import Sequence

import numpy as np

__all__ = ['sizeof_multi_indices', 'graded_lexico_indexof_multi_index', 'generate_graded_lexico_multi_indices',
           'find_indices', 'gram_and_hankel_indices_graded_lexico']


def sizeof_multi_indices(d: int, upper_sum: int, lower_sum: int = 0) -> int:
    """#{x in N^d : lower_sum <= |x| <= upper_sum} (mfs/multi_dims/multi_indices.py:25-59)."""
    if upper_sum < lower_sum:
        return 0
    below = math.comb(lower_sum - 1 + d, d) if lower_sum > 0 else 0
    return math.comb(upper_sum + d, d) - below


def graded_lexico_indexof_multi_index(multi_index: Sequence[int], lower_sum: int = 0) -> int:
    """0-based position of `multi_index` among {lower_sum <= |x|} in graded-lex order (:61-112)."""
    mi = [int(v) for v in multi_index]
    d, total = len(mi), sum(mi)
    pos = math.comb(total - 1 + d, d) if total > 0 else 0          # everything of smaller degree
    remaining = total
    for i, v in enumerate(mi[:-1]):
        # tuples of the same degree that agree on the first i entries and have a smaller entry i
        for smaller in range(v):
            pos += math.comb(remaining - smaller + (d - i - 2), d - i - 2)
        remaining -= v
    if lower_sum > 0:
        pos -= math.comb(lower_sum - 1 + d, d)
    return pos


def generate_graded_lexico_multi_indices(d: int, upper_sum: int, lower_sum: int = 0) -> np.ndarray:
    """(z, d) int64 table of all multi-indices with lower_sum <= |x| <= upper_sum, graded-lex ordered (:139-177)."""
    rows = []
    for s in range(lower_sum, upper_sum + 1):
        rows.extend(sorted(t for t in itertools.product(range(s + 1), repeat=d) if sum(t) == s))
    return np.asarray(rows, dtype='int64').reshape(len(rows), d)


def find_indices(multi_indices) -> np.ndarray:
    """Positions of an array (..., d) of multi-indices (:180-182)."""
    mi = np.asarray(multi_indices)
    flat = mi.reshape(-1, mi.shape[-1])
    return np.array([graded_lexico_indexof_multi_index(r) for r in flat], dtype='int64').reshape(mi.shape[:-1])


def gram_and_hankel_indices_graded_lexico(N: int, d: int) -> np.ndarray:
    """(d + 1, s, s) gather tables, s = C(N - 1 + d, d): G = ms[inds[0]], H_k = ms[inds[1 + k]] (:185-229)."""
    basis = generate_graded_lexico_multi_indices(d, N - 1, 0)
    s = basis.shape[0]
    inds = np.zeros((d + 1, s, s), dtype='int64')
    sums = basis[:, None, :] + basis[None, :, :]
    inds[0] = find_indices(sums)
    for k in range(d):
        shifted = sums.copy()
        shifted[:, :, k] += 1
        inds[k + 1] = find_indices(shifted)
    return inds
