"""On-disk result format of the reference's drivers, so that
`reproduce_paper_plots/*.py` and the post-processing scripts consume this build's outputs unchanged.

Keys (dardel/benes_bernoulli/mf.py:83-92): raw -> rmss, nell; central -> cmss, means, nell; scaled -> scmss, means,
scales, nell.  One file per Monte-Carlo run `{mode}{_normal}_N_{N}_mc_{k}.npz` (mf.py:82)."""
import os

import numpy as np

_KEYS = {'raw': ('rmss', 'nell'), 'central': ('cmss', 'means', 'nell'), 'scaled': ('scmss', 'means', 'scales', 'nell')}


def result_filename(directory, mode, N, k, normal=False):
    return os.path.join(directory, f"{mode}{'_normal' if normal else ''}_N_{N}_mc_{k}.npz")


def save_filter_result(filename, mode, *arrays):
    """`arrays` in the order the filter returns them, for ONE replicate."""
    keys = _KEYS[mode]
    if len(arrays) != len(keys):
        raise ValueError(f'{mode} mode stores {keys}, got {len(arrays)} arrays')
    np.savez_compressed(filename, **dict(zip(keys, arrays)))


def save_batch(directory, mode, N, arrays, normal=False, first_k=0):
    """Split batched filter outputs (leading replicate axis) into the reference's one-file-per-run layout."""
    os.makedirs(directory, exist_ok=True)
    B = np.asarray(arrays[-1]).shape[0]
    for b in range(B):
        save_filter_result(result_filename(directory, mode, N, first_k + b, normal), mode, *[np.asarray(a)[b] for a in arrays])
    return B


def load_filter_result(filename, mode):
    data = np.load(filename)
    return tuple(data[k] for k in _KEYS[mode])
