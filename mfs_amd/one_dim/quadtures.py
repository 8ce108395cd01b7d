"""Moment quadrature on the device, mirroring `mfs.one_dim.quadtures.moment_quadrature`
(mfs/one_dim/quadtures.py:83-133): Hankel Cholesky, two triangular solves, symmetric eigensolve."""
import numpy as np

from mfs_amd import _lib

__all__ = ['moment_quadrature', 'hankel_indices']


def hankel_indices(n: int):
    """G_idx[i, j] = i + j, H_idx = G_idx + 1 (mfs/one_dim/quadtures.py:29-60)."""
    inds = np.arange(n)[:, None] + np.arange(n)[None, :]
    return inds, inds + 1


def moment_quadrature(ms, mean=0., scale=1., sort_nodes: bool = False, ldl: bool = False, *, device: int = 0):
    """Weights and nodes from 2N moments; `ms` is (2N,) or (B, 2N), `mean` / `scale` scalars or (B,)."""
    ms = np.asarray(ms, dtype=np.float64)
    squeeze = ms.ndim == 1
    ms2 = np.ascontiguousarray(ms[None, :] if squeeze else ms)
    B, M2 = ms2.shape
    N = M2 // 2
    if M2 % 2 or not 2 <= N <= _lib.MAX_N:
        raise ValueError(f'need 2N moments with 2 <= N <= {_lib.MAX_N}, got {M2}')
    mean_a = np.ascontiguousarray(np.broadcast_to(np.asarray(mean, dtype=np.float64), (B,)))
    scale_a = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (B,)))
    w, x = np.empty((B, N)), np.empty((B, N))
    _lib.check(_lib.lib().mfs_quadrature_1d(N, B, _lib.ptr(ms2), _lib.ptr(mean_a), _lib.ptr(scale_a), int(bool(ldl)),
                                            _lib.ptr(w), _lib.ptr(x), device, None))
    if sort_nodes:
        order = np.argsort(x, axis=-1)
        w, x = np.take_along_axis(w, order, -1), np.take_along_axis(x, order, -1)
    return (w[0], x[0]) if squeeze else (w, x)
