"""ctypes wrapper of oracle/c/libmfs_oracle.so (TEST INFRASTRUCTURE; see oracle/__init__.py).

The C port takes the same coefficient tables as the device; so that it stays an independent check, the tables passed
here should come from `oracle.tme_sympy.operator_tables_1d` (SymPy derivation), not from mfs_amd.tme_poly."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'c')
_PATH = os.path.join(_HERE, 'libmfs_oracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            subprocess.check_call(['make', '-C', _HERE])
        _lib = C.CDLL(_PATH)
        _lib.mfs_oracle_filter_1d.restype = C.c_int
        _lib.mfs_oracle_quadrature_1d.restype = C.c_int
        _lib.mfs_oracle_max_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def max_threads():
    return lib().mfs_oracle_max_threads()


def filter_1d(mode, N, ys, m0, mean0, scale0, trans_kind, umap, K, coef, mean_x_coef, lik_kind, lik,
              stable=False, want_moments=True, nthreads=0):
    """mode in {0 raw, 1 central, 2 scaled}; coef (n_rows, J+1) or (B, n_rows, J+1); lik (P,) or (B, P);
    ys (B, T); m0 (2N,) or (B, 2N).  Returns (moments or None, means, scales, nell)."""
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    B, T = ys.shape
    coef = np.ascontiguousarray(coef, dtype=np.float64)
    lik = np.ascontiguousarray(lik, dtype=np.float64)
    m0 = np.ascontiguousarray(m0, dtype=np.float64)
    batched = m0.ndim == 2
    nb = B if batched else 1
    mean0 = np.ascontiguousarray(np.broadcast_to(np.asarray(0. if mean0 is None else mean0, dtype=np.float64), (nb,)))
    scale0 = np.ascontiguousarray(np.broadcast_to(np.asarray(1. if scale0 is None else scale0, dtype=np.float64), (nb,)))
    out_m = np.empty((B, T, 2 * N)) if want_moments else None
    out_mean, out_scale, out_nell = np.empty((B, T)), np.empty((B, T)), np.empty((B,))
    rc = lib().mfs_oracle_filter_1d(
        C.c_int(mode), C.c_int(N), C.c_int(T), C.c_int(B), C.c_int(trans_kind), C.c_int(umap), C.c_int(K),
        C.c_int(coef.shape[-1] - 1), C.c_int(coef.shape[-2]), _p(coef), C.c_int(int(coef.ndim == 3)),
        C.c_double(mean_x_coef), C.c_int(lik_kind), C.c_int(lik.shape[-1]), _p(lik), C.c_int(int(lik.ndim == 2)),
        _p(m0), C.c_int(int(batched)), _p(mean0), _p(scale0), _p(ys), C.c_int(int(stable)), _p(out_m), _p(out_mean),
        _p(out_scale), _p(out_nell), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f'mfs_oracle_filter_1d failed with {rc}')
    return out_m, out_mean, out_scale, out_nell


def quadrature_1d(ms, mean=None, scale=None, stable=False):
    ms = np.ascontiguousarray(ms, dtype=np.float64)
    B, M2 = ms.shape
    N = M2 // 2
    mean = None if mean is None else np.ascontiguousarray(np.broadcast_to(np.asarray(mean, dtype=np.float64), (B,)))
    scale = None if scale is None else np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (B,)))
    w, x = np.empty((B, N)), np.empty((B, N))
    rc = lib().mfs_oracle_quadrature_1d(C.c_int(N), C.c_int(B), _p(ms), _p(mean), _p(scale), C.c_int(int(stable)),
                                        _p(w), _p(x))
    if rc != 0:
        raise RuntimeError(f'mfs_oracle_quadrature_1d failed with {rc}')
    return w, x
