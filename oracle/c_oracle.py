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
build_flags = '-O3 -march=x86-64-v3 -fopenmp (prebuilt, portable)'


def use_native_build(outdir):
    """bench.py's cpu_baseline leg: compile the same source with -march=native for THIS host (the prebuilt .so is
    x86-64-v3 because it travels between machines) into `outdir` and use that build from now on.  Must be called before
    the first lib(); returns the flags in use (the portable build's when gcc is missing or fails)."""
    global _PATH, build_flags
    if _lib is not None:
        return build_flags
    try:
        os.makedirs(outdir, exist_ok=True)
        out = os.path.join(outdir, 'libmfs_oracle_native.so')
        subprocess.check_call(['gcc', '-O3', '-march=native', '-fopenmp', '-fPIC', '-shared', '-o', out,
                               os.path.join(_HERE, 'mfs_oracle.c'), '-lm'], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _PATH, build_flags = out, '-O3 -march=native -fopenmp (built on this host)'
    except (OSError, subprocess.CalledProcessError):
        pass
    return build_flags


def host_cores():
    """(cores this process can actually use, affinity-mask size): the cgroup CPU quota bounds the former -- a container
    with a 16-CPU share on a 256-thread host shows 256 in its affinity mask, and 256 OpenMP threads then time-slice 16."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:            # cgroup v2: "<quota|max> <period>"
            q, per = f.read().split()
            if q != 'max':
                quota = int(q) / int(per)
    except (OSError, ValueError):
        try:
            with open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us') as f, open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    cores = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return cores, aff


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
