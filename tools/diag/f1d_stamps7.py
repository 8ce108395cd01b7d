import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
from mfs_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath('tools/diag/libmfs_stamps7.so')
from mfs_amd.one_dim import filtering, moments, ss_models
N, T, B = 7, int(os.environ.get('STAMP_T', '200')), 64
rng = np.random.default_rng(0)
dt, _, _, ic, drift0, dispersion, _, pmf0, _ = ss_models.well_poisson(3., N)
p1 = rng.uniform(0.5, 6., size=B); p2 = rng.uniform(0.5, 6., size=B)
fns = moments.sde_cond_moments_tme_normal(lambda x: drift0(x, p1), dispersion, dt, 2, N)
ys, _ = synth.well_poisson_batch(B, T, p1=3., p2=3., dt=dt, seed=100)
m, means, nell, fn = filtering.moment_filter_cms(fns[1], fns[3], lambda y, x: pmf0(y, x, p2), ic.cms, ic.mean, ys,
                                                 return_first_nan=True)
st = (C.c_ulonglong * 16)()
L = _lib.lib(); L.mfs_debug_stamps.argtypes = [C.c_void_p]; print('rc', L.mfs_debug_stamps(st))
st = np.array(list(st), dtype=np.float64)
names = ['hankel gather', 'cholesky', 'jacobi coeffs', 'laguerre', 'weights', 'predict contributions', 'update contributions', 'moment reduction']
halves = st[9]
print('iterations per quadrature: predict half', st[11] / (st[9] / 2), 'update half', st[12] / (st[9] / 2))
print('filter 0 first_nan', fn[0], 'half-steps', halves, 'laguerre iterations per quadrature', st[10] / halves)
tot = st[:8].sum()
for i, n in enumerate(names):
    per = st[i] / (halves if i not in (5, 6) else halves / 2)
    print(f'{n:24s} {per:9.0f} cycles per occurrence   {100 * st[i] / tot:5.1f} % of stamped')
print('stamped cycles per step', tot / (halves / 2))
