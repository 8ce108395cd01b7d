"""Generate golden multi-index tables from the one reference module that imports without JAX.

`/root/reference/mfs/multi_dims/multi_indices.py` depends only on math/numpy/typing/functools; it is loaded by file
path (bypassing `mfs/multi_dims/__init__.py`, which pulls in JAX).  Run in the build container only (the reference
never travels to the GPU box):

    python tests/golden/make_multi_indices_golden.py

Writes tests/golden/multi_indices.npz holding, per (N, d): `mi_N{N}_d{d}` = generate_graded_lexico_multi_indices(d,
2N-1) and `inds_N{N}_d{d}` = gram_and_hankel_indices_graded_lexico(N, d), plus a few index-of / size spot values.
"""
import importlib.util
import os

import numpy as np

REF = '/root/reference/mfs/multi_dims/multi_indices.py'
spec = importlib.util.spec_from_file_location('ref_multi_indices', REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
for (N, d) in [(3, 1), (3, 2), (4, 2), (5, 2), (6, 2), (2, 3), (3, 3)]:
    out[f'mi_N{N}_d{d}'] = ref.generate_graded_lexico_multi_indices(d, 2 * N - 1, 0)
    out[f'inds_N{N}_d{d}'] = ref.gram_and_hankel_indices_graded_lexico(N, d)

probe = np.array([[0, 0, 0], [1, 0, 2], [3, 1, 0], [0, 4, 1], [2, 2, 2], [5, 0, 0]])
out['probe_mi'] = probe
out['probe_index'] = np.array([ref.graded_lexico_indexof_multi_index(list(p)) for p in probe])
out['probe_index_lower2'] = np.array([ref.graded_lexico_indexof_multi_index(list(p), lower_sum=2)
                                      for p in probe[1:]])
sizes = [(1, 5, 0), (2, 5, 0), (3, 4, 2), (4, 3, 3), (2, 1, 3)]
out['size_args'] = np.array(sizes)
out['size_vals'] = np.array([ref.sizeof_multi_indices(*a) for a in sizes])
out['mi_lower_d3'] = ref.generate_graded_lexico_multi_indices(3, 4, 2)

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'multi_indices.npz'), **out)
print({k: v.shape for k, v in out.items()})
