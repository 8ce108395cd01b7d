"""CPU oracle for the mfs moment-filter hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy/SciPy/SymPy fp64 restatement of the reference's algorithm
(zgbkdlm/mfs, `mfs/one_dim/*`, `mfs/multi_dims/*`; each function cites the reference file:line it
follows).  It exists to check the HIP path; it is never the thing measured or shipped:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it;
  * nothing under `mfs_amd/` imports it, and `mfs_amd` raises loudly when `libmfs_hip.so` is missing.

Pinning status ("how do we know the restatement is the reference's algorithm?"):

  * The reference is pure Python/JAX.  JAX, jaxlib and `tme` are absent from this image (ordinary
    ModuleNotFoundError, not a permission denial) so the reference's hot path cannot be imported or
    executed here, and it holds no stored numeric golden arrays.
  * The oracle is therefore pinned by the reference's own analytic known-answer tests, restated in
    `tests/test_oracle_*.py`: Kalman-filter convergence (reference tests/test_filtering.py:82-111),
    rms/cms/scms routine equivalence (:113-164, :169-242), N-D == 1-D at d=1 and independent 2-D ==
    two 1-D filters (:244-329), quadrature exactness (tests/test_one_dim_quadrature.py:48-113,
    tests/test_multi_dim_quadrature.py), TME vs exact LTI discretisation
    (tests/test_one_dim_moments.py:90-118), `ldl_chol` == Cholesky (tests/test_utils.py:198-209),
    the N-D transition factories against exact LTI moments (tests/test_multi_dim_moments.py:149-246),
    the N-D rule on quadratic / mgf integrands and under basis reordering
    (tests/test_multi_dim_quadrature.py:169-216, 226-265) -- tests/test_oracle_pins.py --
    and by golden multi-index tables generated from the one reference module that does import
    (`mfs/multi_dims/multi_indices.py`, pure NumPy; `tests/golden/make_multi_indices_golden.py`).
  * Oracle outputs for BASELINE configs 1-5 are frozen under `tests/golden/filter_cfg*.npz`
    (`tests/golden/make_filter_golden.py`); the GPU parity tests read them.
  * Third-party arithmetic: the transition moments come from PyPI `tme` (>=0.1.5, unpinned; not
    vendored in the reference).  `oracle/tme_sympy.py` restates its published algorithm
    (Zhao et al., "Taylor moment expansion for continuous-discrete Gaussian filtering", IEEE TAC 2021)
    and is anchored on the reference's call sites and its LTI tests.
  * For the Benes (tanh-drift) model the reference holds no pinned value at all: **parity unpinned**
    by the reference at that model.  Anchors used instead: the exact Benes transition law, whose
    mean/variance the TME expansion reproduces exactly
    (tests/test_oracle_one_dim.py::test_benes_tme_matches_exact_law), and `oracle/exact_mp.py` -- the same
    filter in 80-digit arithmetic with exact rational TME tables (mpmath), the arbiter at N = 15 where two
    correct fp64 implementations (this package on LAPACK, `oracle/c`) disagree about when a replicate
    NaN-poisons (`tests/golden/filter_cfg2_exact*.npz`, `tests/test_gpu_envelope.py`).
  * `oracle/parity.py`: the error measures shared by the tests and `bench.py` (per-order moment floors,
    first-NaN agreement).
"""
