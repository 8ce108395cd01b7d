"""ctypes binding of libmfs_hip.so (C ABI: include/mfs_hip.h).  There is no CPU fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmfs_hip.so')

MFS_OK = 0
MODE = {'raw': 0, 'central': 1, 'scaled': 2}
MODE_ODD_TAIL = 0x100   # MFS_MODE_ODD_TAIL: 2N + 1 moments per row (include/mfs_hip.h)
TRANS = {'operator': 0, 'gaussian': 1}
UMAP = {'x': 0, 'tanh': 1}
LIK = {'bernoulli_logistic': 0, 'poisson_softplus': 1, 'gaussian': 2, 'bearing_gaussian': 3}
MAX_N = 32

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class MfsModel1d(C.Structure):
    """struct mfs_model_1d (include/mfs_hip.h)."""
    _fields_ = [('trans_kind', C.c_int32), ('umap', C.c_int32), ('n_terms', C.c_int32), ('degree', C.c_int32),
                ('n_rows', C.c_int32), ('coef_batched', C.c_int32), ('lik_kind', C.c_int32), ('n_lik', C.c_int32),
                ('lik_batched', C.c_int32), ('reserved', C.c_int32), ('mean_x_coef', C.c_double),
                ('coef', c_double_p), ('lik', c_double_p)]


class MfsModelNd(C.Structure):
    """struct mfs_model_nd (include/mfs_hip.h)."""
    _fields_ = [('d', C.c_int32), ('trans_kind', C.c_int32), ('n_terms', C.c_int32), ('extent', C.c_int32),
                ('n_factors', C.c_int32), ('ny', C.c_int32), ('fac_kind', C.c_int32 * 2),
                ('fac_component', C.c_int32 * 2), ('fac_ycol', C.c_int32 * 2), ('fac_n_par', C.c_int32 * 2),
                ('coef_batched', C.c_int32), ('lik_batched', C.c_int32), ('coef', c_double_p), ('lik', c_double_p)]


ND_TERMS = 14          # kappa terms with |kappa| <= 4 (TME order <= 2): the 16-row table layout
ND_ROWS = 16
ND_TERMS_MAX = 27      # ... |kappa| <= 6 (TME order 3): the 29-row layout
ND_ROWS_MAX = 29
ND_TRANS_OPERATOR, ND_TRANS_GAUSSIAN = 0, 1
ND_MAX_EXTENT = 6
ND_MAX_EXTENT_HI = 7
ND_MAX_FACTORS = 2
MAX_LIK = 4
ABI_VERSION = 2
# derivative multi-indices kappa, 1 <= |kappa| <= 6, graded-lex (the order of mfs_model_nd.coef rows; the first 14 are |kappa| <= 4)
ND_KAPPAS = [(a, s - a) for s in range(1, 7) for a in range(s + 1)]


def nd_table_rows(n_terms: int) -> int:
    """MFS_ND_TABLE_ROWS of include/mfs_hip.h."""
    return ND_ROWS_MAX if n_terms > ND_TERMS else ND_ROWS


class MfsError(RuntimeError):
    pass


_lib = None

# every symbol include/mfs_hip.h declares: (name, restype, argtypes)
_vp, _vpp, _i, _u64 = C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_uint64
_SIGNATURES = [
    ('mfs_version', _i, []),
    ('mfs_last_error', C.c_char_p, []),
    ('mfs_device_count', _i, [C.POINTER(C.c_int)]),
    ('mfs_set_device', _i, [_i]),
    ('mfs_device_synchronize', _i, []),
    ('mfs_device_name', _i, [_i, C.c_char_p, _i]),
    ('mfs_malloc', _i, [_vpp, _u64]),
    ('mfs_free', _i, [_vp]),
    ('mfs_memcpy_h2d', _i, [_vp, _vp, _u64, _vp]),
    ('mfs_memcpy_d2h', _i, [_vp, _vp, _u64, _vp]),
    ('mfs_memset', _i, [_vp, _i, _u64, _vp]),
    ('mfs_host_alloc', _i, [_vpp, _u64, _i]),
    ('mfs_host_free', _i, [_vp]),
    ('mfs_pool_trim', _i, [_i]),
    ('mfs_pool_stats', _i, [_i, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ('mfs_stream_create', _i, [_vpp]),
    ('mfs_stream_destroy', _i, [_vp]),
    ('mfs_stream_synchronize', _i, [_vp]),
    ('mfs_event_create', _i, [_vpp]),
    ('mfs_event_destroy', _i, [_vp]),
    ('mfs_event_record', _i, [_vp, _vp]),
    ('mfs_event_elapsed_ms', _i, [_vp, _vp, C.POINTER(C.c_float)]),
    ('mfs_filter_1d', _i, [C.POINTER(MfsModel1d), _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i,
                           _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    ('mfs_filter_1d_grad', _i, [C.POINTER(MfsModel1d), _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                _i, _vp]),
    ('mfs_plan_1d_create', _i, [_vpp, C.POINTER(MfsModel1d), _i, _i, _i, _i, _i, _i, _i]),
    ('mfs_plan_1d_run', _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ('mfs_plan_1d_destroy', _i, [_vp]),
    ('mfs_plan_1d_geometry', _i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int)]),
    ('mfs_quadrature_1d', _i, [_i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    ('mfs_characteristic_1d', _i, [_i, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp]),
    ('mfs_filter_nd', _i, [C.POINTER(MfsModelNd), _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i,
                           _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    ('mfs_plan_nd_create', _i, [_vpp, C.POINTER(MfsModelNd), _i, _i, _i, _i, _i, _vp, _vp, _i, _i]),
    ('mfs_plan_nd_run', _i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ('mfs_plan_nd_destroy', _i, [_vp]),
    ('mfs_plan_nd_geometry', _i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ('mfs_elementary', _i, [_i, _i, _vp, _vp, _i]),
    ('mfs_comm_unique_id', _i, [_vp]),
    ('mfs_comm_init', _i, [_vpp, _vp, _i, _i, _i]),
    ('mfs_allgather_nell', _i, [_vp, _vp, _vp, _u64, _vp]),
    ('mfs_comm_destroy', _i, [_vp]),
    ('mfs_memcpy_d2d', _i, [_vp, _vp, _u64, _vp]),
]
DECLARED_SYMBOLS = [s[0] for s in _SIGNATURES]


def lib():
    """Load libmfs_hip.so once; raise loudly (never fall back) if it is absent or does not load."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MfsError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'or `make -C mfs_amd/csrc`. mfs_amd has no CPU fallback.')
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise MfsError(f'cannot load {LIB_PATH}: {e}') from e
        for name, res, args in _SIGNATURES:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != MFS_OK:
        raise MfsError(f'libmfs_hip error {rc}: {lib().mfs_last_error().decode()}')


def ptr(a):
    """void* of a C-contiguous float64 / int32 NumPy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags['C_CONTIGUOUS']
    return a.ctypes.data_as(C.c_void_p)


_pinned_live = [0]   # bytes of pinned pool memory currently held by result arrays of this process


class _PinnedBlock:
    """Page-locked host memory from the library's pool; goes back to the pool when the last array viewing it dies."""

    def __init__(self, nbytes, device):
        p = C.c_void_p()
        check(lib().mfs_host_alloc(C.byref(p), max(int(nbytes), 8), device))
        self.ptr = p
        self.nbytes = int(nbytes)
        _pinned_live[0] += self.nbytes

    def __del__(self):
        try:
            if self.ptr is not None and self.ptr.value:
                lib().mfs_host_free(self.ptr)
                self.ptr = None
                _pinned_live[0] -= self.nbytes
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64, device=0):
    """np.empty(shape, dtype) on page-locked memory from the library's pool (mfs_host_alloc): device-to-host copies
    into it run at the PCIe rate (~4x the pageable rate) and need no first-touch page faults.  MFS_PINNED_OUTPUTS=0
    falls back to ordinary NumPy memory."""
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    if os.environ.get('MFS_PINNED_OUTPUTS', '1') == '0':
        return np.empty(shape, dtype=dtype)
    dt = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    # page-locked memory cannot be swapped: beyond MFS_PINNED_BUDGET_MB (default 4096) of it held by live result arrays, and
    # whenever the pinned allocation fails, the array is ordinary (pageable) NumPy memory -- slower copies, same results
    budget = int(float(os.environ.get('MFS_PINNED_BUDGET_MB', '4096')) * (1 << 20))
    if _pinned_live[0] + n > budget:
        return np.empty(shape, dtype=dtype)
    try:
        block = _PinnedBlock(n, device)
    except MfsError:
        return np.empty(shape, dtype=dtype)
    # the array's base chain keeps `block` alive (and nothing points back, so reference counting alone frees it):
    # ndarray -> ctypes array -> (attribute) block
    buf = (C.c_char * max(n, 8)).from_address(block.ptr.value)
    buf._mfs_block = block
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)


def pool_trim(device=0):
    """Give the idle blocks of the library's device and pinned pools back to the runtime (mfs_pool_trim).  Result arrays
    returned by the filters live in the pinned pool until they are garbage-collected; a sweep that keeps many of them can
    call this after dropping them, or set MFS_PINNED_OUTPUTS=0 / MFS_PINNED_BUDGET_MB."""
    check(lib().mfs_pool_trim(device))


def pool_stats(device=0):
    v = [_u64() for _ in range(4)]
    check(lib().mfs_pool_stats(device, *[C.byref(x) for x in v]))
    return dict(zip(('device_bytes', 'pinned_bytes', 'device_allocs', 'pinned_allocs'), (x.value for x in v)))


def device_count():
    n = C.c_int(0)
    check(lib().mfs_device_count(C.byref(n)))
    return n.value


def device_name(device=0):
    buf = C.create_string_buffer(256)
    check(lib().mfs_device_name(device, buf, 256))
    return buf.value.decode()


class DeviceBuffer:
    """Owning handle of a hipMalloc'ed buffer (bench harness / sharded driver keep data resident in HBM)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().mfs_malloc(C.byref(p), self.nbytes))
        self.ptr = p

    @classmethod
    def from_array(cls, a, stream=None):
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        if a.nbytes:
            check(lib().mfs_memcpy_h2d(buf.ptr, ptr(a), a.nbytes, stream))
        return buf

    def to_array(self, shape, dtype=np.float64, stream=None):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(lib().mfs_memcpy_d2h(ptr(out), self.ptr, out.nbytes, stream))
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            lib().mfs_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
