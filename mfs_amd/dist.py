"""Multi-GPU plumbing: one process per GPU, replicates sharded across ranks, one all-gather of the NLL vector.

Launched by `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).  The control
plane (rendezvous, barriers, scalar reductions, the 128-byte RCCL id) is a small TCP all-gather (`mfs_amd/rdzv.py`) or,
on CPU-only boxes and in the tests, `torch.distributed` with gloo.  The data-path exchange -- the per-replicate NLL
all-gather -- is RCCL over xGMI through the C ABI (`mfs_comm_*`, `mfs_allgather_nell`, include/mfs_hip.h), enqueued on
the filter's own HIP stream.
"""
import ctypes as C
import os

import numpy as np

from mfs_amd import _lib


EXIT_RCCL_FAILED = 3   # exit status of a multi-rank run whose NLL gather did not go through RCCL


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous block split of the replicate axis: rank g owns [lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Communicator:
    """Control plane + NLL all-gather for one-process-per-GPU runs.

    control = 'tcp'  : mfs_amd.rdzv.TcpRendezvous (default; no torch in the process -- see rdzv.py for why)
    control = 'gloo' : torch.distributed with the gloo backend (CPU-only boxes / tests)
    data    = 'rccl' : ncclAllGather of device buffers through the C ABI;  'host': gather through the control plane
    """

    def __init__(self, rank=0, world=1, local_rank=0, control='tcp', data='rccl'):
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.control, self.data = control, data
        self._td = None
        self._rdzv = None
        self._comm = None
        self.rccl_error = None
        self._abandoned = False     # a RCCL initialisation that never returned is still sitting on a worker thread
        self.device = local_rank

    @classmethod
    def from_env(cls, backend=None, control=None, data=None, device=None):
        """`backend='gloo'` is shorthand for control='gloo', data='host' (the CPU test configuration)."""
        if backend == 'gloo':
            control, data = 'gloo', 'host'
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self = cls(rank, world, local_rank, control or 'tcp', data or 'rccl')
        self.device = local_rank if device is None else device
        if world > 1:
            if self.control == 'gloo':
                import torch.distributed as td
                if not td.is_initialized():
                    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                    td.init_process_group(backend='gloo', rank=rank, world_size=world)
                self._td = td
            else:
                from mfs_amd.rdzv import TcpRendezvous
                self._rdzv = TcpRendezvous(rank, world)
            if self.data == 'rccl':
                try:
                    self._init_rccl()
                except _lib.MfsError as e:
                    # never silent: the failure is recorded and reported by bench.py; the gather then goes through
                    # host memory so that the run (and its timing) still completes
                    self.rccl_error = str(e)
                    self.data = 'host'
                # every rank must take the same path
                if any(self._allgather_obj(self.data != 'rccl')):
                    if self._comm is not None and not self._abandoned:
                        _lib.lib().mfs_comm_destroy(self._comm)
                    self._comm = None
                    if self.rccl_error is None:
                        self.rccl_error = 'another rank could not join the RCCL communicator'
                    self.data = 'host'
        return self

    def _allgather_obj(self, obj):
        if self.world == 1:
            return [obj]
        if self._rdzv is not None:
            return self._rdzv.allgather(obj)
        out = [None] * self.world
        self._td.all_gather_object(out, obj)
        return out

    def _init_rccl(self, timeout: float = None):
        """ncclCommInitRank through the C ABI.  The call blocks until every rank has joined; a rank that died or never
        got there would leave the others waiting forever, so it runs on a worker thread (ctypes releases the GIL) and is
        given `timeout` seconds (MFS_RCCL_INIT_TIMEOUT, default 180).  On expiry this rank reports the failure and the
        run continues on the host route; the worker is abandoned and `close()` then leaves through os._exit with
        `EXIT_RCCL_FAILED` (an abandoned collective initialisation is never a successful run)."""
        import threading
        L = _lib.lib()
        if timeout is None:
            timeout = float(os.environ.get('MFS_RCCL_INIT_TIMEOUT', '180'))
        idbuf = (C.c_char * 128)()
        err = None
        if self.rank == 0:
            try:
                _lib.check(L.mfs_comm_unique_id(C.cast(idbuf, C.c_void_p)))
            except _lib.MfsError as e:     # e.g. librccl missing: tell the others instead of leaving them in the gather
                err = str(e)
        uid, err0 = self._allgather_obj((bytes(idbuf), err))[0]
        if err0 is not None:
            raise _lib.MfsError(err0)
        idbuf = (C.c_char * 128).from_buffer_copy(uid)
        comm = C.c_void_p()
        result = {}

        def work():
            try:
                _lib.check(L.mfs_comm_init(C.byref(comm), C.cast(idbuf, C.c_void_p), self.world, self.rank, self.device))
                result['ok'] = True
            except Exception as e:   # noqa: BLE001 -- reported below
                result['err'] = str(e)

        th = threading.Thread(target=work, name='mfs-rccl-init', daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            self._abandoned = True
            raise _lib.MfsError(f'ncclCommInitRank did not return within {timeout:.0f} s on rank {self.rank}')
        if 'err' in result:
            raise _lib.MfsError(result['err'])
        self._comm = comm

    @property
    def degraded(self) -> bool:
        """True when a multi-rank run asked for the RCCL data path and is not on it (see `rccl_error`)."""
        return self.world > 1 and self.rccl_error is not None

    def exit_status(self, gather_ok: bool = True, allow_host_gather: bool = False) -> int:
        """Process exit status of a run on this communicator: `EXIT_RCCL_FAILED` for a multi-rank run whose NLL gather did
        not go through RCCL (or did not reproduce the ranks' values), unless the caller accepted the host route."""
        if self.world > 1 and (self.degraded or not gather_ok) and not allow_host_gather:
            return EXIT_RCCL_FAILED
        return 0

    # -- control plane (host)
    def barrier(self):
        if self._rdzv is not None:
            self._rdzv.barrier()
        elif self._td is not None:
            self._td.barrier()

    def max_over_ranks(self, v: float) -> float:
        return max(self._allgather_obj(float(v)))

    def sum_over_ranks(self, v):
        return sum(self._allgather_obj(v))

    # -- data plane
    def allgather_nell(self, d_send: '_lib.DeviceBuffer', d_recv: '_lib.DeviceBuffer', count: int, stream=None):
        """Device buffers: d_recv[rank * count : (rank + 1) * count] <- every rank's d_send[:count]."""
        L = _lib.lib()
        if self.world == 1:
            _lib.check(L.mfs_memcpy_d2d(d_recv.ptr, d_send.ptr, count * 8, stream))
        elif self.data == 'rccl':
            _lib.check(L.mfs_allgather_nell(self._comm, d_send.ptr, d_recv.ptr, count, stream))
        else:
            # host route (RCCL unavailable, see rccl_error): D2H, control-plane gather, H2D
            _lib.check(L.mfs_stream_synchronize(stream))
            local = d_send.to_array((count,), stream=stream)
            allv = self.allgather_host(local)
            _lib.check(L.mfs_memcpy_h2d(d_recv.ptr, _lib.ptr(allv), allv.nbytes, stream))

    def allgather_host(self, local: np.ndarray) -> np.ndarray:
        """Host all-gather of equal-length float64 vectors through the control plane (tests, ragged shards)."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        return np.concatenate(self._allgather_obj(local))

    def close(self, status: int = None):
        """`status`: the exit status the caller is about to leave with (`exit_status(...)`); only the abandoned-init path
        uses it, because that path cannot return to the caller."""
        if self._abandoned:
            # the interpreter's shutdown would wait on (or tear down under) the stuck RCCL call: results are out, leave
            import sys
            try:
                if self._rdzv is not None:
                    self._rdzv.set_timeout(10.)     # a peer that is stuck or gone must not hold this rank for minutes
                self.barrier()
            except Exception:   # noqa: BLE001 -- leaving anyway
                pass
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(EXIT_RCCL_FAILED if status is None else status)
        if self._comm is not None:
            _lib.lib().mfs_comm_destroy(self._comm)
            self._comm = None
        if self._rdzv is not None:
            self._rdzv.close()
            self._rdzv = None
        if self._td is not None and self._td.is_initialized():
            self._td.barrier()
            self._td.destroy_process_group()
            self._td = None


def sharded_nell(filter_fn, ys: np.ndarray, comm: Communicator, per_replicate=()):
    """Run `filter_fn(ys_shard, *shards of per_replicate arrays) -> nell (b,)` on this rank's block of replicates and
    all-gather the per-replicate NLLs so that every rank holds the full (B,) vector, in replicate order.

    `filter_fn` is the HIP filter in production (e.g. a closure over `moment_filter_cms`); the CPU tests inject the
    oracle.  Ragged shards (B not divisible by the world size) are padded with NaN for the gather and trimmed."""
    B = ys.shape[0]
    lo, hi = shard_bounds(B, comm.world, comm.rank)
    nell = np.asarray(filter_fn(ys[lo:hi], *[np.asarray(a)[lo:hi] for a in per_replicate]), dtype=np.float64)
    width = -(-B // comm.world)
    padded = np.full((width,), np.nan)
    padded[:hi - lo] = nell
    allv = comm.allgather_host(padded).reshape(comm.world, width)
    return np.concatenate([allv[r, :shard_bounds(B, comm.world, r)[1] - shard_bounds(B, comm.world, r)[0]]
                           for r in range(comm.world)])
