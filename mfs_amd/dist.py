"""Multi-GPU plumbing: one process per GPU, replicates sharded across ranks, one all-gather of the NLL vector.

Rendezvous, barriers and scalar reductions ride on `torch.distributed` with the gloo backend (CPU side; launched by
`python -m torch.distributed.run`).  The data-path exchange -- the per-replicate NLL all-gather -- is RCCL over xGMI
through the C ABI (`mfs_comm_*`, `mfs_allgather_nell`, include/mfs_hip.h), on the filter's own HIP stream.  The same
class runs with backend='gloo' on CPU-only boxes (world_size-2 tests), where the gather goes through host memory.
"""
import ctypes as C
import os

import numpy as np

from mfs_amd import _lib


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous block split of the replicate axis: rank g owns [lo, hi)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class Communicator:
    def __init__(self, rank=0, world=1, local_rank=0, backend='rccl'):
        self.rank, self.world, self.local_rank, self.backend = rank, world, local_rank, backend
        self._td = None
        self._comm = None

    @classmethod
    def from_env(cls, backend='rccl'):
        rank = int(os.environ.get('RANK', '0'))
        world = int(os.environ.get('WORLD_SIZE', '1'))
        local_rank = int(os.environ.get('LOCAL_RANK', '0'))
        self = cls(rank, world, local_rank, backend)
        if world > 1:
            import torch.distributed as td
            if not td.is_initialized():
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                td.init_process_group(backend='gloo', rank=rank, world_size=world)
            self._td = td
            if backend == 'rccl':
                self._init_rccl()
        return self

    def _init_rccl(self):
        import torch
        L = _lib.lib()
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(L.mfs_comm_unique_id(C.cast(idbuf, C.c_void_p)))
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8)
        self._td.broadcast(t, src=0)
        idbuf = (C.c_char * 128).from_buffer_copy(bytes(t.tolist()))
        comm = C.c_void_p()
        _lib.check(L.mfs_comm_init(C.byref(comm), C.cast(idbuf, C.c_void_p), self.world, self.rank, self.local_rank))
        self._comm = comm

    # -- control plane (host)
    def barrier(self):
        if self._td is not None:
            self._td.barrier()

    def _reduce(self, v, op):
        if self._td is None:
            return v
        import torch
        t = torch.tensor([float(v)], dtype=torch.float64)
        self._td.all_reduce(t, op=op)
        return t.item()

    def max_over_ranks(self, v: float) -> float:
        return self._reduce(v, self._td.ReduceOp.MAX) if self._td is not None else v

    def sum_over_ranks(self, v):
        return int(round(self._reduce(v, self._td.ReduceOp.SUM))) if self._td is not None else v

    # -- data plane
    def allgather_nell(self, d_send: '_lib.DeviceBuffer', d_recv: '_lib.DeviceBuffer', count: int, stream=None):
        """Device buffers: d_recv[rank * count : (rank + 1) * count] <- every rank's d_send[:count]."""
        L = _lib.lib()
        if self.world == 1:
            _lib.check(L.mfs_memcpy_d2d(d_recv.ptr, d_send.ptr, count * 8, stream))
        elif self.backend == 'rccl':
            _lib.check(L.mfs_allgather_nell(self._comm, d_send.ptr, d_recv.ptr, count, stream))
        else:
            raise RuntimeError('device all-gather needs backend="rccl"')

    def allgather_host(self, local: np.ndarray) -> np.ndarray:
        """Host all-gather of equal-length float64 vectors (gloo); used by the CPU tests and by ragged shards."""
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self._td is None:
            return local.copy()
        import torch
        out = [torch.empty(local.shape[0], dtype=torch.float64) for _ in range(self.world)]
        self._td.all_gather(out, torch.from_numpy(local))
        return np.concatenate([o.numpy() for o in out])

    def close(self):
        if self._comm is not None:
            _lib.lib().mfs_comm_destroy(self._comm)
            self._comm = None
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
