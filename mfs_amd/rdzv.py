"""A minimal single-node rendezvous / control plane over TCP (no torch in the GPU processes).

`bench.py --gpus N` is launched by `python -m torch.distributed.run`, which exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT.  Importing torch inside a worker would load torch's bundled HIP runtime and RCCL next to the
system ones libmfs_hip.so is linked against -- two HIP runtimes in one process make ncclCommInitRank fail
("unhandled cuda error", observed on MI355X / ROCm 7.2 + torch 2.10 rocm7.0).  The control plane therefore needs
nothing but sockets: rank 0 listens on an ephemeral port of MASTER_ADDR and publishes it in a file keyed by the
launcher's pid and MASTER_PORT; the other ranks connect; every collective is an all-gather of small pickled objects
through rank 0 (barriers, max / sum of scalars, the 128-byte RCCL id).  Data never travels this way.
"""
import os
import pickle
import socket
import struct
import tempfile
import time


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack('<Q', len(data)) + data)


def _recv(sock):
    hdr = b''
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        hdr += chunk
    n = struct.unpack('<Q', hdr)[0]
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return pickle.loads(bytes(buf))


class TcpRendezvous:
    def __init__(self, rank: int, world: int, addr: str = None, key: str = None, timeout: float = 300.):
        self.rank, self.world = rank, world
        addr = addr or os.environ.get('MASTER_ADDR', '127.0.0.1')
        key = key or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        self._path = os.path.join(tempfile.gettempdir(), f'mfs_rdzv_{key}.port')
        self._peers = []
        self._sock = None
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, 0))
            srv.listen(world)
            srv.settimeout(timeout)
            tmp = self._path + f'.{os.getpid()}.tmp'
            with open(tmp, 'w') as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, self._path)
            peers = {}
            while len(peers) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(timeout)
                peers[_recv(conn)] = conn
            srv.close()
            self._peers = [peers[r] for r in range(1, world)]
        else:
            deadline = time.time() + timeout
            port = None
            while port is None:
                try:
                    with open(self._path) as f:
                        port = int(f.read().strip())
                except (OSError, ValueError):
                    if time.time() > deadline:
                        raise TimeoutError(f'rank 0 never published {self._path}')
                    time.sleep(0.01)
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            while True:
                try:
                    s.connect((addr, port))
                    break
                except ConnectionRefusedError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.01)
            _send(s, rank)
            self._sock = s

    def allgather(self, obj):
        """Rank-ordered list of every rank's `obj`."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [_recv(p) for p in self._peers]
            for p in self._peers:
                _send(p, objs)
            return objs
        _send(self._sock, obj)
        return _recv(self._sock)

    def barrier(self):
        self.allgather(None)

    def close(self):
        if self.world > 1:
            try:
                self.barrier()
            except Exception:
                pass
        for p in self._peers:
            p.close()
        if self._sock is not None:
            self._sock.close()
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._path)
            except OSError:
                pass
        self._peers, self._sock = [], None
