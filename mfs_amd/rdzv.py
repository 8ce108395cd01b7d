"""A minimal single-node rendezvous / control plane over TCP (no torch in the GPU processes).

`bench.py --gpus N` is launched by `python -m torch.distributed.run`, which exports RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT.  Importing torch inside a worker would load torch's bundled HIP runtime and RCCL next to the
system ones libmfs_hip.so is linked against -- two HIP runtimes in one process make ncclCommInitRank fail
("unhandled cuda error", observed on MI355X / ROCm 7.2 + torch 2.10 rocm7.0).  The control plane therefore needs
nothing but sockets: rank 0 listens on an ephemeral loopback port and publishes `port token` in a 0600 file keyed by
the launcher's pid and MASTER_PORT; the other ranks connect and present the token; every collective is an all-gather of
small values through rank 0 (barriers, max / sum of scalars, the 128-byte RCCL id).  Data never travels this way.

Wire format: 8-byte little-endian length + UTF-8 JSON.  Only None / bool / int / float / str / list / tuple, bytes and
float64 vectors are representable (`_enc` / `_dec`); nothing received from a socket is ever executed or unpickled.
"""
import base64
import hmac
import json
import os
import secrets
import socket
import struct
import tempfile
import time

import numpy as np

_MAX_MESSAGE = 64 << 20


def _enc(obj):
    if obj is None or isinstance(obj, (bool, int, str)):
        return obj
    if isinstance(obj, float):
        # JSON has no NaN / inf: carry non-finite floats by name
        return obj if np.isfinite(obj) else {'f': repr(obj)}
    if isinstance(obj, (np.floating, np.integer, np.bool_)):
        return _enc(obj.item())
    if isinstance(obj, (bytes, bytearray)):
        return {'b': base64.b64encode(bytes(obj)).decode('ascii')}
    if isinstance(obj, np.ndarray):
        if obj.dtype != np.float64:      # no silent conversion: an int64 index vector would come back as floats
            raise TypeError(f'the rendezvous control plane carries float64 vectors only, not {obj.dtype}')
        a = np.ascontiguousarray(obj)
        return {'a': base64.b64encode(a.tobytes()).decode('ascii'), 's': list(a.shape)}
    if isinstance(obj, (list, tuple)):
        return {'t' if isinstance(obj, tuple) else 'l': [_enc(v) for v in obj]}
    raise TypeError(f'the rendezvous control plane cannot carry a {type(obj).__name__}')


def _dec(v):
    if isinstance(v, dict):
        if 'f' in v:
            return float(v['f'])
        if 'b' in v:
            return base64.b64decode(v['b'])
        if 'a' in v:
            return np.frombuffer(base64.b64decode(v['a']), dtype=np.float64).reshape(v['s']).copy()
        if 't' in v:
            return tuple(_dec(x) for x in v['t'])
        if 'l' in v:
            return [_dec(x) for x in v['l']]
        raise ValueError('malformed rendezvous message')
    return v


def _send(sock, obj):
    data = json.dumps(_enc(obj), allow_nan=False).encode('utf-8')
    sock.sendall(struct.pack('<Q', len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _recv(sock, limit=_MAX_MESSAGE):
    n = struct.unpack('<Q', _recv_exact(sock, 8))[0]
    if n > limit:
        raise ConnectionError(f'rendezvous message of {n} bytes refused')
    return _dec(json.loads(_recv_exact(sock, n).decode('utf-8')))


class TcpRendezvous:
    """Single-node: binds and connects on loopback unless `addr` (or MFS_RDZV_ADDR) says otherwise."""

    def __init__(self, rank: int, world: int, addr: str = None, key: str = None, timeout: float = 300.):
        self.rank, self.world = rank, world
        addr = addr or os.environ.get('MFS_RDZV_ADDR', '127.0.0.1')
        key = key or f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        self._path = os.path.join(tempfile.gettempdir(), f'mfs_rdzv_{key}.port')
        self._peers = []
        self._sock = None
        if world == 1:
            return
        if rank == 0:
            self._serve(addr, timeout)
        else:
            self._join(addr, timeout)

    def _serve(self, addr, timeout):
        try:    # a file left behind by a crashed run with the same key must not send anybody to a dead port
            os.remove(self._path)
        except OSError:
            pass
        token = secrets.token_hex(16)
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, 0))
        srv.listen(self.world)
        srv.settimeout(timeout)
        tmp = self._path + f'.{os.getpid()}.tmp'
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)   # the token is for this user's processes only
        with os.fdopen(fd, 'w') as f:
            f.write(f'{srv.getsockname()[1]} {token}')
        os.replace(tmp, self._path)
        peers = {}
        deadline = time.time() + timeout
        while len(peers) < self.world - 1:
            if time.time() > deadline:
                raise TimeoutError(f'only {len(peers) + 1} of {self.world} ranks reached the rendezvous')
            conn, _ = srv.accept()
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.settimeout(10.)
            try:
                hello = _recv(conn)
                ok = (isinstance(hello, (list, tuple)) and len(hello) == 2 and isinstance(hello[0], int)
                      and isinstance(hello[1], str) and hmac.compare_digest(hello[1], token)
                      and 1 <= hello[0] < self.world and hello[0] not in peers)
            except Exception:  # noqa: BLE001 -- a stranger on the port: drop it and keep listening
                ok = False
            if not ok:
                conn.close()
                continue
            _send(conn, 'ok')
            conn.settimeout(timeout)
            peers[hello[0]] = conn
        srv.close()
        self._peers = [peers[r] for r in range(1, self.world)]

    def _join(self, addr, timeout):
        deadline = time.time() + timeout
        while True:
            if time.time() > deadline:
                raise TimeoutError(f'rank {self.rank} could not reach rank 0 through {self._path}')
            # the file is re-read on every attempt: a stale one (crashed run, same key) is replaced by rank 0
            try:
                with open(self._path) as f:
                    port_s, token = f.read().split()
                port = int(port_s)
            except (OSError, ValueError):
                time.sleep(0.01)
                continue
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(10.)
            try:
                s.connect((addr, port))
                _send(s, (self.rank, token))
                if _recv(s) != 'ok':
                    raise ConnectionError('rendezvous handshake refused')
            except (OSError, ValueError):   # refused, reset, timed out, or not our rank 0 behind that port
                s.close()
                time.sleep(0.05)
                continue
            s.settimeout(timeout)
            self._sock = s
            return

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
        return _recv(self._sock, _MAX_MESSAGE * self.world)     # the reply is the concatenation of every rank's part

    def set_timeout(self, seconds: float):
        """Socket timeout of every later collective (a rank leaving after a failure must not wait for the full one)."""
        for p in self._peers:
            p.settimeout(seconds)
        if self._sock is not None:
            self._sock.settimeout(seconds)

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
