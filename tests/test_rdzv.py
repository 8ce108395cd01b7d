"""The TCP control plane (mfs_amd/rdzv.py): wire encoding without pickle, token handshake, stale port files."""
import os
import socket
import struct
import threading
import time

import numpy as np
import pytest

from mfs_amd import rdzv


def test_wire_encoding_round_trip_and_rejects_objects():
    vec = np.array([1.5, np.nan, -np.inf, 3.0])
    obj = [None, True, 7, 2.5, float('nan'), float('inf'), 'text', b'\x00\x01\xff' * 43, (b'id' * 64, None),
           [1, (2.0, 'x')], vec]
    a, b = socket.socketpair()
    try:
        rdzv._send(a, obj)
        back = rdzv._recv(b)
    finally:
        a.close()
        b.close()
    assert back[0] is None and back[1] is True and back[2] == 7 and back[3] == 2.5
    assert np.isnan(back[4]) and back[5] == float('inf') and back[6] == 'text'
    assert back[7] == b'\x00\x01\xff' * 43
    assert isinstance(back[8], tuple) and back[8] == (b'id' * 64, None)
    assert back[9] == [1, (2.0, 'x')]
    np.testing.assert_array_equal(back[10], vec)

    class Thing:
        pass
    with pytest.raises(TypeError):
        rdzv._enc(Thing())
    assert 'pickle' not in open(rdzv.__file__).read().split('"""', 2)[2]   # no unpickling of socket bytes, at all


def _run_world(world, key, results, errors, delay_rank0=0.):
    def worker(rank):
        try:
            if rank == 0 and delay_rank0:
                time.sleep(delay_rank0)
            r = rdzv.TcpRendezvous(rank, world, key=key, timeout=30.)
            results[rank] = r.allgather((rank, float(rank) / 2))
            r.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
    ths = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(60.)
    assert not any(t.is_alive() for t in ths)


def test_stale_port_file_and_stranger_are_survived(tmp_path, monkeypatch):
    """A port file left by a crashed run (same key) points at a dead port -- and, worse, at a port somebody else now
    listens on.  Non-zero ranks re-read the file until rank 0 has replaced it; a connection that does not present the
    token is dropped by rank 0."""
    monkeypatch.setattr(rdzv.tempfile, 'gettempdir', lambda: str(tmp_path))
    key = 'stale_test'
    # somebody else's listener on the stale port: accepts and says nothing useful
    other = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    other.bind(('127.0.0.1', 0))
    other.listen(4)
    other.settimeout(0.2)
    stop = threading.Event()

    def stranger_server():
        while not stop.is_set():
            try:
                c, _ = other.accept()
                c.close()
            except OSError:
                pass
    th = threading.Thread(target=stranger_server)
    th.start()
    (tmp_path / f'mfs_rdzv_{key}.port').write_text(f'{other.getsockname()[1]} deadbeef')
    results, errors = {}, []
    try:
        _run_world(3, key, results, errors, delay_rank0=0.5)
    finally:
        stop.set()
        th.join()
        other.close()
    assert not errors, errors
    for r in range(3):
        assert results[r] == [(0, 0.0), (1, 0.5), (2, 1.0)]
    assert not (tmp_path / f'mfs_rdzv_{key}.port').exists()     # rank 0 removed its file on close


def test_connection_without_token_is_dropped(tmp_path, monkeypatch):
    monkeypatch.setattr(rdzv.tempfile, 'gettempdir', lambda: str(tmp_path))
    key = 'token_test'
    results, errors = {}, []
    path = tmp_path / f'mfs_rdzv_{key}.port'

    def intruder():
        deadline = time.time() + 20
        while not path.exists() and time.time() < deadline:
            time.sleep(0.01)
        port = int(path.read_text().split()[0])
        s = socket.create_connection(('127.0.0.1', port), timeout=5)
        payload = b'[1, "not-the-token"]'
        s.sendall(struct.pack('<Q', len(payload)) + payload)
        try:
            assert s.recv(16) == b''          # rank 0 closes without an answer
        except OSError:
            pass
        s.close()
        results['intruder'] = True

    th = threading.Thread(target=intruder)
    th.start()
    # rank 1 joins late, after the intruder has claimed to be rank 1
    def late_rank1():
        time.sleep(1.0)
        r = rdzv.TcpRendezvous(1, 2, key=key, timeout=30.)
        results[1] = r.allgather('one')
        r.close()
    t1 = threading.Thread(target=late_rank1)
    t1.start()
    r0 = rdzv.TcpRendezvous(0, 2, key=key, timeout=30.)
    results[0] = r0.allgather('zero')
    r0.close()
    th.join(30)
    t1.join(30)
    assert results.get('intruder') and results[0] == ['zero', 'one'] and results[1] == ['zero', 'one']
    assert oct(os.stat(tmp_path).st_mode)  # (directory still there)
