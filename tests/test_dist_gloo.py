"""The N > 1 path on CPU: two gloo ranks shard the replicates, run the (injected) filter on their block and all-gather
the per-replicate NLLs in replicate order.  On the GPU box the same Communicator gathers through RCCL instead."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from mfs_amd import dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_partition():
    for B in (0, 1, 7, 8, 4096, 4099):
        for world in (1, 2, 3, 8):
            blocks = [dist.shard_bounds(B, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_single_rank_sharded_nell_is_identity():
    comm = dist.Communicator()
    ys = np.arange(12.).reshape(4, 3)
    out = dist.sharded_nell(lambda y: y.sum(axis=1), ys, comm)
    np.testing.assert_array_equal(out, ys.sum(axis=1))


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from mfs_amd import dist, synth
    from oracle import one_dim as o, models as om, tme_sympy

    comm = dist.Communicator.from_env(**%(kw)s)
    N, T, B = 4, 25, 5                       # B = 5 over 2 ranks: ragged shards
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    fns = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 2, 2 * N)
    ys, _ = synth.benes_bernoulli_batch(B, T, odt, seed=3)

    def filt(ys_shard):                      # the oracle stands in for the HIP filter on this CPU-only box
        return np.array([o.moment_filter_cms(fns[1], fns[3], opmf, oic.cms, oic.mean, y)[2] for y in ys_shard])

    full = dist.sharded_nell(filt, ys, comm)
    lo, hi = dist.shard_bounds(B, comm.world, comm.rank)
    t = comm.max_over_ranks(float(comm.rank))
    s = comm.sum_over_ranks(hi - lo)
    comm.barrier()
    np.save(os.path.join(%(out)r, f'rank{comm.rank}.npy'), np.concatenate([full, [t, s, lo, hi]]))
    comm.close()
''')


import pytest


@pytest.mark.parametrize('kw,port', [({'backend': 'gloo'}, 29541), ({'control': 'tcp', 'data': 'host'}, 29543)])
def test_two_rank_sharding_and_allgather(tmp_path, kw, port):
    """world_size 2 on CPU: once over torch.distributed / gloo, once over the TCP control plane bench.py uses."""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT, 'out': str(tmp_path), 'kw': repr(kw)})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    r0, r1 = np.load(tmp_path / 'rank0.npy'), np.load(tmp_path / 'rank1.npy')
    np.testing.assert_array_equal(r0[:5], r1[:5])          # every rank holds the full vector
    assert np.all(np.isfinite(r0[:5]))
    assert r0[5] == 1.0 and r1[5] == 1.0                    # max over ranks
    assert r0[6] == 5 and r1[6] == 5                        # shard sizes sum to B
    assert (r0[7], r0[8], r1[7], r1[8]) == (0, 3, 3, 5)     # contiguous block split

    # replicate order is preserved: compare with an unsharded evaluation
    from mfs_amd import synth
    from oracle import one_dim as o, models as om, tme_sympy
    N, T, B = 4, 25, 5
    odt, _, oic, odrift, odisp, _, opmf = om.benes_bernoulli(N)
    fns = tme_sympy.sde_cond_moments_tme_1d(odrift, odisp, odt, 2, 2 * N)
    ys, _ = synth.benes_bernoulli_batch(B, T, odt, seed=3)
    ref = np.array([o.moment_filter_cms(fns[1], fns[3], opmf, oic.cms, oic.mean, y)[2] for y in ys])
    np.testing.assert_allclose(r0[:5], ref, rtol=1e-13)


FAIL_WORKER = textwrap.dedent('''
    import json, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from mfs_amd import dist
    comm = dist.Communicator.from_env(control='tcp', data='rccl')      # what bench.py asks for; there is no GPU here
    gathered = comm.allgather_host(np.array([float(comm.rank)]))          # the host route still works ...
    status = comm.exit_status(gather_ok=True, allow_host_gather=%(allow)r)
    with open(os.path.join(%(out)r, f'rank{comm.rank}.json'), 'w') as f:
        json.dump({'degraded': comm.degraded, 'data': comm.data, 'error': comm.rccl_error, 'gathered': gathered.tolist(),
                   'status': status}, f)
    comm.close()
    sys.exit(status)                                                      # ... but the run does not count as a success
''')


@pytest.mark.parametrize('allow,expected', [(False, dist.EXIT_RCCL_FAILED), (True, 0)])
def test_multi_rank_run_without_rccl_is_a_visible_failure(tmp_path, allow, expected):
    """with WORLD_SIZE > 1, a run whose NLL gather cannot go through RCCL (here: no GPU at all) says so
    -- `degraded`, an error text, exit status 3 on every rank -- and still delivers the gather through host memory; only
    an explicit --allow-host-gather turns that into status 0."""
    import json
    script = tmp_path / 'worker.py'
    script.write_text(FAIL_WORKER % {'root': ROOT, 'out': str(tmp_path), 'allow': allow})
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT='29547', MFS_RCCL_INIT_TIMEOUT='30',
                   HIP_VISIBLE_DEVICES='')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    codes = [p.wait(timeout=180) for p in procs]
    assert codes == [expected, expected], [p.stderr.read().decode()[-400:] for p in procs]
    for rank in range(2):
        r = json.load(open(tmp_path / f'rank{rank}.json'))
        assert r['degraded'] is True and r['data'] == 'host' and r['error']
        assert r['gathered'] == [0.0, 1.0]
        assert r['status'] == expected


@pytest.mark.parametrize('allow,expected', [(False, dist.EXIT_RCCL_FAILED), (True, 0)])
def test_bench_self_launches_its_ranks(allow, expected):
    """`python bench.py --gpus 2` without torchrun.  The parent only supervises (it never loads the
    library or counts devices), starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank
    0's single JSON line and leaves with the worst child status -- here, on a box without a GPU, the RCCL failure status
    3 (0 with --allow-host-gather).  --dry-run stops after the communicator: no workload, no kernel."""
    import json
    env = dict(os.environ, MFS_RCCL_INIT_TIMEOUT='30', HIP_VISIBLE_DEVICES='')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'] + (['--allow-host-gather'] if allow else [])
    r = subprocess.run(cmd, env=env, capture_output=True, timeout=300)
    assert r.returncode == expected, r.stderr.decode()[-600:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['dry_run'] is True and line['nll_allgather_ok'] is False
    assert 'FAILED' in line['nll_allgather']


def test_bench_launcher_parent_never_touches_the_gpu_library():
    """The launcher branch must run before anything of mfs_amd is imported: a parent that had initialised the GPU could
    not safely start children (and on the driver's node it would hold a device the ranks need)."""
    import ast
    src = open(os.path.join(ROOT, 'bench.py')).read()
    tree = ast.parse(src)
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'main')
    launch_line = next(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, 'id', '') == 'launch_ranks')
    first_import = min(n.lineno for n in ast.walk(main) if isinstance(n, (ast.Import, ast.ImportFrom)))
    assert launch_line < first_import
    launcher = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'launch_ranks')
    imported = {a.name for n in ast.walk(launcher) if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    assert imported <= {'socket', 'subprocess', 'threading'}, imported
    top = {a.name.split('.')[0] for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    assert 'mfs_amd' not in top and 'torch' not in top
