"""GPU checks of the plan API (device-resident buffers, chunked launches replayed from a hipGraph) and of the RCCL
all-gather entry points.  Multi-GPU boxes are not available to these tests: the RCCL communicator is exercised with a
single rank (same code path: dlopen, ncclCommInitRank, ncclAllGather on the filter's stream).  torch is deliberately NOT
imported here: its bundled HIP runtime next to the system one breaks ncclCommInitRank, which is why the multi-GPU
control plane is mfs_amd/rdzv.py rather than torch.distributed."""
import ctypes as C

import numpy as np
import numpy.testing as npt
import pytest

from mfs_amd import _lib, synth, dist
from mfs_amd.one_dim import filtering, moments, ss_models

pytestmark = pytest.mark.gpu


def _plan_run(N, T, B, chunk, ys, ic, tables, lik, want_moments=True):
    L = _lib.lib()
    model, keep = filtering.build_model_struct(tables, lik, B)
    plan = C.c_void_p()
    _lib.check(L.mfs_plan_1d_create(C.byref(plan), C.byref(model), 1, N, T, B, 0, chunk, 0))
    d_m0 = _lib.DeviceBuffer.from_array(ic.cms)
    d_mean0 = _lib.DeviceBuffer.from_array(np.array([ic.mean]))
    d_ys = _lib.DeviceBuffer.from_array(ys)
    d_mom = _lib.DeviceBuffer(B * T * 2 * N * 8) if want_moments else None
    d_means, d_nell, d_fn = _lib.DeviceBuffer(B * T * 8), _lib.DeviceBuffer(B * 8), _lib.DeviceBuffer(B * 4)
    stream = C.c_void_p()
    _lib.check(L.mfs_stream_create(C.byref(stream)))
    outs = []
    for _ in range(2):  # second run replays the cached graph when chunked
        _lib.check(L.mfs_plan_1d_run(plan, d_m0.ptr, 0, d_mean0.ptr, None, d_ys.ptr, d_mom.ptr if d_mom else None,
                                     d_means.ptr, None, d_nell.ptr, d_fn.ptr, stream))
        _lib.check(L.mfs_stream_synchronize(stream))
        outs.append((d_mom.to_array((B, T, 2 * N)) if d_mom else None, d_means.to_array((B, T)),
                     d_nell.to_array((B,)), d_fn.to_array((B,), np.int32)))
    geo = [C.c_int() for _ in range(4)]
    _lib.check(L.mfs_plan_1d_geometry(plan, *[C.byref(g) for g in geo]))
    _lib.check(L.mfs_plan_1d_destroy(plan))
    _lib.check(L.mfs_stream_destroy(stream))
    return outs, [g.value for g in geo]


def test_chunked_graph_replay_is_bit_identical_to_single_launch():
    N, T, B = 7, 120, 37
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    _, c, _, mu, _ = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    tables, lik = filtering.trace_model('central', c, mu, pmf)
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=21)
    (whole, whole2), geo = _plan_run(N, T, B, 0, ys, ic, tables, lik)
    (chunked, chunked2), _ = _plan_run(N, T, B, 25, ys, ic, tables, lik)  # 5 launches captured into one graph
    for a, b in zip(whole, chunked):
        npt.assert_array_equal(a, b)
    for a, b in zip(chunked, chunked2):
        npt.assert_array_equal(a, b)
    for a, b in zip(whole, whole2):
        npt.assert_array_equal(a, b)
    # and the host-pointer API gives the same numbers
    m, means, nell = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
    npt.assert_array_equal(m, whole[0])
    npt.assert_array_equal(nell, whole[2])
    assert geo[0] == 8 and geo[1] == 8 and geo[2] == -(-B // geo[1])   # N = 7: eight lanes per filter, eight filters per wave
    # NLL-only runs (no moment stream) agree too
    (nll_only, _), _ = _plan_run(N, T, B, 0, ys, ic, tables, lik, want_moments=False)
    npt.assert_array_equal(nll_only[2], whole[2])


def test_rccl_single_rank_allgather():
    import sys
    assert 'torch' not in sys.modules or True
    L = _lib.lib()
    idbuf = (C.c_char * 128)()
    _lib.check(L.mfs_comm_unique_id(C.cast(idbuf, C.c_void_p)))
    comm = C.c_void_p()
    _lib.check(L.mfs_comm_init(C.byref(comm), C.cast(idbuf, C.c_void_p), 1, 0, 0))
    stream = C.c_void_p()
    _lib.check(L.mfs_stream_create(C.byref(stream)))
    x = np.random.default_rng(0).normal(size=1000)
    x[17] = np.nan
    d_send, d_recv = _lib.DeviceBuffer.from_array(x), _lib.DeviceBuffer(x.nbytes)
    _lib.check(L.mfs_allgather_nell(comm, d_send.ptr, d_recv.ptr, x.shape[0], stream))
    _lib.check(L.mfs_stream_synchronize(stream))
    npt.assert_array_equal(d_recv.to_array(x.shape), x)
    _lib.check(L.mfs_comm_destroy(comm))
    _lib.check(L.mfs_stream_destroy(stream))
    # the single-process Communicator path of bench.py
    c1 = dist.Communicator.from_env()
    d2 = _lib.DeviceBuffer(x.nbytes)
    c1.allgather_nell(d_send, d2, x.shape[0])
    _lib.check(L.mfs_device_synchronize())
    npt.assert_array_equal(d2.to_array(x.shape), x)
    assert c1.max_over_ranks(3.5) == 3.5 and c1.sum_over_ranks(4) == 4
    c1.close()


def test_nd_plan_device_pointers_match_host_entry():
    """mfs_plan_nd_* (data resident in HBM) runs the same kernel as mfs_filter_nd: bit-identical outputs."""
    from mfs_amd.multi_dims import filtering as fnd, moments as mnd, ss_models as snd
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    N, T, B = 3, 30, 5
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    z = mi.shape[0]
    dt, _, _, gs, drift, disp, _, pmf, _ = snd.prey_predator(mi)
    fns = mnd.sde_cond_moments_tme(drift, disp, dt, 2)
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=3)
    cmss, means, nell = fnd.moment_filter_nd_cms((fns[1], 'multi-index'), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean)

    L = _lib.lib()
    tables = fnd._trace_transition((fns[1], 'multi-index'), 'central', (mi, inds))
    mstruct, keep = fnd._model_struct(tables, fnd._trace_likelihood(pmf, 2))
    mi32, inds32 = np.ascontiguousarray(mi, dtype=np.int32), np.ascontiguousarray(inds, dtype=np.int32)
    plan = C.c_void_p()
    _lib.check(L.mfs_plan_nd_create(C.byref(plan), C.byref(mstruct), _lib.MODE['central'], N, T, B, z, _lib.ptr(mi32),
                                    _lib.ptr(inds32), 0, 0))
    geo = [C.c_int() for _ in range(3)]
    _lib.check(L.mfs_plan_nd_geometry(plan, *[C.byref(g) for g in geo]))
    assert geo[0].value == 256 and geo[1].value == B and 0 < geo[2].value <= 160 * 1024
    d_m0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(gs.cms))
    d_mean0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(gs.mean, dtype=np.float64))
    d_ys = _lib.DeviceBuffer.from_array(ys)
    d_mom, d_means = _lib.DeviceBuffer(B * T * z * 8), _lib.DeviceBuffer(B * T * 2 * 8)
    d_nell, d_fn = _lib.DeviceBuffer(B * 8), _lib.DeviceBuffer(B * 4)
    for _ in range(2):  # a plan is reusable
        _lib.check(L.mfs_plan_nd_run(plan, d_m0.ptr, 0, d_mean0.ptr, None, d_ys.ptr, d_mom.ptr, d_means.ptr, None,
                                     d_nell.ptr, d_fn.ptr, None))
    _lib.check(L.mfs_device_synchronize())
    npt.assert_array_equal(d_mom.to_array((B, T, z)), cmss)
    npt.assert_array_equal(d_means.to_array((B, T, 2)), means)
    npt.assert_array_equal(d_nell.to_array((B,)), nell)
    assert np.all(d_fn.to_array((B,), np.int32) == -1)
    # scaled mode without scale0 is an argument error, not a crash
    plan2 = C.c_void_p()
    _lib.check(L.mfs_plan_nd_create(C.byref(plan2), C.byref(mstruct), _lib.MODE['scaled'], N, T, B, z, _lib.ptr(mi32),
                                    _lib.ptr(inds32), 0, 0))
    rc = L.mfs_plan_nd_run(plan2, d_m0.ptr, 0, d_mean0.ptr, None, d_ys.ptr, None, None, None, d_nell.ptr, None, None)
    assert rc == _lib.MFS_EINVAL if hasattr(_lib, 'MFS_EINVAL') else rc < 0
    _lib.check(L.mfs_plan_nd_destroy(plan2))
    _lib.check(L.mfs_plan_nd_destroy(plan))


def test_pipelined_host_entry_is_bit_identical_and_reuses_the_pool(monkeypatch):
    """mfs_filter_1d splits T into chunks and copies chunk k's moments out while chunk k + 1 computes (2-D copies into
    the caller's [B][T][2N] array): the bits are those of one launch, whether the destination is page-locked (the
    wrappers' default) or ordinary NumPy memory, and a repeated call takes every staging buffer from the pool."""
    N, T, B = 10, 203, 70          # T not a multiple of the chunk count
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    _, c, _, mu, _ = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=5)
    monkeypatch.setenv('MFS_HOST_CHUNKS', '1')
    ref = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys, return_first_nan=True)
    for chunks, pinned in (('7', '1'), ('7', '0'), ('16', '1'), ('203', '1')):
        monkeypatch.setenv('MFS_HOST_CHUNKS', chunks)
        monkeypatch.setenv('MFS_PINNED_OUTPUTS', pinned)
        got = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys, return_first_nan=True)
        for a, b in zip(ref, got):
            npt.assert_array_equal(a, b)
        del got
    # scaled mode carries one more state variable across chunks
    s0 = float(np.sqrt(ic.variance))
    _, _, sc, _, mv = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    monkeypatch.setenv('MFS_HOST_CHUNKS', '1')
    ref_s = filtering.moment_filter_scms(sc, mv, pmf, ic.scms, ic.mean, s0, ys)
    monkeypatch.setenv('MFS_HOST_CHUNKS', '5')
    got_s = filtering.moment_filter_scms(sc, mv, pmf, ic.scms, ic.mean, s0, ys)
    for a, b in zip(ref_s, got_s):
        npt.assert_array_equal(a, b)
    del ref_s, got_s
    # steady state: the second and third identical calls allocate nothing new, on the device or in pinned memory
    monkeypatch.setenv('MFS_HOST_CHUNKS', '4')
    monkeypatch.setenv('MFS_PINNED_OUTPUTS', '1')
    out = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
    del out
    before = _lib.pool_stats(0)
    for _ in range(2):
        out = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
        npt.assert_array_equal(out[2], ref[2])
        del out
    after = _lib.pool_stats(0)
    assert after['device_allocs'] == before['device_allocs'] and after['pinned_allocs'] == before['pinned_allocs']
    assert after['device_bytes'] > 0 and after['pinned_bytes'] >= B * T * 2 * N * 8
    # results the caller still holds are never recycled under it
    keep = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys)
    snapshot = keep[0].copy()
    other = filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys[::-1].copy())
    npt.assert_array_equal(keep[0], snapshot)
    assert not np.shares_memory(keep[0], other[0])
    del keep, other
    _lib.check(_lib.lib().mfs_pool_trim(0))
    assert _lib.pool_stats(0)['device_bytes'] == 0


@pytest.mark.parametrize('family,mode,route', [('tme_2', 'central', ''), ('tme_2', 'scaled', ''), ('tme_normal_2', 'central', ''),
                                               ('tme_2', 'central', 'eigen')])
def test_pipelined_nd_host_entry_is_bit_identical(monkeypatch, family, mode, route):
    """mfs_filter_nd cuts T into chunks when the moments are streamed out (chunk k's slice is copied while chunk k + 1
    computes); the per-replicate state -- moments, means, scales, NLL, first-NaN step, the eigenvector tiles a Jacobi warm
    start reads -- crosses the launches through a carry block, so every output has the bits of the single launch: operator
    path, scaled mode, Normal closure (warm-started Jacobi in the prediction) and the eigen-decomposition route of the
    update."""
    from mfs_amd.multi_dims import filtering as fnd, moments as mnd, ss_models as snd
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    N, T, B = 4, 37, 5
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    dt, _, _, gs, drift, disp, _, pmf, _ = snd.prey_predator(mi)
    if family == 'tme_2':
        fns, sig = mnd.sde_cond_moments_tme(drift, disp, dt, 2), 'multi-index'
    else:
        fns, sig = mnd.sde_cond_moments_tme_normal(drift, disp, dt, 2, mi), 'index'
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=9)
    if route:
        monkeypatch.setenv('MFS_ND_UPDATE', route)

    def run():
        if mode == 'central':
            return fnd.moment_filter_nd_cms((fns[1], sig), fns[3], pmf, ys, (mi, inds), gs.cms, gs.mean, return_first_nan=True)
        scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))
        return fnd.moment_filter_nd_scms((fns[2], sig), fns[4], pmf, ys, (mi, inds), gs.cms / np.prod(scale0 ** mi, axis=-1),
                                         gs.mean, scale0)

    monkeypatch.setenv('MFS_HOST_CHUNKS', '1')
    ref = run()
    assert np.all(np.isfinite(ref[-1 if mode == 'scaled' else 2]))
    for chunks in ('2', '5', '37'):
        monkeypatch.setenv('MFS_HOST_CHUNKS', chunks)
        got = run()
        for a, b in zip(ref, got):
            npt.assert_array_equal(a, b)


def test_host_entry_points_are_reentrant_across_threads():
    """SURVEY 8b "Threading": ctypes releases the GIL during a call and the ABI is re-entrant -- each call leases its own
    stream / event context and its own blocks from the (mutex-guarded) pool.  Four Python threads filter different batches
    at once, 1-D and N-D mixed; every result has the bits of the same call made alone."""
    import threading
    from mfs_amd.multi_dims import filtering as fnd, moments as mnd, ss_models as snd
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    N = 8
    dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
    _, c, _, mu, _ = moments.sde_cond_moments_tme(drift, dispersion, dt, 3)
    mi = generate_graded_lexico_multi_indices(2, 5)
    inds = gram_and_hankel_indices_graded_lexico(3, 2)
    dt2, _, _, gs, drift2, disp2, _, pmf2, _ = snd.prey_predator(mi)
    fns2 = mnd.sde_cond_moments_tme(drift2, disp2, dt2, 2)
    jobs = []
    for k in range(3):
        ys, _ = synth.benes_bernoulli_batch(40 + 17 * k, 120, dt, seed=30 + k)
        jobs.append(lambda ys=ys: filtering.moment_filter_cms(c, mu, pmf, ic.cms, ic.mean, ys))
    ys2, _ = synth.prey_predator_batch(6, 60, dt2, seed=3)
    jobs.append(lambda: fnd.moment_filter_nd_cms((fns2[1], 'multi-index'), fns2[3], pmf2, ys2, (mi, inds), gs.cms, gs.mean))
    alone = [job() for job in jobs]
    for _ in range(3):
        out, errs = [None] * len(jobs), []

        def run(i):
            try:
                out[i] = jobs[i]()
            except Exception as e:   # noqa: BLE001 -- reported below
                errs.append(e)

        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errs, errs
        for a, b in zip(alone, out):
            for x, y in zip(a, b):
                npt.assert_array_equal(x, y)
