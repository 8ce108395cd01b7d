#!/usr/bin/env python
"""bench.py -- filter time-steps/s of the moment-filter hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Benes--Bernoulli 1-D, N = 15, T = 1000, B = 4096 replicates PER GPU (weak
scaling: replicates are independent, sharded with no data-path collective; the per-replicate NLL vector is
all-gathered once per pass), central moments, TME-3, fp64, synthetic measurements.  A "step" is one pass of the hot
path over the whole batch (B x T filter time-steps), inputs already resident in HBM.

`value` counts LIVE filter time-steps only: at N = 15, T = 1000 a large share of replicates NaN-poison part-way (the
Hankel matrix loses positive definiteness -- the reference does the same, SURVEY.md section 7 hard part 3) and the
kernel stops computing for a poisoned replicate, so counting B x T would credit skipped work.  The nominal B x T rate
is reported next to it as `nominal_value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64 (half the 157.3 TF fp32 vector peak)

WORKLOADS = {
    # name: (model, N, T, B per GPU, mode, transition)
    'benes_bernoulli_N15_T1000_B4096_central_tme3': ('benes', 15, 1000, 4096, 'central', 'tme_3'),
    'benes_bernoulli_N7_T100_B4096_central_tme3': ('benes', 7, 100, 4096, 'central', 'tme_3'),
    'benes_bernoulli_N15_T1000_B4096_central_tme_normal3': ('benes', 15, 1000, 4096, 'central', 'tme_normal_3'),
    # SURVEY section 8d asks for the raw (and scaled) representation of config 2 next to the central one, with divergence counts
    'benes_bernoulli_N15_T1000_B4096_raw_tme3': ('benes', 15, 1000, 4096, 'raw', 'tme_3'),
    'benes_bernoulli_N15_T1000_B4096_scaled_tme3': ('benes', 15, 1000, 4096, 'scaled', 'tme_3'),
    'well_poisson_N7_T1000_B131072_central_tme_normal2': ('well', 7, 1000, 131072, 'central', 'tme_normal_2'),
    # BASELINE config 5 (d = 2): per-GPU shard of the 512-replicate batch is set with --B (128 on 4 GPUs)
    'prey_predator_N6_T500_B512_central_tme2': ('prey', 6, 500, 512, 'central', 'tme_2'),
    'prey_predator_N6_T500_B512_central_tme_normal2': ('prey', 6, 500, 512, 'central', 'tme_normal_2'),
}
DEFAULT_WORKLOAD = 'benes_bernoulli_N15_T1000_B4096_central_tme3'


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument('--workload', type=str, default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    p.add_argument('--B', type=int, default=0, help='override replicates per GPU')
    p.add_argument('--T', type=int, default=0, help='override time steps')
    p.add_argument('--chunk', type=int, default=0, help='time steps per kernel launch (0 = whole T in one launch)')
    p.add_argument('--no-moments', action='store_true', help='NLL only: do not stream the (B, T, 2N) moments out')
    p.add_argument('--cpu-seconds', type=float, default=12., help='target wall time of the cpu_baseline sample')
    p.add_argument('--no-cpu-baseline', action='store_true')
    return p.parse_args()


def build_model(model, N, transition, B, rng):
    from mfs_amd.one_dim import moments, ss_models
    if model == 'benes':
        dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
        theta = None
    else:
        dt, _, _, ic, drift0, dispersion, _, pmf0, _ = ss_models.well_poisson(3., N)
        # parameter grid of BASELINE config 4 (theta in [0.5, 6]^2), one point per replicate
        p1 = rng.uniform(0.5, 6., size=B)
        p2 = rng.uniform(0.5, 6., size=B)
        theta = (p1, p2)

        def drift(x):
            return drift0(x, p1)

        def pmf(y, x):
            return pmf0(y, x, p2)
    kind, order = transition.rsplit('_', 1)
    if kind == 'tme':
        fns = moments.sde_cond_moments_tme(drift, dispersion, dt, int(order))
    elif kind == 'tme_normal':
        fns = moments.sde_cond_moments_tme_normal(drift, dispersion, dt, int(order), N)
    else:
        raise ValueError(transition)
    return dt, ic, fns, pmf, theta


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with python -m torch.distributed.run --nproc-per-node N bench.py')
        args.gpus = world

    import ctypes as C
    from mfs_amd import _lib, synth, dist
    from mfs_amd.one_dim import filtering

    model, N, T, B, mode, transition = WORKLOADS[args.workload]
    B = args.B or B
    T = args.T or T
    L = _lib.lib()
    if model == 'prey':
        return main_nd(args, rank, local_rank, world, N, T, B, mode, transition)
    # one GPU per rank; on a box with fewer GPUs than ranks (rehearsals only) ranks share devices and RCCL, which
    # refuses duplicate GPUs, falls back to the reported host gather
    device = local_rank % max(_lib.device_count(), 1)
    comm = dist.Communicator.from_env(device=device)  # TCP control plane (mfs_amd/rdzv.py) + RCCL for the NLL gather
    _lib.check(L.mfs_set_device(device))

    # ---- synthetic inputs for this rank's shard (seeded per rank), uploaded before the timed region
    rng = np.random.default_rng(1234 + rank)
    dt, ic, fns, pmf, theta = build_model(model, N, transition, B, rng)
    if model == 'benes':
        ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=100 + rank)
    else:
        ys, _ = synth.well_poisson_batch(B, T, p1=3., p2=3., dt=dt, seed=100 + rank)
    tables, lik = filtering.trace_model(mode, fns[{'raw': 0, 'central': 1, 'scaled': 2}[mode]],
                                        fns[3] if mode == 'central' else None, pmf)
    mstruct, keep = filtering.build_model_struct(tables, lik, B)
    m0 = {'raw': ic.rms, 'central': ic.cms, 'scaled': ic.scms}[mode]
    d_m0 = _lib.DeviceBuffer.from_array(m0)
    d_mean0 = _lib.DeviceBuffer.from_array(np.array([ic.mean]))
    d_scale0 = _lib.DeviceBuffer.from_array(np.array([np.sqrt(ic.variance)]))
    d_ys = _lib.DeviceBuffer.from_array(ys)
    d_mom = None if args.no_moments else _lib.DeviceBuffer(B * T * 2 * N * 8)
    d_means = _lib.DeviceBuffer(B * T * 8)
    d_nell = _lib.DeviceBuffer(B * 8)
    d_fn = _lib.DeviceBuffer(B * 4)
    d_nell_all = _lib.DeviceBuffer(B * 8 * world)

    plan = C.c_void_p()
    _lib.check(L.mfs_plan_1d_create(C.byref(plan), C.byref(mstruct), _lib.MODE[mode], N, T, B, 0, args.chunk,
                                    device))
    geo = [C.c_int() for _ in range(4)]
    _lib.check(L.mfs_plan_1d_geometry(plan, *[C.byref(g) for g in geo]))
    stream = C.c_void_p()
    _lib.check(L.mfs_stream_create(C.byref(stream)))
    ev = [C.c_void_p() for _ in range(2 * max(args.steps, 1))]
    for e in ev:
        _lib.check(L.mfs_event_create(C.byref(e)))

    def one_pass(i=None):
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i], stream))
        _lib.check(L.mfs_plan_1d_run(plan, d_m0.ptr, 0, d_mean0.ptr, d_scale0.ptr, d_ys.ptr,
                                     d_mom.ptr if d_mom else None, d_means.ptr, None, d_nell.ptr, d_fn.ptr, stream))
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i + 1], stream))
        comm.allgather_nell(d_nell, d_nell_all, B, stream)  # RCCL over xGMI when world > 1; no-op copy otherwise

    for _ in range(args.warmup):
        one_pass()
    _lib.check(L.mfs_stream_synchronize(stream))
    comm.barrier()
    _lib.check(L.mfs_device_synchronize())
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_pass(i)
    _lib.check(L.mfs_stream_synchronize(stream))
    _lib.check(L.mfs_device_synchronize())
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.max_over_ranks(elapsed)

    # ---- per-launch kernel time from HIP events on the launch stream
    kern_ms = []
    for i in range(args.steps):
        ms = C.c_float()
        _lib.check(L.mfs_event_elapsed_ms(ev[2 * i], ev[2 * i + 1], C.byref(ms)))
        kern_ms.append(ms.value)
    kern_ms_avg = float(np.mean(kern_ms)) if kern_ms else float('nan')

    first_nan = d_fn.to_array((B,), np.int32)
    nell = d_nell.to_array((B,))
    live_steps = int(np.where(first_nan >= 0, first_nan + 1, T).sum())
    live_total = comm.sum_over_ranks(live_steps)
    alive_total = comm.sum_over_ranks(int((first_nan < 0).sum()))
    nell_all = d_nell_all.to_array((world * B,))
    gather_ok = bool(np.array_equal(nell_all[rank * B:(rank + 1) * B], nell, equal_nan=True))

    if rank == 0:
        nominal_steps = world * B * T
        value = live_total * args.steps / elapsed
        nominal = nominal_steps * args.steps / elapsed
        # algorithmic HBM bytes per launch: per live filter-step one y in, 2N moments + mean out
        per_step = 8 * (2 * N + (2 if mode != 'raw' else 1)) if d_mom else 8 * 2
        algo_bytes = live_steps * per_step + B * (8 * 2 * N + 8)
        hbm_gbs = algo_bytes / (kern_ms_avg * 1e-3) / 1e9
        M = int(transition.rsplit('_', 1)[1])
        flops_step = 22.7 * N ** 3 + (8 * M + 12) * N ** 2
        tflops = live_steps * flops_step / (kern_ms_avg * 1e-3) / 1e12
        # (the recorded counters belong to the workload's own size: no figure for --B / --T overrides)
        full_size = (B == WORKLOADS[args.workload][3] and T == WORKLOADS[args.workload][2])
        traffic = recorded_hbm_traffic(args.workload, d_mom is not None) if full_size else None
        out = {
            'metric': 'filter time-steps/sec', 'value': value, 'unit': 'filter-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'model': None, 'N': N, 'T': T, 'replicates_per_gpu': B,
                       'mode': mode, 'transition': transition, 'parallelism': f'replicate-sharded x{world}',
                       'lanes_per_filter': geo[0].value, 'filters_per_block': geo[1].value, 'grid': geo[2].value,
                       'lds_bytes_per_block': geo[3].value, 'chunk': args.chunk or T,
                       'moments_streamed_out': d_mom is not None},
            'nominal_value': nominal, 'live_fraction': live_total / nominal_steps,
            'replicates_alive_at_T': alive_total, 'replicates': world * B, 'nll_allgather_ok': gather_ok,
            'nll_allgather': ('rccl ncclAllGather' if comm.data == 'rccl' and world > 1 else
                              'single rank: device copy' if world == 1 else f'host fallback: {comm.rccl_error}'),
            'target_1e6_steps_per_s_met': bool(value >= 1e6 * world),
            'roofline': {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': hbm_gbs / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'mfs::filter1d_fast_kernel', 'avg_launch_ms': kern_ms_avg,
                         'traffic_source': 'profiles/*/pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate '
                                           'passes, FETCH_SIZE x2 on gfx950)' if traffic else None,
                         'algorithmic_bytes_per_launch': algo_bytes,
                         'note': 'latency/VALU-bound fp64 recursion, not HBM-bound: see valu_fp64'},
            'valu_fp64': {'achieved': tflops, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': tflops / FP64_VALU_PEAK_TFLOPS, 'algorithmic_flops_per_filter_step': flops_step},
        }
        if not args.no_cpu_baseline and world == 1:  # contract: the CPU baseline is a rank-0, N = 1 measurement
            out['cpu_baseline'] = cpu_baseline(args, model, N, T, mode, tables, lik, ic, ys, theta, nell, first_nan)
        print(json.dumps(out))
    comm.close()


def main_nd(args, rank, local_rank, world, N, T, B, mode, transition):
    """BASELINE config 5: the d = 2 prey--predator filter through the device-pointer N-D plan."""
    import ctypes as C
    from mfs_amd import _lib, synth, dist
    from mfs_amd.multi_dims import filtering, moments, ss_models
    from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
        gram_and_hankel_indices_graded_lexico
    L = _lib.lib()
    device = local_rank % max(_lib.device_count(), 1)
    comm = dist.Communicator.from_env(device=device)
    _lib.check(L.mfs_set_device(device))
    mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
    inds = gram_and_hankel_indices_graded_lexico(N, 2)
    z = mi.shape[0]
    dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(mi)
    kind, order = transition.rsplit('_', 1)
    if kind == 'tme':
        fns, sig = moments.sde_cond_moments_tme(drift, disp, dt, int(order)), 'multi-index'
    else:
        fns, sig = moments.sde_cond_moments_tme_normal(drift, disp, dt, int(order), mi), 'index'
    ys, _ = synth.prey_predator_batch(B, T, dt, seed=100 + rank)
    tables = filtering._trace_transition((fns[{'raw': 0, 'central': 1, 'scaled': 2}[mode]], sig), mode, (mi, inds))
    lik = filtering._trace_likelihood(pmf, 2)
    mstruct, keep = filtering._model_struct(tables, lik)
    scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))
    m0 = {'raw': gs.rms, 'central': gs.cms, 'scaled': gs.cms / np.prod(scale0 ** mi, axis=-1)}[mode]
    d_m0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(m0))
    d_mean0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(gs.mean, dtype=np.float64))
    d_scale0 = _lib.DeviceBuffer.from_array(scale0)
    d_ys = _lib.DeviceBuffer.from_array(ys)
    d_mom = None if args.no_moments else _lib.DeviceBuffer(B * T * z * 8)
    d_means = _lib.DeviceBuffer(B * T * 2 * 8)
    d_scales = _lib.DeviceBuffer(B * T * 2 * 8)
    d_nell = _lib.DeviceBuffer(B * 8)
    d_fn = _lib.DeviceBuffer(B * 4)
    d_nell_all = _lib.DeviceBuffer(B * 8 * world)
    mi32 = np.ascontiguousarray(mi, dtype=np.int32)
    inds32 = np.ascontiguousarray(inds, dtype=np.int32)
    plan = C.c_void_p()
    _lib.check(L.mfs_plan_nd_create(C.byref(plan), C.byref(mstruct), _lib.MODE[mode], N, T, B, z, _lib.ptr(mi32),
                                    _lib.ptr(inds32), 0, device))
    geo = [C.c_int() for _ in range(3)]
    _lib.check(L.mfs_plan_nd_geometry(plan, *[C.byref(g) for g in geo]))
    stream = C.c_void_p()
    _lib.check(L.mfs_stream_create(C.byref(stream)))
    ev = [C.c_void_p() for _ in range(2 * max(args.steps, 1))]
    for e in ev:
        _lib.check(L.mfs_event_create(C.byref(e)))

    def one_pass(i=None):
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i], stream))
        _lib.check(L.mfs_plan_nd_run(plan, d_m0.ptr, 0, d_mean0.ptr, d_scale0.ptr, d_ys.ptr,
                                     d_mom.ptr if d_mom else None, d_means.ptr, d_scales.ptr, d_nell.ptr, d_fn.ptr,
                                     stream))
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i + 1], stream))
        comm.allgather_nell(d_nell, d_nell_all, B, stream)

    for _ in range(args.warmup):
        one_pass()
    _lib.check(L.mfs_stream_synchronize(stream))
    comm.barrier()
    _lib.check(L.mfs_device_synchronize())
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_pass(i)
    _lib.check(L.mfs_stream_synchronize(stream))
    _lib.check(L.mfs_device_synchronize())
    comm.barrier()
    elapsed = comm.max_over_ranks(time.perf_counter() - t0)
    kern_ms = []
    for i in range(args.steps):
        ms = C.c_float()
        _lib.check(L.mfs_event_elapsed_ms(ev[2 * i], ev[2 * i + 1], C.byref(ms)))
        kern_ms.append(ms.value)
    kern_ms_avg = float(np.mean(kern_ms)) if kern_ms else float('nan')
    first_nan = d_fn.to_array((B,), np.int32)
    nell = d_nell.to_array((B,))
    live_steps = int(np.where(first_nan >= 0, first_nan + 1, T).sum())
    live_total = comm.sum_over_ranks(live_steps)
    alive_total = comm.sum_over_ranks(int((first_nan < 0).sum()))
    nell_all = d_nell_all.to_array((world * B,))
    gather_ok = bool(np.array_equal(nell_all[rank * B:(rank + 1) * B], nell, equal_nan=True))
    if rank == 0:
        nominal_steps = world * B * T
        value = live_total * args.steps / elapsed
        per_step = 8 * (z + (3 if mode == 'central' else 5 if mode == 'scaled' else 1)) if d_mom else 8 * 2
        algo_bytes = live_steps * per_step + B * 8 * (z + 3)
        hbm_gbs = algo_bytes / (kern_ms_avg * 1e-3) / 1e9
        flops_step = 2.0e6 if N == 6 else None   # SURVEY section 8d figure for config 5
        out = {
            'metric': 'filter time-steps/sec', 'value': value, 'unit': 'filter-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'model': None, 'd': 2, 'N': N, 'z': int(z), 'T': T,
                       'replicates_per_gpu': B, 'mode': mode, 'transition': transition,
                       'parallelism': f'replicate-sharded x{world}', 'threads_per_filter': geo[0].value,
                       'grid': geo[1].value, 'lds_bytes_per_block': geo[2].value,
                       'moments_streamed_out': d_mom is not None},
            'nominal_value': nominal_steps * args.steps / elapsed, 'live_fraction': live_total / nominal_steps,
            'replicates_alive_at_T': alive_total, 'replicates': world * B, 'nll_allgather_ok': gather_ok,
            'nll_allgather': ('rccl ncclAllGather' if comm.data == 'rccl' and world > 1 else
                              'single rank: device copy' if world == 1 else f'host fallback: {comm.rccl_error}'),
            'roofline': {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': hbm_gbs / HBM_PEAK_GBS,
                         'traffic': recorded_hbm_traffic(args.workload, d_mom is not None)
                         if (B == WORKLOADS[args.workload][3] and T == WORKLOADS[args.workload][2]) else None,
                         'kernel': 'mfs::filternd_kernel',
                         'avg_launch_ms': kern_ms_avg, 'algorithmic_bytes_per_launch': algo_bytes,
                         'traffic_source': 'profiles/*/pmc.json (rocprofv3 --pmc, separate passes; the write counter includes '
                                           'the kernel\'s scratch reload, see DESIGN.md section 3.3)',
                         'note': 'latency-bound small-matrix recursion (Jacobi rounds), not HBM-bound'},
        }
        if flops_step:
            tf = live_steps * flops_step / (kern_ms_avg * 1e-3) / 1e12
            out['valu_fp64'] = {'achieved': tf, 'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                'frac': tf / FP64_VALU_PEAK_TFLOPS, 'algorithmic_flops_per_filter_step': flops_step}
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline_nd(args, N, T, mode, transition, mi, inds, ys, nell)
        print(json.dumps(out))
    _lib.check(L.mfs_plan_nd_destroy(plan))
    comm.close()


def cpu_baseline_nd(args, N, T, mode, transition, mi, inds, ys, dev_nell):
    """The NumPy/LAPACK oracle (one core) on one replicate over a bounded number of steps; kind = "port"."""
    from oracle import multi_dims as omd, tme_sympy
    dt, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    kind, order = transition.rsplit('_', 1)
    if kind == 'tme':
        _, ocms, omean, _ = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, int(order), mi)
        sig = 'multi-index'
    else:
        _, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, int(order), mi)
        sig = 'index'
    steps = 4
    t0 = time.perf_counter()
    omd.moment_filter_nd_cms((ocms, sig), omean, opmf, ys[0, :steps], (mi, inds), ogs.cms, ogs.mean)
    el = time.perf_counter() - t0
    steps2 = int(min(T, max(steps, steps / el * args.cpu_seconds)))
    t0 = time.perf_counter()
    res = omd.moment_filter_nd_cms((ocms, sig), omean, opmf, ys[0, :steps2], (mi, inds), ogs.cms, ogs.mean)
    el = time.perf_counter() - t0
    out = {'value': steps2 / el, 'unit': 'filter-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'replicate 0, first {steps2} of T={T} steps of the same workload, oracle/multi_dims.py '
                     f'(NumPy + LAPACK, lambdified SymPy transition), {el:.1f} s'}
    if steps2 == T:
        out['nll_rel_diff_vs_device'] = float(abs(res[2] - dev_nell[0]) / abs(res[2]))
    return out


def recorded_hbm_traffic(workload, moments_streamed):
    """HBM bytes per launch from the committed PMC summary of this workload (counters cannot be read from inside the
    process; they are collected with rocprofv3 --pmc in separate passes and committed under profiles/)."""
    import glob
    best = None
    import re

    def natural(path):   # r01_v9 before r01_v10
        return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', path)]

    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'pmc.json')), key=natural):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get('workload') == workload and moments_streamed and 'hbm_bytes_per_launch' in rec:
            best = rec['hbm_bytes_per_launch']
    return best


def cpu_baseline(args, model, N, T, mode, tables, lik, ic, ys, theta, dev_nell, dev_first_nan):
    """The oracle's C port (OpenMP over replicates) timed on this box's host cores, on a bounded sample of the same
    workload.  The reference's own JAX-CPU path cannot run here (no JAX in the image): kind = "port"."""
    from oracle import c_oracle, tme_sympy, models as om
    # all host cores this process may run on, regardless of OMP_NUM_THREADS (torchrun exports OMP_NUM_THREADS=1)
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    # independent SymPy derivation of the coefficient tables where it is cheap (shared-parameter models)
    if model == 'benes':
        odt, _, _, odrift, odisp, _, _ = om.benes_bernoulli(N)
        M = int(tables.label.rsplit('_', 1)[1])
        full = tme_sympy.operator_tables_1d(odrift, odisp, odt, M, 'tanh')
        coef = full if tables.kind == 'operator' else np.stack([full[0], full[-1]])
    else:
        coef, _ = tables.table(ys.shape[0])
    kind = 0 if tables.kind == 'operator' else 1
    umap = 1 if tables.umap == 'tanh' else 0
    lik_kind = {'bernoulli_logistic': 0, 'poisson_softplus': 1, 'gaussian': 2}[lik.kind]
    m0 = {'raw': ic.rms, 'central': ic.cms, 'scaled': ic.scms}[mode]
    modei = {'raw': 0, 'central': 1, 'scaled': 2}[mode]

    def run(nb):
        cf = coef[:nb] if coef.ndim == 3 else coef
        lp = lik.params[:nb] if lik.params.ndim == 2 else lik.params
        t0 = time.perf_counter()
        res = c_oracle.filter_1d(modei, N, ys[:nb], m0, ic.mean, np.sqrt(ic.variance), kind, umap, tables.n_terms,
                                 cf, tables.mean_x_coef, lik_kind, lp, want_moments=False, nthreads=threads)
        return time.perf_counter() - t0, res

    nb = min(ys.shape[0], 4 * threads)
    el, res = run(nb)
    rate = nb * T / el
    nb2 = int(min(ys.shape[0], max(nb, rate * args.cpu_seconds / T)))
    if nb2 > nb:
        el, res = run(nb2)
        nb = nb2
    means = res[1]
    first = np.where(np.isnan(means).any(1), np.argmax(np.isnan(means), 1), T)
    live = int(first.sum())
    both = np.isfinite(res[3]) & np.isfinite(dev_nell[:nb])
    rel = np.abs(res[3][both] - dev_nell[:nb][both]) / np.abs(res[3][both]) if both.any() else np.array([np.nan])
    return {'value': live / el, 'unit': 'filter-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'first {nb} replicates x T={T} of the same workload, oracle/c/mfs_oracle.c, OpenMP x{threads}, '
                      f'{el:.1f} s; live steps only ({live / (nb * T):.2f} of nominal)',
            'nominal_value': nb * T / el,
            'nll_max_rel_diff_vs_device': float(np.max(rel)),
            # the maximum is set by the one or two replicates closest to losing positive definiteness (any change of
            # summation order moves them); the bulk shows the agreement of two fp64 implementations of the same algorithm
            'nll_rel_diff_vs_device_p50_p90_p99': [float(v) for v in np.quantile(rel, [0.5, 0.9, 0.99])],
            'replicates_finite_in_both': int(both.sum()),
            'replicates_finite_device_only': int((np.isfinite(dev_nell[:nb]) & ~np.isfinite(res[3])).sum()),
            'replicates_finite_cpu_only': int((~np.isfinite(dev_nell[:nb]) & np.isfinite(res[3])).sum())}


if __name__ == '__main__':
    main()
