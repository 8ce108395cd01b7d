#!/usr/bin/env python
"""bench.py -- filter time-steps/s of the moment-filter hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (self-launching: the parent starts one child per GPU, below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no WORLD_SIZE in the environment the process is a LAUNCHER: it never touches the GPU (no library
load, no device count), starts N copies of itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set
(subprocess.Popen -- never an exec from a process that has initialised the GPU), relays rank 0's JSON line and exits with
the worst child status.

Workload (BASELINE.json configs[1]): Benes--Bernoulli 1-D, N = 15, T = 1000, B = 4096 replicates PER GPU (weak
scaling: replicates are independent, sharded with no data-path collective; the per-replicate NLL vector is
all-gathered once per pass), central moments, TME-3, fp64, synthetic measurements.  A "step" is one pass of the hot
path over the whole batch (B x T filter time-steps), inputs already resident in HBM.

`value` counts LIVE filter time-steps only: at N = 15, T = 1000 a large share of replicates NaN-poison part-way (the
Hankel matrix loses positive definiteness -- the reference does the same, SURVEY.md section 7 hard part 3) and the
kernel stops computing for a poisoned replicate, so counting B x T would credit skipped work.  The nominal B x T rate
is reported next to it as `nominal_value`.

Beside the contract's fields the line carries (rank 0, N = 1 only): `cpu_baseline` (the oracle's C port on the host
cores, with BASELINE.json's "max |moment err|" per quantity and the first-NaN agreement), `end_to_end_ms` (the
reference-shaped Python API moment_filter_cms on the same workload, host pointers in and out) and `other_workloads`
(short runs of BASELINE configs 3, 4 and 5).

Multi-rank runs: exit status 3 (mfs_amd.dist.EXIT_RCCL_FAILED) and `nll_allgather_ok: false` when the NLL all-gather did
not go through RCCL, unless --allow-host-gather is given; the fallback gather is never inside the timed region.
"""
import argparse
import glob
import json
import os
import re
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X vector fp64: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz
SIMDS = 1024
PEAK_CLOCK_GHZ = 2.4

WORKLOADS = {
    # name: (model, N, T, B per GPU, mode, transition)
    'benes_bernoulli_N15_T1000_B4096_central_tme3': ('benes', 15, 1000, 4096, 'central', 'tme_3'),
    'benes_bernoulli_N7_T100_B4096_central_tme3': ('benes', 7, 100, 4096, 'central', 'tme_3'),
    'benes_bernoulli_N15_T1000_B4096_central_tme_normal3': ('benes', 15, 1000, 4096, 'central', 'tme_normal_3'),
    # SURVEY section 8d asks for the raw (and scaled) representation of config 2 next to the central one, with divergence counts
    'benes_bernoulli_N15_T1000_B4096_raw_tme3': ('benes', 15, 1000, 4096, 'raw', 'tme_3'),
    'benes_bernoulli_N15_T1000_B4096_scaled_tme3': ('benes', 15, 1000, 4096, 'scaled', 'tme_3'),
    # config 2 with stable=True (mfs/utils.py:525-538: LDL^T completion where a pivot is negative) -- SURVEY 8f rank 2
    'benes_bernoulli_N15_T1000_B4096_central_tme3_stable': ('benes', 15, 1000, 4096, 'central', 'tme_3', {'stable': 1}),
    # ... and with the predict-half rule reconstructed from the posterior moments as the reference does (MFS_PREDICT_RULE=recompute)
    'benes_bernoulli_N15_T1000_B4096_central_tme3_recompute': ('benes', 15, 1000, 4096, 'central', 'tme_3', {'recompute': 1}),
    'well_poisson_N7_T1000_B131072_central_tme_normal2': ('well', 7, 1000, 131072, 'central', 'tme_normal_2'),
    # BASELINE config 3 (dardel/convergence): OU / Gaussian, exact linear-Gaussian transition
    'ou_gaussian_N15_T1000_B1024_central': ('ou', 15, 1000, 1024, 'central', 'exact'),
    'ou_gaussian_N25_T1000_B1024_central': ('ou', 25, 1000, 1024, 'central', 'exact'),
    'ou_gaussian_N15_T1000_B1024_central_stable': ('ou', 15, 1000, 1024, 'central', 'exact', {'stable': 1}),
    # BASELINE config 5 (d = 2): per-GPU shard of the 512-replicate batch is set with --B (128 on 4 GPUs)
    'prey_predator_N6_T500_B512_central_tme2': ('prey', 6, 500, 512, 'central', 'tme_2'),
    'prey_predator_N6_T500_B512_central_tme_normal2': ('prey', 6, 500, 512, 'central', 'tme_normal_2'),
    # ... with stable=True (quadratures.py:154 `ldl=True`: the LDL^T completion of mfs/utils.py:495-538 in every rule)
    'prey_predator_N6_T500_B512_central_tme2_stable': ('prey', 6, 500, 512, 'central', 'tme_2', {'stable': 1}),
    # the order the reference's CPU script runs (dardel/run_prey_predator_mf.sh:29-30: --N=5, tme_2 and tme_normal_2)
    'prey_predator_N5_T500_B512_central_tme2': ('prey', 5, 500, 512, 'central', 'tme_2'),
    'prey_predator_N5_T500_B512_central_tme_normal2': ('prey', 5, 500, 512, 'central', 'tme_normal_2'),
}
DEFAULT_WORKLOAD = 'benes_bernoulli_N15_T1000_B4096_central_tme3'
# (workload, B override, label): short runs reported under `other_workloads` by the default single-GPU run
OTHER_WORKLOADS = [
    ('benes_bernoulli_N7_T100_B4096_central_tme3', 0, 'config1_size_B4096'),
    # config 2 in the other two representations (with the divergence counts) and with the Normal closure
    ('benes_bernoulli_N15_T1000_B4096_raw_tme3', 0, 'config2_raw'),
    ('benes_bernoulli_N15_T1000_B4096_scaled_tme3', 0, 'config2_scaled'),
    ('benes_bernoulli_N15_T1000_B4096_central_tme_normal3', 0, 'config2_tme_normal3'),
    ('benes_bernoulli_N15_T1000_B4096_central_tme3_stable', 0, 'config2_stable'),
    ('benes_bernoulli_N15_T1000_B4096_central_tme3_recompute', 0, 'config2_recompute_rule'),
    ('ou_gaussian_N15_T1000_B1024_central', 0, 'config3_N15'),
    ('ou_gaussian_N25_T1000_B1024_central', 0, 'config3_N25'),
    ('well_poisson_N7_T1000_B131072_central_tme_normal2', 0, 'config4_shard_B131072'),
    ('prey_predator_N6_T500_B512_central_tme2', 0, 'config5_B512'),
    ('prey_predator_N6_T500_B512_central_tme2', 128, 'config5_4gpu_shard_B128'),
    ('prey_predator_N6_T500_B512_central_tme_normal2', 0, 'config5_tme_normal2'),
    # the N-D kernel's throughput ceiling (two workgroups per CU: four rounds of 512) and the order the reference's CPU script runs
    ('prey_predator_N6_T500_B512_central_tme2_stable', 0, 'config5_stable'),
    ('prey_predator_N6_T500_B512_central_tme2', 2048, 'config5_B2048_throughput'),
    ('prey_predator_N5_T500_B512_central_tme2', 0, 'config5_N5_tme2'),
    ('prey_predator_N5_T500_B512_central_tme_normal2', 0, 'config5_N5_tme_normal2'),
]


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
    p.add_argument('--no-end-to-end', action='store_true', help='skip the reference-shaped host API timing')
    p.add_argument('--no-other-workloads', action='store_true', help='skip the short runs of configs 3 / 4 / 5')
    p.add_argument('--extra-kernels', action='store_true',
                   help='run only the gradient and characteristic-function entries of `other_workloads` and print them '
                        '(what tools/profile.sh puts under rocprofv3 for those two kernels)')
    p.add_argument('--allow-host-gather', action='store_true',
                   help='multi-rank: exit 0 even if the NLL all-gather fell back from RCCL to the host route')
    p.add_argument('--dry-run', action='store_true',
                   help='rendezvous, communicator set-up and the JSON line only: no workload, no kernel (launcher rehearsal)')
    p.add_argument('--launch-timeout', type=float, default=1500., help='self-launched runs: seconds before the children are stopped')
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# workloads
# ---------------------------------------------------------------------------------------------------------------------
class Workload1D:
    """Model closures, synthetic measurements and device-resident buffers of one 1-D workload on this rank."""

    def __init__(self, name, B, T, rank, device, fast_data=False):
        from mfs_amd import _lib, synth, stats
        from mfs_amd.one_dim import filtering, moments, ss_models
        self.name = name
        self.model, self.N, T0, B0, self.mode, self.transition = WORKLOADS[name][:6]
        self.flags = WORKLOADS[name][6] if len(WORKLOADS[name]) > 6 else {}
        self.stable = int(self.flags.get('stable', 0))
        self.B, self.T = B or B0, T or T0
        self.full_size = (self.B == B0 and self.T == T0)
        N, B, T = self.N, self.B, self.T
        rng = np.random.default_rng(1234 + rank)
        self.theta = None
        sub = 2 if fast_data else 10    # Euler sub-steps of the synthetic paths (bench-only data: cheaper generation)
        if self.model == 'benes':
            dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.benes_bernoulli(N)
            self.ys, _ = synth.benes_bernoulli_batch(B, T, dt, seed=100 + rank, substeps=sub)
        elif self.model == 'well':
            dt, _, _, ic, drift0, dispersion, _, pmf0, _ = ss_models.well_poisson(3., N)
            # parameter grid of BASELINE config 4 (theta in [0.5, 6]^2), one point per replicate
            p1, p2 = rng.uniform(0.5, 6., size=B), rng.uniform(0.5, 6., size=B)
            self.theta = (p1, p2)
            drift, pmf = (lambda x: drift0(x, p1)), (lambda y, x: pmf0(y, x, p2))
            nb = min(B, 4096) if fast_data else B    # short runs: 4096 measurement sets tiled over the theta grid
            ys, _ = synth.well_poisson_batch(nb, T, p1=3., p2=3., dt=dt, seed=100 + rank, substeps=sub)
            self.ys = np.ascontiguousarray(np.tile(ys, (-(-B // nb), 1))[:B])
        else:   # 'ou': dardel/convergence/convergence_mf.py:32-61 -- dt = 0.1, ell = 1, sigma = 0.5, R = 1, x0 ~ N(0, sigma^2)
            import math
            from mfs_amd.utils import GaussianSum1D
            dt_ou, ell, sigma = 0.1, 1., 0.5
            F, Sigma = math.exp(-dt_ou / ell), sigma ** 2 * (1 - math.exp(-2 * dt_ou / ell))
            self.ys, _ = synth.ou_gaussian_batch(B, T, dt=dt_ou, ell=ell, sigma=sigma, seed=100 + rank)
            ic = GaussianSum1D.new(means=[0.], variances=[sigma ** 2], weights=[1.], N=N)
            pmf = lambda y, x: stats.norm_pdf(y, x, 1.)   # noqa: E731
        if self.model == 'ou':
            fns = moments.sde_cond_moments_normal(lambda x: F * x, lambda x: Sigma)
        else:
            kind, order = self.transition.rsplit('_', 1)
            if kind == 'tme':
                fns = moments.sde_cond_moments_tme(drift, dispersion, dt, int(order))
            else:
                fns = moments.sde_cond_moments_tme_normal(drift, dispersion, dt, int(order), N)
        self.ic, self.fns, self.pmf = ic, fns, pmf
        mean_fn = fns[3] if self.mode == 'central' else fns[4] if self.mode == 'scaled' else None
        self.tables, self.lik = filtering.trace_model(self.mode, fns[{'raw': 0, 'central': 1, 'scaled': 2}[self.mode]],
                                                      mean_fn, pmf)
        self.mstruct, self._keep = filtering.build_model_struct(self.tables, self.lik, B)
        self.m0 = {'raw': ic.rms, 'central': ic.cms, 'scaled': ic.scms}[self.mode]
        self.device = device
        self.z = 2 * N

    def upload(self, want_moments, chunk):
        import ctypes as C
        from mfs_amd import _lib
        L = _lib.lib()
        B, T, N = self.B, self.T, self.N
        self.d_m0 = _lib.DeviceBuffer.from_array(self.m0)
        self.d_mean0 = _lib.DeviceBuffer.from_array(np.array([self.ic.mean], dtype=np.float64))
        self.d_scale0 = _lib.DeviceBuffer.from_array(np.array([np.sqrt(self.ic.variance)]))
        self.d_ys = _lib.DeviceBuffer.from_array(self.ys)
        self.d_mom = _lib.DeviceBuffer(B * T * 2 * N * 8) if want_moments else None
        self.d_means = _lib.DeviceBuffer(B * T * 8)
        self.d_scales = _lib.DeviceBuffer(B * T * 8) if self.mode == 'scaled' else None
        self.d_nell = _lib.DeviceBuffer(B * 8)
        self.d_fn = _lib.DeviceBuffer(B * 4)
        self.plan = C.c_void_p()
        _lib.check(L.mfs_plan_1d_create(C.byref(self.plan), C.byref(self.mstruct), _lib.MODE[self.mode], N, T, B,
                                        self.stable, chunk, self.device))
        geo = [C.c_int() for _ in range(4)]
        _lib.check(L.mfs_plan_1d_geometry(self.plan, *[C.byref(g) for g in geo]))
        self.geometry = {'lanes_per_filter': geo[0].value, 'filters_per_block': geo[1].value, 'grid': geo[2].value,
                         'lds_bytes_per_block': geo[3].value}
        self.kernel = 'mfs::filter1d_fast_kernel'

    def launch(self, stream):
        from mfs_amd import _lib
        if self.flags.get('recompute'):      # read by the library at every run (capi.hip: A/B switch of the predict-half rule)
            os.environ['MFS_PREDICT_RULE'] = 'recompute'
        try:
            _lib.check(_lib.lib().mfs_plan_1d_run(self.plan, self.d_m0.ptr, 0, self.d_mean0.ptr, self.d_scale0.ptr,
                                                  self.d_ys.ptr, self.d_mom.ptr if self.d_mom else None, self.d_means.ptr,
                                                  self.d_scales.ptr if self.d_scales else None, self.d_nell.ptr,
                                                  self.d_fn.ptr, stream))
        finally:
            if self.flags.get('recompute'):
                os.environ.pop('MFS_PREDICT_RULE', None)

    def release(self):
        from mfs_amd import _lib
        _lib.check(_lib.lib().mfs_plan_1d_destroy(self.plan))
        for b in (self.d_m0, self.d_mean0, self.d_scale0, self.d_ys, self.d_mom, self.d_means, self.d_scales,
                  self.d_nell, self.d_fn):
            if b is not None:
                b.free()

    def algorithmic_bytes(self, live_steps, want_moments):
        # per live filter-step one y in, 2N moments + mean (+ scale) out; per filter 2N + 1
        extra = {'raw': 1, 'central': 2, 'scaled': 3}[self.mode]
        per_step = 8 * (2 * self.N + extra) if want_moments else 8 * 2
        return live_steps * per_step + self.B * (8 * 2 * self.N + 8)

    def host_api_call(self):
        """The reference-shaped entry point (mfs/one_dim/filtering.py:92-98 signature), host arrays in and out."""
        from mfs_amd.one_dim import filtering
        ic, f = self.ic, self.fns
        if self.mode == 'raw':
            return filtering.moment_filter_rms(f[0], self.pmf, ic.rms, self.ys, device=self.device)
        if self.mode == 'central':
            return filtering.moment_filter_cms(f[1], f[3], self.pmf, ic.cms, ic.mean, self.ys, device=self.device)
        return filtering.moment_filter_scms(f[2], f[4], self.pmf, ic.scms, ic.mean, float(np.sqrt(ic.variance)),
                                            self.ys, device=self.device)


class WorkloadND:
    """BASELINE config 5: the d = 2 prey--predator filter through the device-pointer N-D plan."""

    def __init__(self, name, B, T, rank, device, fast_data=False):
        from mfs_amd import synth
        from mfs_amd.multi_dims import filtering, moments, ss_models
        from mfs_amd.multi_dims.multi_indices import generate_graded_lexico_multi_indices, \
            gram_and_hankel_indices_graded_lexico
        self.name = name
        self.model, self.N, T0, B0, self.mode, self.transition = WORKLOADS[name][:6]
        self.flags = WORKLOADS[name][6] if len(WORKLOADS[name]) > 6 else {}
        self.stable = int(self.flags.get('stable', 0))
        self.B, self.T = B or B0, T or T0
        self.full_size = (self.B == B0 and self.T == T0)
        N = self.N
        self.mi = generate_graded_lexico_multi_indices(2, 2 * N - 1)
        self.inds = gram_and_hankel_indices_graded_lexico(N, 2)
        self.z = self.mi.shape[0]
        dt, _, _, gs, drift, disp, _, pmf, _ = ss_models.prey_predator(self.mi)
        kind, order = self.transition.rsplit('_', 1)
        if kind == 'tme':
            self.fns, self.sig = moments.sde_cond_moments_tme(drift, disp, dt, int(order)), 'multi-index'
        else:
            self.fns, self.sig = moments.sde_cond_moments_tme_normal(drift, disp, dt, int(order), self.mi), 'index'
        self.ys, _ = synth.prey_predator_batch(self.B, self.T, dt, seed=100 + rank, substeps=4 if fast_data else 20)
        self.gs, self.pmf, self.device = gs, pmf, device
        self.filtering = filtering
        self.theta = None

    def upload(self, want_moments, chunk):
        import ctypes as C
        from mfs_amd import _lib
        L = _lib.lib()
        f, mode, gs, mi, B, T, z = self.filtering, self.mode, self.gs, self.mi, self.B, self.T, self.z
        tables = f._trace_transition((self.fns[{'raw': 0, 'central': 1, 'scaled': 2}[mode]], self.sig), mode,
                                     (mi, self.inds))
        lik = f._trace_likelihood(self.pmf, 2)
        self.mstruct, self._keep = f._model_struct(tables, lik)
        scale0 = np.sqrt(np.array([gs.cms[5], gs.cms[3]]))
        m0 = {'raw': gs.rms, 'central': gs.cms, 'scaled': gs.cms / np.prod(scale0 ** mi, axis=-1)}[mode]
        self.d_m0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(m0))
        self.d_mean0 = _lib.DeviceBuffer.from_array(np.ascontiguousarray(gs.mean, dtype=np.float64))
        self.d_scale0 = _lib.DeviceBuffer.from_array(scale0)
        self.d_ys = _lib.DeviceBuffer.from_array(self.ys)
        self.d_mom = _lib.DeviceBuffer(B * T * z * 8) if want_moments else None
        self.d_means = _lib.DeviceBuffer(B * T * 2 * 8)
        self.d_scales = _lib.DeviceBuffer(B * T * 2 * 8)
        self.d_nell = _lib.DeviceBuffer(B * 8)
        self.d_fn = _lib.DeviceBuffer(B * 4)
        self._mi32 = np.ascontiguousarray(mi, dtype=np.int32)
        self._inds32 = np.ascontiguousarray(self.inds, dtype=np.int32)
        self.plan = C.c_void_p()
        _lib.check(L.mfs_plan_nd_create(C.byref(self.plan), C.byref(self.mstruct), _lib.MODE[mode], self.N, T, B, z,
                                        _lib.ptr(self._mi32), _lib.ptr(self._inds32), self.stable, self.device))
        geo = [C.c_int() for _ in range(3)]
        _lib.check(L.mfs_plan_nd_geometry(self.plan, *[C.byref(g) for g in geo]))
        self.geometry = {'threads_per_filter': geo[0].value, 'grid': geo[1].value, 'lds_bytes_per_block': geo[2].value,
                         'd': 2, 'z': int(z)}
        self.kernel = 'mfs::filternd_kernel'

    def launch(self, stream):
        from mfs_amd import _lib
        _lib.check(_lib.lib().mfs_plan_nd_run(self.plan, self.d_m0.ptr, 0, self.d_mean0.ptr, self.d_scale0.ptr,
                                              self.d_ys.ptr, self.d_mom.ptr if self.d_mom else None, self.d_means.ptr,
                                              self.d_scales.ptr, self.d_nell.ptr, self.d_fn.ptr, stream))

    def release(self):
        from mfs_amd import _lib
        _lib.check(_lib.lib().mfs_plan_nd_destroy(self.plan))
        for b in (self.d_m0, self.d_mean0, self.d_scale0, self.d_ys, self.d_mom, self.d_means, self.d_scales,
                  self.d_nell, self.d_fn):
            if b is not None:
                b.free()

    def algorithmic_bytes(self, live_steps, want_moments):
        extra = {'raw': 1, 'central': 3, 'scaled': 5}[self.mode]
        per_step = 8 * (self.z + extra) if want_moments else 8 * 2
        return live_steps * per_step + self.B * 8 * (self.z + 3)

    def host_api_call(self):
        f, gs = self.filtering, self.gs
        if self.mode == 'central':
            return f.moment_filter_nd_cms((self.fns[1], self.sig), self.fns[3], self.pmf, self.ys, (self.mi, self.inds),
                                          gs.cms, gs.mean, device=self.device)
        return f.moment_filter_nd_rms((self.fns[0], self.sig), self.pmf, self.ys, (self.mi, self.inds), gs.rms,
                                      device=self.device)


def make_workload(name, B, T, rank, device, fast_data=False):
    cls = WorkloadND if WORKLOADS[name][0] == 'prey' else Workload1D
    return cls(name, B, T, rank, device, fast_data)


# ---------------------------------------------------------------------------------------------------------------------
# the timed region
# ---------------------------------------------------------------------------------------------------------------------
def timed_passes(w, comm, steps, warmup, want_moments, chunk):
    """W untimed + K timed passes of the hot path, barrier + device sync on both sides, max over ranks; the NLL
    all-gather rides on the launch stream (RCCL) -- a host-route fallback is done once, after the clock has stopped."""
    import ctypes as C
    from mfs_amd import _lib
    L = _lib.lib()
    w.upload(want_moments, chunk)
    d_nell_all = _lib.DeviceBuffer(w.B * 8 * comm.world)
    stream = C.c_void_p()
    _lib.check(L.mfs_stream_create(C.byref(stream)))
    ev = [C.c_void_p() for _ in range(2 * max(steps, 1))]
    for e in ev:
        _lib.check(L.mfs_event_create(C.byref(e)))
    gather_in_loop = not comm.degraded

    def one_pass(i=None):
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i], stream))
        w.launch(stream)
        if i is not None:
            _lib.check(L.mfs_event_record(ev[2 * i + 1], stream))
        if gather_in_loop:
            comm.allgather_nell(w.d_nell, d_nell_all, w.B, stream)  # RCCL over xGMI when world > 1; device copy otherwise

    for _ in range(warmup):
        one_pass()
    _lib.check(L.mfs_stream_synchronize(stream))
    comm.barrier()
    _lib.check(L.mfs_device_synchronize())
    t0 = time.perf_counter()
    for i in range(steps):
        one_pass(i)
    _lib.check(L.mfs_stream_synchronize(stream))
    _lib.check(L.mfs_device_synchronize())
    comm.barrier()
    elapsed = comm.max_over_ranks(time.perf_counter() - t0)
    if not gather_in_loop:
        comm.allgather_nell(w.d_nell, d_nell_all, w.B, stream)      # host route (reported as a failure), untimed
        _lib.check(L.mfs_stream_synchronize(stream))

    kern_ms = []
    for i in range(steps):
        ms = C.c_float()
        _lib.check(L.mfs_event_elapsed_ms(ev[2 * i], ev[2 * i + 1], C.byref(ms)))
        kern_ms.append(ms.value)
    first_nan = w.d_fn.to_array((w.B,), np.int32)
    nell = w.d_nell.to_array((w.B,))
    nell_all = d_nell_all.to_array((comm.world * w.B,))
    res = {'elapsed': elapsed, 'kern_ms': kern_ms, 'first_nan': first_nan, 'nell': nell,
           'live_steps': int(np.where(first_nan >= 0, first_nan + 1, w.T).sum()),
           'alive': int((first_nan < 0).sum()),
           'gather_matches': bool(np.array_equal(nell_all[comm.rank * w.B:(comm.rank + 1) * w.B], nell, equal_nan=True))}
    for e in ev:
        _lib.check(L.mfs_event_destroy(e))
    _lib.check(L.mfs_stream_destroy(stream))
    d_nell_all.free()
    return res


def natural(path):   # r01_v9 before r01_v10, r01 before r02
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', path)]


def recorded_pmc(workload, moments_streamed=True):
    """The newest committed counter summary of this workload (counters cannot be read from inside the process; they are
    collected with rocprofv3 --pmc in separate passes, condensed by tools/pmc_summary.py and committed under profiles/)."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*', 'pmc*.json')), key=natural):
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get('workload') == workload and moments_streamed and rec.get('B_override') in (None, 0):
            best = dict(rec, source=os.path.relpath(path, ROOT))
    return best


def executed_work(pmc, T, kern_ms):
    """Utilisation of the vector ALU from the committed counters (per launch) and this run's launch time.  What the
    kernel EXECUTED, not a flop model of the reference's dense algorithm."""
    if not pmc or 'sq' not in pmc:
        return None
    sq = pmc['sq']
    out = {'source': pmc['source']}
    if 'SQ_INSTS_VALU' in sq and 'SQ_WAVE_CYCLES' in sq:
        # SQ_WAVE_CYCLES counts quad-cycles (MI355X_MICROARCH.md), a wave64 VALU instruction occupies the SIMD for 4 clocks
        out['valu_issue_occupancy_per_wave'] = sq['SQ_INSTS_VALU'] / sq['SQ_WAVE_CYCLES']
        out['valu_insts_per_wave_step'] = sq['SQ_INSTS_VALU'] / sq.get('SQ_WAVES', 1.) / T
        # chip level: wave-instructions per second against SIMDS x clock / 4
        rate = sq['SQ_INSTS_VALU'] / (kern_ms * 1e-3)
        out['valu_wave_insts_per_s'] = rate
        out['valu_issue_fraction_of_chip_peak'] = rate / (SIMDS * PEAK_CLOCK_GHZ * 1e9 / 4)
    f64 = {k: sq[k] for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64',
                              'SQ_INSTS_VALU_TRANS_F64') if k in sq}
    if 'SQ_INSTS_VALU_FMA_F64' in f64:
        flops = 64. * (2 * f64['SQ_INSTS_VALU_FMA_F64'] + f64.get('SQ_INSTS_VALU_MUL_F64', 0.) +
                       f64.get('SQ_INSTS_VALU_ADD_F64', 0.) + f64.get('SQ_INSTS_VALU_TRANS_F64', 0.))
        out['fp64_insts'] = f64
        out['fp64_tflops_executed'] = flops / (kern_ms * 1e-3) / 1e12      # 64 lanes per wave-instruction, idle lanes included
        out['fp64_fraction_of_vector_peak'] = out['fp64_tflops_executed'] / FP64_VALU_PEAK_TFLOPS
    if 'wait_fraction_of_wave_cycles' in pmc:
        out['wait_fraction_of_wave_cycles'] = pmc['wait_fraction_of_wave_cycles']
    return out


def summarise(w, res, steps, world, want_moments, live_total=None):
    kern_ms = float(np.mean(res['kern_ms'])) if res['kern_ms'] else float('nan')
    live_total = res['live_steps'] if live_total is None else live_total
    algo = w.algorithmic_bytes(res['live_steps'], want_moments)
    return {'ms_per_step': res['elapsed'] / steps * 1e3, 'kernel_ms': kern_ms,
            'value': live_total * steps / res['elapsed'], 'nominal_value': world * w.B * w.T * steps / res['elapsed'],
            'live_fraction': live_total / (world * w.B * w.T), 'algorithmic_bytes_per_launch': algo,
            'hbm_gbs': algo / (kern_ms * 1e-3) / 1e9}



def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: one child process per GPU, this process only supervises.  It must not
    initialise the GPU (the children own the devices, and a process that has touched the GPU must never exec), so nothing
    of mfs_amd is imported here."""
    import socket
    import subprocess
    import threading
    n = args.gpus
    with socket.socket() as sock:            # a free rendezvous port on the loopback interface
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL across processes needs it on this driver
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))

    def relay():                             # rank 0 prints the one JSON line; library chatter (RCCL writes its
        for raw in procs[0].stdout:          # warnings to stdout) goes to stderr so that stdout stays one line
            line = raw.decode(errors='replace')
            out = sys.stdout if line.lstrip().startswith('{"metric"') else sys.stderr
            out.write(line)
            out.flush()

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    deadline = time.time() + args.launch_timeout
    grace = None                             # once a rank has failed the others get a minute to notice and leave
    while any(p.poll() is None for p in procs):
        now = time.time()
        if grace is None and any(p.poll() not in (None, 0) for p in procs):
            grace = now + 60.
        if now > deadline or (grace is not None and now > grace):
            for p in procs:                  # exactly the processes started above
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    th.join(10.)
    worst = 0
    for c in codes:
        c = 128 - c if c < 0 else c          # killed by a signal
        worst = max(worst, c)
    if worst:
        print(f'bench.py launcher: child exit statuses {codes}', file=sys.stderr)
    return worst


def dry_run(args, comm, rank, world):
    """--dry-run: everything up to the communicator (rendezvous, RCCL initialisation or its reported failure), then the line."""
    exit_code = comm.exit_status(True, args.allow_host_gather)
    if rank == 0:
        print(json.dumps({
            'metric': 'filter time-steps/sec', 'value': None, 'unit': 'filter-steps/s', 'n_gpus': world, 'dry_run': True,
            'steps': args.steps, 'warmup': args.warmup, 'config': {'workload': args.workload},
            'nll_allgather_ok': not comm.degraded,
            'nll_allgather': ('rccl ncclAllGather' if comm.data == 'rccl' and world > 1 else
                              'single rank' if world == 1 else f'FAILED: {comm.rccl_error}')}), flush=True)
    comm.close(exit_code)
    sys.exit(exit_code)


# ---------------------------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))         # launcher: nothing below runs in this process
    if world != args.gpus:
        args.gpus = world

    from mfs_amd import _lib, dist
    L = _lib.lib()
    # one GPU per rank; on a box with fewer GPUs than ranks (rehearsals only) ranks share devices and RCCL, which
    # refuses duplicate GPUs, fails: reported, and the exit status says so
    try:
        n_dev = _lib.device_count()
    except _lib.MfsError:
        if not args.dry_run:                 # no device, no measurement: fail loudly
            raise
        n_dev = 0                            # launcher rehearsal on a box without a GPU
    device = local_rank % max(n_dev, 1)
    comm = dist.Communicator.from_env(device=device)  # TCP control plane (mfs_amd/rdzv.py) + RCCL for the NLL gather
    if args.dry_run:
        dry_run(args, comm, rank, world)
    _lib.check(L.mfs_set_device(device))
    if args.extra_kernels:
        if rank == 0:
            print(json.dumps(extra_kernels(device)), flush=True)
        comm.close(0)
        sys.exit(0)
    want_moments = not args.no_moments

    w = make_workload(args.workload, args.B, args.T, rank, device)
    res = timed_passes(w, comm, args.steps, args.warmup, want_moments, args.chunk)
    live_total = comm.sum_over_ranks(res['live_steps'])
    alive_total = comm.sum_over_ranks(res['alive'])
    gather_ok = all(comm._allgather_obj(bool(res['gather_matches']))) and not comm.degraded
    exit_code = 0

    if rank == 0:
        s = summarise(w, res, args.steps, world, want_moments, live_total)
        pmc = recorded_pmc(args.workload, want_moments) if w.full_size else None
        traffic = pmc.get('hbm_bytes_per_launch') if pmc else None
        work = executed_work(pmc, w.T, s['kernel_ms'])
        hbm = {'achieved': s['hbm_gbs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': s['hbm_gbs'] / HBM_PEAK_GBS,
               'algorithmic_bytes_per_launch': s['algorithmic_bytes_per_launch'],
               'note': 'reported because BASELINE.json asks for the HBM fraction; ~1 % by construction (SURVEY 8d)'}
        roofline = {'kernel': w.kernel, 'avg_launch_ms': s['kernel_ms'], 'traffic': traffic,
                    'traffic_source': (pmc['source'] + ' (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, '
                                       'FETCH_SIZE x2 on gfx950)') if traffic else None,
                    'algorithmic_bytes_per_launch': s['algorithmic_bytes_per_launch'], 'hbm': hbm}
        if work and 'fp64_tflops_executed' in work:
            # the bound of this path is the fp64 vector ALU (issue + dependent-instruction latency), not HBM, not MFMA
            roofline.update({'bound': 'valu_fp64', 'achieved': work['fp64_tflops_executed'],
                             'peak': FP64_VALU_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                             'frac': work['fp64_fraction_of_vector_peak'],
                             'basis': 'fp64 flops EXECUTED per launch (SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 x 64 lanes, '
                                      'FMA = 2) from the committed counters / this run\'s launch time'})
        else:
            roofline.update({'bound': 'hbm', 'achieved': hbm['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': hbm['frac'],
                             'note': 'no fp64 instruction counters recorded for this workload: HBM roofline only; the '
                                     'kernel is bound by fp64 VALU issue + dependency latency'})
        roofline['valu'] = work
        out = {
            'metric': 'filter time-steps/sec', 'value': s['value'], 'unit': 'filter-steps/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': s['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': dict({'workload': args.workload, 'model': None, 'N': w.N, 'T': w.T, 'replicates_per_gpu': w.B,
                            'mode': w.mode, 'transition': w.transition, 'parallelism': f'replicate-sharded x{world}',
                            'chunk': args.chunk or w.T, 'moments_streamed_out': want_moments}, **w.geometry),
            'nominal_value': s['nominal_value'], 'live_fraction': s['live_fraction'],
            'replicates_alive_at_T': alive_total, 'replicates': world * w.B,
            'nll_allgather_ok': bool(gather_ok),
            'nll_allgather': ('rccl ncclAllGather' if comm.data == 'rccl' and world > 1 else
                              'single rank: device copy' if world == 1 else
                              f'FAILED, host fallback outside the timed region: {comm.rccl_error}'),
            'target_1e6_steps_per_s_met': bool(s['value'] >= 1e6 * world),
            'roofline': roofline,
        }
        if world == 1:
            if not args.no_cpu_baseline:
                try:
                    out['cpu_baseline'] = (cpu_baseline_nd(args, w, res) if isinstance(w, WorkloadND)
                                           else cpu_baseline(args, w, res))
                except Exception as e:   # noqa: BLE001 -- a failing checker must not lose the measurement
                    out['cpu_baseline'] = {'error': repr(e)}
            if not args.no_end_to_end:
                out.update(end_to_end(w))
        w.release()
        if world == 1 and not args.no_other_workloads and args.workload == DEFAULT_WORKLOAD and w.full_size:
            out['other_workloads'] = other_workloads(comm, device)
        print(json.dumps(out), flush=True)
    else:
        w.release()
    exit_code = comm.exit_status(gather_ok, args.allow_host_gather)
    comm.close(exit_code)
    sys.exit(exit_code)


def end_to_end(w):
    """Wall time of the reference-shaped Python entry point on the same workload: trace the callables, stage the host
    arrays through the library's pool, run, copy every output back (PCIe-inclusive; never `value`)."""
    from mfs_amd import _lib
    t0 = time.perf_counter()
    out = w.host_api_call()
    first = time.perf_counter() - t0
    del out
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        out = w.host_api_call()
        times.append(time.perf_counter() - t0)
        del out
    ms = float(np.median(times)) * 1e3
    stats = _lib.pool_stats(w.device)
    _lib.check(_lib.lib().mfs_pool_trim(w.device))
    return {'end_to_end_ms': ms, 'end_to_end': {
        'api': 'mfs_amd.*.filtering.moment_filter_* (host arrays in, pinned NumPy arrays out)', 'median_of': 3,
        'first_call_ms': first * 1e3, 'all_ms': [t * 1e3 for t in times],
        'nominal_steps_per_s': w.B * w.T / (ms * 1e-3),
        'output_bytes': int(w.B * w.T * (w.z + 1) * 8), 'pool': stats}}


def extra_kernels(device):
    """The two kernels beside the filters through their host-pointer entry points -- wall time of
    the call (trace, H2D, kernel, D2H), median of three after a warm-up: PCIe-inclusive, never `value`.  Their kernel-only
    times are in the rocprofv3 traces under profiles/."""
    from mfs_amd import synth, estimation
    from mfs_amd.one_dim import moments, ss_models
    out = {}
    try:   # forward-mode NLL gradient: well--Poisson N = 7, TME-normal-2, P = 2, one theta point per replicate (config 4's model)
        N, T, B = 7, 1000, 16384
        dt, _, _, ic, drift, dispersion, _, pmf, _ = ss_models.well_poisson(3., N)
        rng = np.random.default_rng(7)
        th = rng.uniform(0.5, 6., size=(B, 2))
        ys, _ = synth.well_poisson_batch(256, T, p1=3., p2=3., dt=dt, seed=100, substeps=2)
        ys = np.ascontiguousarray(np.tile(ys, (B // 256, 1)))

        def model(P):
            _, c, _, mu, _ = moments.sde_cond_moments_tme_normal(lambda x: drift(x, P[:, 0]), dispersion, dt, 2, N)
            return c, mu, (lambda y, x: pmf(y, x, P[:, 1]))

        def call():
            return estimation.nell_and_grad_forward(model, th, ic.cms, ic.mean, ys)
        call()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            nell, grad = call()
            ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        out['gradient_N7_P2'] = {'workload': 'well_poisson_N7_T1000 central tme_normal_2, d NLL / d theta in the time loop, P = 2',
                                 'kernel': 'mfs::filter1d_grad_kernel', 'filters': B, 'T': T, 'call_ms': ms,
                                 'filters_per_s': B / (ms * 1e-3), 'filter_steps_per_s': B * T / (ms * 1e-3),
                                 'finite': int(np.isfinite(nell).sum()), 'timing': 'wall time of the host-pointer call'}
    except Exception as e:   # noqa: BLE001
        out['gradient_N7_P2'] = {'error': repr(e)}
    try:   # characteristic function of the filtering distributions (post_processing_mf.py:37-60): N = 15, 2000 z points
        N, count, nz = 15, 8192, 2000
        rng = np.random.default_rng(8)
        mus, vs = rng.normal(scale=0.5, size=count), rng.uniform(0.3, 1.2, size=count)
        from mfs_amd.one_dim.moments import central_moment_of_normal
        base = np.array([central_moment_of_normal(1., p) for p in range(2 * N)])
        cms = base[None, :] * np.sqrt(vs)[:, None] ** np.arange(2 * N)[None, :]
        zs = np.linspace(-2., 2., nz)
        moments.characteristic_fn(zs, cms[:64], mus[:64])
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            cf = moments.characteristic_fn(zs, cms, mus)
            ts.append(time.perf_counter() - t0)
        ms = float(np.median(ts)) * 1e3
        out['characteristic_fn_N15'] = {'kernel': 'mfs::cf1d_fast_kernel', 'moment_vectors': count, 'z_points': nz, 'call_ms': ms,
                                        'vector_z_points_per_s': count * nz / (ms * 1e-3), 'output_bytes': int(cf.nbytes),
                                        'timing': 'wall time of the host-pointer call (D2H of the complex128 grid included)'}
        del cf
    except Exception as e:   # noqa: BLE001
        out['characteristic_fn_N15'] = {'error': repr(e)}
    return out


def other_workloads(comm, device):
    """Short runs (1 warm-up + 2 timed passes) of the other BASELINE configurations, kernel time from HIP events."""
    out = {}
    for name, B, label in OTHER_WORKLOADS:
        try:
            w = make_workload(name, B, 0, 0, device, fast_data=(label != 'config2_scaled'))   # (the exact-arithmetic fixture holds the default data)
            res = timed_passes(w, comm, 2, 1, True, 0)
            s = summarise(w, res, 2, 1, True)
            out[label] = {'workload': name, 'replicates': w.B, 'T': w.T, 'N': w.N, 'kernel_ms': s['kernel_ms'],
                          'ms_per_step': s['ms_per_step'], 'value': s['value'], 'nominal_value': s['nominal_value'],
                          'live_fraction': s['live_fraction'], 'replicates_alive_at_T': int(res['alive']),
                          'unit': 'filter-steps/s', 'kernel': w.kernel,
                          'hbm_gbs_algorithmic': s['hbm_gbs']}
            if label == 'config2_scaled':   # the scaled representation's worst replicates against exact arithmetic (device side)
                t = exact_tails_errors(w, 'scaled', w.d_mom.to_array((w.B, w.T, 2 * w.N)), w.d_means.to_array((w.B, w.T)),
                                       w.d_scales.to_array((w.B, w.T)), res['nell'])
                if t:
                    out[label]['max_rel_err_vs_exact_arithmetic_worst_replicates'] = t
            if label == 'config5_B512':     # the reference-shaped N-D entry point on the same data, host arrays in and out
                out[label]['end_to_end_ms'] = end_to_end(w)['end_to_end_ms']
            w.release()
        except Exception as e:   # noqa: BLE001
            out[label] = {'workload': name, 'error': repr(e)}
    out.update(extra_kernels(device))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the oracle on the host cores, as the checker and as a reported baseline
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline_nd(args, w, res):
    """The NumPy/LAPACK oracle (one core) on one replicate over a bounded number of steps; kind = "port"."""
    from oracle import multi_dims as omd, tme_sympy, parity
    mi, inds, T = w.mi, w.inds, w.T
    dt, _, ogs, odrift, odisp, _, opmf = omd.prey_predator(mi)
    kind, order = w.transition.rsplit('_', 1)
    if kind == 'tme':
        _, ocms, omean, _ = tme_sympy.sde_cond_moments_tme_nd(odrift, odisp, 2, dt, int(order), mi)
        sig = 'multi-index'
    else:
        _, ocms, omean = tme_sympy.sde_cond_moments_normal_nd(odrift, odisp, 2, dt, int(order), mi)
        sig = 'index'
    steps = 4
    t0 = time.perf_counter()
    omd.moment_filter_nd_cms((ocms, sig), omean, opmf, w.ys[0, :steps], (mi, inds), ogs.cms, ogs.mean)
    el = time.perf_counter() - t0
    steps2 = int(min(T, max(steps, steps / el * args.cpu_seconds)))
    t0 = time.perf_counter()
    ref = omd.moment_filter_nd_cms((ocms, sig), omean, opmf, w.ys[0, :steps2], (mi, inds), ogs.cms, ogs.mean)
    el = time.perf_counter() - t0
    out = {'value': steps2 / el, 'unit': 'filter-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'replicate 0, first {steps2} of T={T} steps of the same workload, oracle/multi_dims.py '
                     f'(NumPy + LAPACK, lambdified SymPy transition), {el:.1f} s'}
    if w.d_mom is not None and w.mode == 'central':
        dev_m = w.d_mom.to_array((w.B, T, w.z))[0, :steps2]
        dev_mean = w.d_means.to_array((w.B, T, 2))[0, :steps2]
        deg = mi.sum(axis=1)
        e = parity.rel_err(dev_m, ref[0], parity.natural_magnitude_nd(ref[0], mi))[:, deg >= 2]
        out['max_rel_err_vs_device'] = {
            'definition': 'relative error |device - cpu| / max(|cpu|, 1e-2 prod_k sd_k^n_k); moments of total degree >= 2',
            'steps_compared': steps2,
            'means': parity.quantity_errors(dev_mean, ref[1]),
            'variances': parity.quantity_errors(dev_m[:, [5, 3]], ref[0][:, [5, 3]]),
            'moments_all_degrees': {'max': float(e.max()), 'p99': float(np.quantile(e, .99)), 'p50': float(np.quantile(e, .5)),
                                    'n': int(e.size)},
            'moments_max_by_total_degree': {int(dg): float(e[:, deg[deg >= 2] == dg].max())
                                            for dg in range(2, int(deg.max()) + 1)}}
    if steps2 == T:
        out['nll_rel_diff_vs_device'] = float(abs(ref[2] - res['nell'][0]) / abs(ref[2]))
    return out


def cpu_baseline(args, w, res):
    """The oracle's C port (OpenMP over replicates) timed on this box's host cores, on a bounded sample of the same
    workload, WITH moments: BASELINE.json's metric is "steps/s + max |moment err|", so the same sample yields the
    per-quantity maximum relative error and the first-NaN agreement between the device and the CPU implementation.
    The reference's own JAX-CPU path cannot run here (no JAX in the image): kind = "port"."""
    from oracle import c_oracle, tme_sympy, models as om, parity
    model, N, T, mode, tables, lik, ic, ys = w.model, w.N, w.T, w.mode, w.tables, w.lik, w.ic, w.ys
    dev_nell, dev_first = res['nell'], np.where(res['first_nan'] >= 0, res['first_nan'], T)
    # threads: the cores this process can really use (cgroup quota, not the affinity mask: 256 OpenMP threads on a 16-CPU
    # share time-slice each other), pinned one per core; the source is rebuilt with -march=native for this host
    os.environ.setdefault('OMP_PROC_BIND', 'close')
    os.environ.setdefault('OMP_PLACES', 'cores')
    flags = c_oracle.use_native_build(os.path.join(tempfile.gettempdir(), f'mfs_oracle_native_{os.getuid()}'))
    quota_cores, affinity = c_oracle.host_cores()
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

    def run(nb, want_moments):
        cf = coef[:nb] if coef.ndim == 3 else coef
        lp = lik.params[:nb] if lik.params.ndim == 2 else lik.params
        t0 = time.perf_counter()
        r = c_oracle.filter_1d(modei, N, ys[:nb], m0, ic.mean, np.sqrt(ic.variance), kind, umap, tables.n_terms,
                               cf, tables.mean_x_coef, lik_kind, lp, want_moments=want_moments, nthreads=threads)
        return time.perf_counter() - t0, r

    # the thread count that is fastest on a short probe (the quota may be invisible from inside the container)
    probe = {}
    for threads in sorted({c for c in (quota_cores, 8, 16, 32, 64, 128, affinity) if 1 <= c <= affinity}):
        nbp = min(ys.shape[0], 4 * threads)
        el, _ = run(nbp, False)
        probe[threads] = nbp * T / el
    threads = max(probe, key=probe.get)
    rate = probe[threads]
    nb = int(min(ys.shape[0], max(4 * threads, rate * args.cpu_seconds / T)))
    nb = min(nb, max(1, int(3e9 // (T * 2 * N * 8))))     # host-memory bound of the moment sample
    el, r = run(nb, True)
    cm, cmeans, cscales, cnell = r
    c_first = parity.first_nan_steps(np.concatenate([cmeans[..., None], cm], axis=-1), T) if mode != 'raw' \
        else parity.first_nan_steps(cm, T)
    live = int(np.minimum(c_first + 1, T).sum())
    both = np.isfinite(cnell) & np.isfinite(dev_nell[:nb])
    out = {'value': live / el, 'unit': 'filter-steps/s', 'cores': threads, 'kind': 'port',
           'sample': f'first {nb} replicates x T={T} of the same workload incl. all 2N moments, oracle/c/mfs_oracle.c, '
                     f'OpenMP x{threads}, {el:.1f} s; live steps only ({live / (nb * T):.2f} of nominal)',
           'nominal_value': nb * T / el, 'per_thread_value': live / el / threads,
           'build': flags, 'threads': threads, 'affinity_mask_cpus': affinity, 'cgroup_quota_cpus': quota_cores,
           'pinning': f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}",
           'thread_probe_nominal_steps_per_s': {str(k): v for k, v in probe.items()},
           'replicates_finite_in_both': int(both.sum()),
           'replicates_finite_device_only': int((np.isfinite(dev_nell[:nb]) & ~np.isfinite(cnell)).sum()),
           'replicates_finite_cpu_only': int((~np.isfinite(dev_nell[:nb]) & np.isfinite(cnell)).sum()),
           'first_nan_agreement_device_vs_cpu': parity.first_nan_agreement(dev_first[:nb], c_first, T)}
    # ---- BASELINE.json: "max |moment err|" -- per quantity, over every filter-step finite on both sides
    err = {'definition': 'relative error |device - cpu| / max(|cpu|, floor) over the filter-steps finite in both; moments '
                         'scaled per order as in oracle/parity.py; the maxima are set by the replicates closest to losing '
                         'positive definiteness (cond(Hankel) up to 1e16 at N = 15), the quantiles show the bulk',
           'nll': parity.quantity_errors(dev_nell[:nb], cnell)}
    if w.d_mom is not None:
        dm = w.d_mom.to_array((w.B, T, 2 * N))[:nb]
        floor = parity.moment_floor(cm)
        err['moments_all_orders'] = parity.quantity_errors(dm, cm, floor)
        err['moments_by_order_max'] = parity.moment_errors_by_order(dm, cm)
        if mode == 'central':
            err['variance'] = parity.quantity_errors(dm[..., 2], cm[..., 2])
        del dm
    if mode != 'raw':
        err['mean'] = parity.quantity_errors(w.d_means.to_array((w.B, T))[:nb], cmeans, 1e-12)
    if mode == 'scaled' and w.d_scales is not None:
        err['scale'] = parity.quantity_errors(w.d_scales.to_array((w.B, T))[:nb], cscales)
    out['max_rel_err_vs_device'] = err
    exact = exact_arithmetic_errors(w, cm, cmeans, cnell, nb, dev_nell)
    if exact:
        out['max_rel_err_vs_exact_arithmetic'] = exact
    return out


def exact_arithmetic_errors(w, cm, cmeans, cnell, nb, dev_nell=None):
    """Device and C port against the 80-digit trajectories of tests/golden/filter_cfg2_exact_T1000.npz (the reference's
    algorithm without rounding, oracle/exact_mp.py) on the replicates the fixture holds -- the benchmark batch's first 8.
    Two fp64 implementations can only be compared with each other up to their own errors; this is each one's distance from
    the truth, over the filter-steps where it is finite."""
    from oracle import parity
    path = os.path.join(ROOT, 'tests', 'golden', 'filter_cfg2_exact_T1000.npz')
    if w.name != DEFAULT_WORKLOAD or not w.full_size or w.d_mom is None or not os.path.exists(path):
        return None
    e = np.load(path)
    B, T = int(e['B']), int(e['T'])
    if nb < B or T != w.T or not np.array_equal(np.packbits(w.ys[:B].astype(np.uint8), axis=1), e['ys_bits']):
        return None
    steps = e['moment_steps']
    dm = w.d_mom.to_array((w.B, T, 2 * w.N))[:B]
    dmeans = w.d_means.to_array((w.B, T))[:B]
    sd = np.sqrt(e['central_variances'])
    floor = parity.moment_floor(e['central_moments'])

    def score(mom, means):
        with np.errstate(all='ignore'):
            return {'mean': parity.quantity_errors(means, e['central_means'], np.maximum(sd, 1e-300)),
                    'variance': parity.quantity_errors(mom[..., 2], e['central_variances']),
                    'moments_all_orders': parity.quantity_errors(mom[:, steps], e['central_moments'], floor),
                    'moments_by_order_max': parity.moment_errors_by_order(mom[:, steps], e['central_moments']),
                    'first_non_finite_step': [int(v) for v in parity.first_nan_steps(means[..., None], T)]}
    out = {'fixture': 'tests/golden/filter_cfg2_exact_T1000.npz (80-digit mpmath, 8 replicates x 1000 steps)',
           'device': score(dm, dmeans), 'c_port': score(cm[:B], cmeans[:B])}
    tails = exact_tails_errors(w, 'central', w.d_mom.to_array((w.B, T, 2 * w.N)), w.d_means.to_array((w.B, T)), None, dev_nell,
                               (cm, cmeans, None, cnell) if nb >= w.B else None)
    if tails:
        out['worst_replicates'] = tails
    return out


def exact_tails_errors(w, mode, dm, dmeans, dscales, dnell, cpu=None):
    """The replicates that set `max_rel_err_vs_device` (largest variance / scale, NLL, mean deviation and first-NaN gap
    between the device and the C port over the whole 4096 x 1000 batch: tools/select_tails.py) against their exact-arithmetic
    trajectories, tests/golden/filter_cfg2_exact_tails.npz (oracle/exact_mp.py at 200 / 500 digits): each side's own distance
    from the truth on exactly the replicates where the two disagree most -- which attributes the maxima to a side."""
    from oracle import parity
    path = os.path.join(ROOT, 'tests', 'golden', 'filter_cfg2_exact_tails.npz')
    if not os.path.exists(path) or not w.full_size or w.model != 'benes' or w.N != 15 or w.transition != 'tme_3':
        return None
    e = np.load(path)
    idx, T = e[f'{mode}_idx'], int(e['T'])
    if T != w.T or idx.max() >= w.B or not np.array_equal(np.packbits(w.ys[idx].astype(np.uint8), axis=1), e[f'{mode}_ys_bits']):
        return None

    def one(m, means, second, nell):
        sc = parity.score_against_exact_tails(e, mode, m, means, second, nell)
        first = parity.first_nan_steps(means[..., None], T)
        hor = np.where(e[f'{mode}_exact_first_nan'] >= 0, e[f'{mode}_exact_first_nan'], T)
        return {'max': sc['max'], 'replicates_over_1e-6': sc['replicates_over_1e-6'],
                'finite_up_to_the_exact_horizon': int(np.sum(first >= hor)), 'filter_steps_compared': int(sc['finite_steps'].sum())}
    second = dm[idx][..., 2] if mode == 'central' else dscales[idx]
    nd = dnell[idx] if dnell is not None else np.full(len(idx), np.nan)
    out = {'fixture': 'tests/golden/filter_cfg2_exact_tails.npz: the %d replicates of this batch with the largest device-vs-C-port '
                      'deviations (8 by variance / scale, 8 by NLL, 4 by mean, the rest by first-NaN gap), exact arithmetic at '
                      '200 / 500 digits over T = 1000 or up to the algorithm\'s own loss of positive definiteness' % len(idx),
           'mode': mode, 'replicates': int(len(idx)), 'device': one(dm[idx], dmeans[idx], second, nd)}
    if cpu is not None:
        cm, cmeans, cscales, cnell = cpu
        out['c_port'] = one(cm[idx], cmeans[idx], cm[idx][..., 2] if mode == 'central' else cscales[idx], cnell[idx])
        out['attribution'] = ('maxima of max_rel_err_vs_device = the C port\'s distance from exact arithmetic'
                              if out['c_port']['max']['variance' if mode == 'central' else 'scale'] >
                              10 * out['device']['max']['variance' if mode == 'central' else 'scale'] else 'see both sides')
    return out


if __name__ == '__main__':
    main()
