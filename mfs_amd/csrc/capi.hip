// capi.hip -- the C ABI of libmfs_hip.so (include/mfs_hip.h): argument checking, device staging, chunked launches,
// hipGraph capture.  No compute lives here; the kernels are in filter1d_kernel.hpp.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

#include "filternd_kernel.hpp"
#include "filter1d_grad.hpp"
#include "pool.hpp"

namespace mfs {
KernelEntry g_table[MFS_MAX_N + 1][kSlots];  // filled by the static registrars in filter1d_inst.hip
Filter1dFastLaunch g_fast_filter[MFS_MAX_N + 1][4];
Filter1dFastLaunch g_fast_filter_wide[MFS_MAX_N + 1][4];
Filter1dFastLaunch g_fast_filter_ext[MFS_MAX_N + 1][4];
Filter1dFastLaunch g_fast_filter_ext_wide[MFS_MAX_N + 1][4];
int g_fast_ext_shift[MFS_MAX_N + 1][4];
Quad1dLaunch g_quad_ext[MFS_MAX_N + 1][4];
int g_quad_ext_lds[MFS_MAX_N + 1][4];
using Cf1dLaunch = hipError_t (*)(const Cf1dArgs&, int grid, int lds, hipStream_t);
Cf1dLaunch g_cf[MFS_MAX_N + 1][4];
using FilterNdLaunch = hipError_t (*)(const FilterNdArgs&, int grid, hipStream_t);
struct NdEntry { FilterNdLaunch launch, launch_gauss, launch_hi, launch_joint; int S, Z, lds_bytes, carry_doubles; };
extern NdEntry g_nd_table[8];  // filternd_inst.hip
hipError_t launch_elementary(int which, int n, const double* d_x, double* d_out, hipStream_t s);
extern Filter1dGradLaunch g_grad_table[17][5];  // filter1d_grad_inst.hip: [N <= 16][P <= 4]
}

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(MFS_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int check_model(const mfs_model_1d* m, int mode) {
    if (!m) return fail(MFS_EINVAL, "model is NULL");
    if (m->trans_kind != MFS_TRANS_OPERATOR && m->trans_kind != MFS_TRANS_GAUSSIAN)
        return fail(MFS_EINVAL, "unknown trans_kind %d", m->trans_kind);
    if (m->umap != MFS_U_IDENTITY && m->umap != MFS_U_TANH) return fail(MFS_EINVAL, "unknown umap %d", m->umap);
    if (m->degree < 0 || m->degree > MFS_MAX_DEGREE)
        return fail(MFS_EUNSUPPORTED, "polynomial degree %d outside [0, %d]", m->degree, MFS_MAX_DEGREE);
    if (m->trans_kind == MFS_TRANS_OPERATOR) {
        if (m->n_terms < 1 || m->n_terms > MFS_MAX_TERMS)
            return fail(MFS_EUNSUPPORTED, "n_terms %d outside [1, %d]", m->n_terms, MFS_MAX_TERMS);
        if (m->n_rows != m->n_terms + 1) return fail(MFS_EINVAL, "operator table needs n_rows = n_terms + 1");
    } else if (m->n_rows != 2) {
        return fail(MFS_EINVAL, "gaussian table needs n_rows = 2");
    }
    if (m->lik_kind < 0 || m->lik_kind > MFS_LIK_GAUSSIAN) return fail(MFS_EINVAL, "unknown lik_kind %d", m->lik_kind);
    if (m->n_lik < 1 || m->n_lik > MFS_MAX_LIK) return fail(MFS_EINVAL, "n_lik %d outside [1, %d]", m->n_lik, MFS_MAX_LIK);
    if (!m->coef || !m->lik) return fail(MFS_EINVAL, "model tables are NULL");
    if ((mode & 0xff) < MFS_MODE_RAW || (mode & 0xff) > MFS_MODE_SCALED || (mode & ~(0xff | MFS_MODE_ODD_TAIL)))
        return fail(MFS_EINVAL, "unknown mode %d", mode);
    return MFS_OK;
}

// Kernel choice.  Default: the register-resident fast path with the smallest lane group that holds the N + 1 rows of
// the extended Hankel matrix; stable = 1 and odd moment counts run its extended variant (the LDL^T completion only changes
// a rule when a pivot is not > 0: that rule alone takes the dense route, inside the same kernel).  MFS_SOLVER=dense uses
// the LDS-tile dense path throughout.  MFS_LANES_PER_FILTER=16|32|64 overrides the group width (experiments; with
// stable = 1 or an odd count a non-default width runs the dense path -- the extended variant exists for the default only).
int pick_slot(int N, int stable, int odd_tail = 0) {
    bool dense = false;
    if (const char* e = getenv("MFS_SOLVER")) dense = dense || (strcmp(e, "dense") == 0);
    int want = 0;
    if (const char* e = getenv("MFS_LANES_PER_FILTER")) want = atoi(e);
    if ((stable != 0 || odd_tail != 0) && want != 0) dense = true;
    if (dense) {
        int gi = (N <= 16) ? 0 : (N <= 32) ? 1 : 2;
        if (want == 64) gi = 2;
        else if (want == 32 && N <= 32) gi = 1;
        else if (want == 16 && N <= 16) gi = 0;
        return gi;
    }
    int gi = (N + 1 <= 8) ? 3 : (N + 1 <= 16) ? 0 : (N + 1 <= 32) ? 1 : 2;
    if (want == 64) gi = 2;
    else if (want == 32 && N + 1 <= 32) gi = 1;
    else if (want == 16 && N + 1 <= 16) gi = 0;
    else if (want == 8 && N + 1 <= 8) gi = 3;
    return 3 + gi;
}

}  // namespace

struct mfs_plan_1d {
    mfs_model_1d model;  // device pointers inside
    int mode, N, T, B, stable, chunk, device, extra;
    int slot, G, fpb, grid, lds_bytes, lds_doubles;
    bool single_wave_per_simd = false;
    bool ext = false;               // fast path, extended variant (stable = 1 or an odd moment count)
    double* d_coef = nullptr;
    double* d_lik = nullptr;
    double* c_mom = nullptr;
    double* c_mean = nullptr;
    double* c_scale = nullptr;
    double* c_nell = nullptr;
    int32_t* c_first_nan = nullptr;
    double* c_lam = nullptr;
    hipStream_t own_stream = nullptr;   // created on first use (run() without a caller stream, graph capture)
    // cached graph for the last set of run() pointers
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const void* key[10] = {nullptr};
    int key_m0_batched = -1;
};

extern "C" {

int mfs_version(void) { return MFS_ABI_VERSION; }
const char* mfs_last_error(void) { return g_err.c_str(); }

int mfs_device_count(int* count) {
    if (!count) return fail(MFS_EINVAL, "count is NULL");
    HIP_TRY(hipGetDeviceCount(count));
    return MFS_OK;
}
int mfs_set_device(int device) { HIP_TRY(hipSetDevice(device)); return MFS_OK; }
int mfs_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return MFS_OK; }
int mfs_device_name(int device, char* buf, int buflen) {
    if (!buf || buflen <= 0) return fail(MFS_EINVAL, "bad buffer");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return MFS_OK;
}
int mfs_malloc(void** dptr, uint64_t bytes) {
    if (!dptr) return fail(MFS_EINVAL, "dptr is NULL");
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(MFS_ENOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return MFS_OK;
}
int mfs_free(void* dptr) { if (dptr) HIP_TRY(hipFree(dptr)); return MFS_OK; }
int mfs_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MFS_OK;
}
int mfs_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MFS_OK;
}
int mfs_memset(void* dst, int value, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return MFS_OK;
}
int mfs_stream_create(void** stream) {
    if (!stream) return fail(MFS_EINVAL, "stream is NULL");
    HIP_TRY(hipStreamCreateWithFlags((hipStream_t*)stream, hipStreamNonBlocking));
    return MFS_OK;
}
int mfs_stream_destroy(void* stream) { if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream)); return MFS_OK; }
int mfs_stream_synchronize(void* stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return MFS_OK; }
int mfs_event_create(void** event) {
    if (!event) return fail(MFS_EINVAL, "event is NULL");
    HIP_TRY(hipEventCreate((hipEvent_t*)event));
    return MFS_OK;
}
int mfs_event_destroy(void* event) { if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event)); return MFS_OK; }
int mfs_event_record(void* event, void* stream) { HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); return MFS_OK; }
int mfs_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!ms) return fail(MFS_EINVAL, "ms is NULL");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------------------------
// `quiesced`: the caller has already synchronised every stream the plan's buffers were used on
static void destroy_plan_1d(mfs_plan_1d* p, bool quiesced) {
    hipSetDevice(p->device);
    if (!quiesced) hipDeviceSynchronize();   // what hipFree did implicitly: pool blocks must be idle when they go back
    if (p->exec) hipGraphExecDestroy(p->exec);
    if (p->graph) hipGraphDestroy(p->graph);
    // model tables and carry state come from the device pool (pool.hpp): a plan per call costs no hipMalloc / hipFree
    mfs::BlockPool<false>& pool = mfs::device_state(p->device).device;
    for (void* b : {(void*)p->d_coef, (void*)p->d_lik, (void*)p->c_mom, (void*)p->c_mean, (void*)p->c_scale,
                    (void*)p->c_nell, (void*)p->c_first_nan, (void*)p->c_lam})
        pool.release(b);
    if (p->own_stream) hipStreamDestroy(p->own_stream);
    delete p;
}

int mfs_plan_1d_destroy(mfs_plan_1d* p) {
    if (p) destroy_plan_1d(p, false);
    return MFS_OK;
}

int mfs_plan_1d_create(mfs_plan_1d** plan, const mfs_model_1d* model, int mode, int N, int T, int B, int stable,
                       int chunk, int device) {
    if (!plan) return fail(MFS_EINVAL, "plan is NULL");
    *plan = nullptr;
    if (int rc = check_model(model, mode)) return rc;
    if (N < 2 || N > MFS_MAX_N) return fail(MFS_EUNSUPPORTED, "N = %d outside [2, %d]", N, MFS_MAX_N);
    if (T < 0 || B < 0) return fail(MFS_EINVAL, "negative T or B");
    if (chunk < 0) return fail(MFS_EINVAL, "negative chunk");
    const int extra = (mode & MFS_MODE_ODD_TAIL) ? 1 : 0;
    mode &= 0xff;
    if (extra && chunk != 0 && chunk < T) return fail(MFS_EUNSUPPORTED, "an odd moment count runs in one launch (chunk = 0)");
    const int slot = pick_slot(N, stable, extra);
    const mfs::KernelEntry& ke = mfs::g_table[N][slot];
    if (!ke.quad) return fail(MFS_EUNSUPPORTED, "no kernel compiled for N = %d", N);
    HIP_TRY(hipSetDevice(device));

    mfs_plan_1d* p = new mfs_plan_1d();
    p->model = *model;
    p->mode = mode; p->N = N; p->T = T; p->B = B; p->stable = stable; p->device = device; p->extra = extra;
    p->chunk = (chunk == 0 || chunk > T) ? T : chunk;
    p->slot = slot;
    p->G = ke.lanes_per_filter;
    p->fpb = ke.waves_per_block * (64 / p->G);
    p->grid = (B + p->fpb - 1) / p->fpb;
    p->lds_doubles = ke.lds_doubles_per_filter;
    if (slot >= 3) p->lds_doubles += ((model->degree + 4) & ~3) * 10;  // fast path: + model table [ceil4(degree + 1)][kCoefRows]
    p->ext = slot >= 3 && (stable != 0 || extra != 0);
    if (p->ext) {
        if (!mfs::g_fast_filter_ext[N][slot - 3]) { delete p; return fail(MFS_EUNSUPPORTED, "no extended kernel for N = %d", N); }
        p->lds_doubles += mfs::g_fast_ext_shift[N][slot - 3];
    }
    p->lds_bytes = p->fpb * p->lds_doubles * 8;
    {   // blocks of the fast path are single waves: with at most one per SIMD the wide-register variant costs nothing
        int cus = 0;   // (hipDeviceGetAttribute: one integer, not the whole property structure, per plan)
        p->single_wave_per_simd = (slot >= 3 &&
                                   hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
                                   p->grid <= 4 * cus);
    }

    const size_t ncoef = (size_t)(model->coef_batched ? B : 1) * model->n_rows * (model->degree + 1);
    const size_t nlik = (size_t)(model->lik_batched ? B : 1) * model->n_lik;
    hipError_t e = hipSuccess;
    mfs::BlockPool<false>& pool = mfs::device_state(device).device;
    auto alloc = [&](void** d, size_t bytes) { if (e == hipSuccess) e = pool.acquire(d, bytes); };
    alloc((void**)&p->d_coef, ncoef * 8);
    alloc((void**)&p->d_lik, nlik * 8);
    alloc((void**)&p->c_mom, (size_t)B * 2 * N * 8);
    alloc((void**)&p->c_mean, (size_t)B * 8);
    alloc((void**)&p->c_scale, (size_t)B * 8);
    alloc((void**)&p->c_nell, (size_t)B * 8);
    alloc((void**)&p->c_first_nan, (size_t)B * 4);
    if (p->chunk < T && slot >= 3) alloc((void**)&p->c_lam, (size_t)B * 2 * p->G * 8);
    if (e == hipSuccess) e = hipMemcpy(p->d_coef, model->coef, ncoef * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_lik, model->lik, nlik * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        destroy_plan_1d(p, true);
        return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "plan setup failed: %s", hipGetErrorString(e));
    }
    p->model.coef = p->d_coef;
    p->model.lik = p->d_lik;
    *plan = p;
    return MFS_OK;
}

int mfs_plan_1d_geometry(const mfs_plan_1d* p, int* lanes_per_filter, int* filters_per_block, int* grid,
                         int* lds_bytes_per_block) {
    if (!p) return fail(MFS_EINVAL, "plan is NULL");
    if (lanes_per_filter) *lanes_per_filter = p->G;
    if (filters_per_block) *filters_per_block = p->fpb;
    if (grid) *grid = p->grid;
    if (lds_bytes_per_block) *lds_bytes_per_block = p->lds_bytes;
    return MFS_OK;
}

static hipError_t launch_filter(mfs_plan_1d* p, const mfs::Filter1dArgs& a, hipStream_t s) {
    if (p->slot >= 3) {
        const mfs::Filter1dFastLaunch wide = (p->ext ? mfs::g_fast_filter_ext_wide : mfs::g_fast_filter_wide)[p->N][p->slot - 3];
        const mfs::Filter1dFastLaunch narrow = (p->ext ? mfs::g_fast_filter_ext : mfs::g_fast_filter)[p->N][p->slot - 3];
        return (wide && p->single_wave_per_simd ? wide : narrow)(a, p->grid, p->lds_doubles, s);
    }
    return mfs::g_table[p->N][p->slot].filter(a, p->grid, p->lds_bytes, s);
}

static int enqueue_chunks(mfs_plan_1d* p, const mfs::Filter1dArgs& base, hipStream_t s) {
    mfs::Filter1dArgs a = base;
    if (p->T == 0) {  // empty measurement sequence: nell = 0, nothing else to write
        a.t_begin = 0; a.t_end = 0;
        hipError_t e = launch_filter(p, a, s);
        if (e != hipSuccess) return fail(MFS_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
        return MFS_OK;
    }
    for (int t0 = 0; t0 < p->T; t0 += p->chunk) {
        a.t_begin = t0;
        a.t_end = (t0 + p->chunk < p->T) ? t0 + p->chunk : p->T;
        hipError_t e = launch_filter(p, a, s);
        if (e != hipSuccess) return fail(MFS_EHIP, "kernel launch failed: %s", hipGetErrorString(e));
    }
    return MFS_OK;
}

static mfs::Filter1dArgs plan_args(const mfs_plan_1d* p, const double* d_m0, int m0_batched, const double* d_mean0,
                                    const double* d_scale0, const double* d_ys, double* d_out_moments,
                                    double* d_out_means, double* d_out_scales, double* d_out_nell,
                                    int32_t* d_out_first_nan) {
    mfs::Filter1dArgs a;
    memset(&a, 0, sizeof(a));
    a.mode = p->mode; a.T = p->T; a.B = p->B; a.stable = p->stable; a.extra = p->extra;
    a.trans_kind = p->model.trans_kind; a.umap = p->model.umap; a.n_terms = p->model.n_terms;
    a.degree = p->model.degree; a.n_rows = p->model.n_rows; a.coef_batched = p->model.coef_batched;
    a.lik_kind = p->model.lik_kind; a.n_lik = p->model.n_lik; a.lik_batched = p->model.lik_batched;
    a.mean_x_coef = p->model.mean_x_coef; a.coef = p->model.coef; a.lik = p->model.lik;
    a.m0 = d_m0; a.m0_batched = m0_batched; a.mean0 = d_mean0; a.scale0 = d_scale0; a.ys = d_ys;
    a.c_mom = p->c_mom; a.c_mean = p->c_mean; a.c_scale = p->c_scale; a.c_nell = p->c_nell;
    a.c_first_nan = p->c_first_nan;
    a.c_lam = p->c_lam;
    if (const char* e = getenv("MFS_PREDICT_RULE")) a.recompute_rule = (strcmp(e, "recompute") == 0);   // A/B switch
    a.out_mom = d_out_moments; a.out_mean = (p->mode != MFS_MODE_RAW) ? d_out_means : nullptr;
    a.out_scale = (p->mode == MFS_MODE_SCALED) ? d_out_scales : nullptr;
    a.out_nell = d_out_nell; a.out_first_nan = d_out_first_nan;
    return a;
}

int mfs_plan_1d_run(mfs_plan_1d* p, const double* d_m0, int m0_batched, const double* d_mean0,
                    const double* d_scale0, const double* d_ys, double* d_out_moments, double* d_out_means,
                    double* d_out_scales, double* d_out_nell, int32_t* d_out_first_nan, void* stream) {
    if (!p) return fail(MFS_EINVAL, "plan is NULL");
    if (!d_m0 || !d_out_nell || (p->T > 0 && !d_ys)) return fail(MFS_EINVAL, "m0 / ys / out_nell must not be NULL");
    if (p->mode != MFS_MODE_RAW && !d_mean0) return fail(MFS_EINVAL, "mean0 is required in central / scaled mode");
    if (p->mode == MFS_MODE_SCALED && !d_scale0) return fail(MFS_EINVAL, "scale0 is required in scaled mode");
    if (p->B == 0) return MFS_OK;
    HIP_TRY(hipSetDevice(p->device));
    const int nchunks = (p->T + p->chunk - 1) / (p->chunk > 0 ? p->chunk : 1);
    if (!p->own_stream && (!stream || nchunks > 1)) HIP_TRY(hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking));
    hipStream_t s = stream ? (hipStream_t)stream : p->own_stream;

    const mfs::Filter1dArgs a = plan_args(p, d_m0, m0_batched, d_mean0, d_scale0, d_ys, d_out_moments, d_out_means,
                                          d_out_scales, d_out_nell, d_out_first_nan);
    if (nchunks <= 1) return enqueue_chunks(p, a, s);  // one launch: a graph adds only replay overhead

    // several chunk launches: capture them once into a hipGraph keyed on the buffer set, then replay
    const void* key[10] = {d_m0, d_mean0, d_scale0, d_ys, d_out_moments, d_out_means, d_out_scales, d_out_nell,
                           d_out_first_nan, nullptr};
    const bool hit = p->exec && memcmp(key, p->key, sizeof(key)) == 0 && p->key_m0_batched == m0_batched;
    if (!hit) {
        if (p->exec) { hipGraphExecDestroy(p->exec); p->exec = nullptr; }
        if (p->graph) { hipGraphDestroy(p->graph); p->graph = nullptr; }
        HIP_TRY(hipStreamBeginCapture(p->own_stream, hipStreamCaptureModeThreadLocal));
        int rc = enqueue_chunks(p, a, p->own_stream);
        hipError_t e = hipStreamEndCapture(p->own_stream, &p->graph);
        if (rc != MFS_OK) return rc;
        if (e != hipSuccess) return fail(MFS_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
        HIP_TRY(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
        memcpy(p->key, key, sizeof(key));
        p->key_m0_batched = m0_batched;
    }
    HIP_TRY(hipGraphLaunch(p->exec, s));
    return MFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// host-pointer convenience path
// ---------------------------------------------------------------------------------------------------------------
// How many T-chunks the host entry splits a run into when the moments are streamed out: the kernel of chunk k + 1 runs
// while chunk k's slice of out_moments travels to the host (2-D copy: B rows of chunk x 2N doubles).  MFS_HOST_CHUNKS
// overrides (1 = one launch, copies afterwards).
// hipMemcpy2DAsync takes a pitch: a row of T x (moments per step) x 8 bytes beyond the runtime's limit (runs of ~1e6 steps and
// more) would fail the whole call, so such runs go out in one linear copy instead (no copy / compute overlap)
static bool pitch_ok(size_t pitch_bytes, int device) {
    int maxp = 0;
    if (hipDeviceGetAttribute(&maxp, hipDeviceAttributeMaxPitch, device) != hipSuccess || maxp <= 0) return pitch_bytes < ((size_t)1 << 31);
    return pitch_bytes <= (size_t)maxp;
}

static int host_chunks(size_t moment_bytes, int T) {
    int n = (int)(moment_bytes / ((size_t)32 << 20));
    if (const char* e = getenv("MFS_HOST_CHUNKS")) n = atoi(e);
    if (n > 16) n = 16;
    if (n > T) n = T;
    return n < 1 ? 1 : n;
}

int mfs_filter_1d(const mfs_model_1d* model, int mode, int N, int T, int B, const double* m0, int m0_batched,
                  const double* mean0, const double* scale0, const double* ys, int stable, double* out_moments,
                  double* out_means, double* out_scales, double* out_nell, int32_t* out_first_nan, int device,
                  void* stream) {
    if (!m0 || !out_nell || (T > 0 && B > 0 && !ys)) return fail(MFS_EINVAL, "m0 / ys / out_nell must not be NULL");
    const int extra = (mode & MFS_MODE_ODD_TAIL) ? 1 : 0, full_mode = mode;
    mode &= 0xff;
    if (mode != MFS_MODE_RAW && !mean0) return fail(MFS_EINVAL, "mean0 is required in central / scaled mode");
    if (mode == MFS_MODE_SCALED && !scale0) return fail(MFS_EINVAL, "scale0 is required in scaled mode");
    const size_t M2 = 2 * (size_t)N + extra, nb = m0_batched ? B : 1;     // doubles per moment row
    const size_t mom_bytes = out_moments ? (size_t)B * T * M2 * 8 : 0;
    const int nchunks = (B > 0 && T > 0 && !extra && pitch_ok((size_t)T * M2 * 8, device)) ? host_chunks(mom_bytes, T) : 1;
    const int chunk = (nchunks > 1) ? (T + nchunks - 1) / nchunks : 0;
    mfs_plan_1d* p = nullptr;
    if (int rc = mfs_plan_1d_create(&p, model, full_mode, N, T, B, stable, chunk, device)) return rc;
    struct PlanGuard { mfs_plan_1d* p; ~PlanGuard() { destroy_plan_1d(p, true); } } guard{p};  // (every exit below is quiesced)
    if (B == 0) return MFS_OK;

    // device staging from the library's pool, streams / events from its context cache: no hipMalloc, hipFree,
    // hipStreamCreate or hipEventCreate on a steady-state call (SURVEY.md section 8b "Ownership")
    mfs::Lease lease(device);
    mfs::CallContext* cx = nullptr;
    double *d_m0 = nullptr, *d_mean0 = nullptr, *d_scale0 = nullptr, *d_ys = nullptr, *d_mom = nullptr,
           *d_means = nullptr, *d_scales = nullptr, *d_nell = nullptr;
    int32_t* d_fn = nullptr;
    hipError_t e = lease.context(&cx);
    auto alloc = [&](auto** d, size_t bytes) { if (e == hipSuccess) e = lease.device_block(d, bytes); };
    alloc(&d_m0, nb * M2 * 8);
    alloc(&d_mean0, nb * 8);
    alloc(&d_scale0, nb * 8);
    alloc(&d_ys, (size_t)B * T * 8);
    if (out_moments) alloc(&d_mom, mom_bytes);
    if (out_means && mode != MFS_MODE_RAW) alloc(&d_means, (size_t)B * T * 8);
    if (out_scales && mode == MFS_MODE_SCALED) alloc(&d_scales, (size_t)B * T * 8);
    alloc(&d_nell, (size_t)B * 8);
    alloc(&d_fn, (size_t)B * 4);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_filter_1d staging: %s", hipGetErrorString(e));
    hipStream_t s = stream ? (hipStream_t)stream : cx->compute;
    auto h2d = [&](void* d, const void* h, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
    };
    h2d(d_m0, m0, nb * M2 * 8);
    if (mean0) h2d(d_mean0, mean0, nb * 8);
    if (scale0) h2d(d_scale0, scale0, nb * 8);
    h2d(d_ys, ys, (size_t)B * T * 8);

    int rc = MFS_OK;
    if (e == hipSuccess) {
        mfs::Filter1dArgs a = plan_args(p, d_m0, m0_batched, d_mean0, d_scale0, d_ys, d_mom, d_means, d_scales, d_nell, d_fn);
        if (nchunks <= 1) {
            rc = enqueue_chunks(p, a, s);
            if (rc == MFS_OK && out_moments && T > 0) e = hipMemcpyAsync(out_moments, d_mom, mom_bytes, hipMemcpyDeviceToHost, s);
        } else {
            // chunk launches on `s`; after each one an event releases that chunk's slice of the moments to the copy
            // stream -- same chunking, same carry and therefore the same bits as the plan's graph of chunk launches
            const size_t pitch = (size_t)T * M2 * 8;
            for (int k = 0, t0 = 0; t0 < T && rc == MFS_OK && e == hipSuccess; ++k, t0 += chunk) {
                a.t_begin = t0;
                a.t_end = (t0 + chunk < T) ? t0 + chunk : T;
                e = launch_filter(p, a, s);
                if (e != hipSuccess) { rc = fail(MFS_EHIP, "kernel launch failed: %s", hipGetErrorString(e)); break; }
                e = hipEventRecord(cx->ev[k], s);
                if (e == hipSuccess) e = hipStreamWaitEvent(cx->copy, cx->ev[k], 0);
                if (e == hipSuccess && out_moments)
                    e = hipMemcpy2DAsync(out_moments + (size_t)t0 * M2, pitch, d_mom + (size_t)t0 * M2, pitch,
                                         (size_t)(a.t_end - t0) * M2 * 8, (size_t)B, hipMemcpyDeviceToHost, cx->copy);
            }
        }
    }
    auto d2h = [&](void* h, const void* d, size_t bytes) {
        if (rc == MFS_OK && e == hipSuccess && h && d && bytes) e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    };
    d2h(out_means, d_means, (size_t)B * T * 8);
    d2h(out_scales, d_scales, (size_t)B * T * 8);
    d2h(out_nell, d_nell, (size_t)B * 8);
    d2h(out_first_nan, d_fn, (size_t)B * 4);
    // quiesce both streams whatever happened above: the pool blocks go back when `lease` and `guard` unwind
    const hipError_t e1 = hipStreamSynchronize(s), e2 = hipStreamSynchronize(cx->copy);
    if (rc != MFS_OK) return rc;
    if (e == hipSuccess) e = (e1 != hipSuccess) ? e1 : e2;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_filter_1d: %s", hipGetErrorString(e));
    return MFS_OK;
}

int mfs_filter_1d_grad(const mfs_model_1d* model, const double* dcoef, const double* dlik, int n_par, int mode, int N,
                       int T, int B, const double* m0, int m0_batched, const double* mean0, const double* scale0,
                       const double* ys, double* out_nell, double* out_grad, int32_t* out_first_nan, int device,
                       void* stream) {
    if (int rc = check_model(model, mode)) return rc;
    if (mode & MFS_MODE_ODD_TAIL) return fail(MFS_EUNSUPPORTED, "the gradient entry point takes 2N moments");
    if (n_par < 1 || n_par > 4) return fail(MFS_EUNSUPPORTED, "n_par = %d outside [1, 4]", n_par);
    if (N < 2 || N > 16) return fail(MFS_EUNSUPPORTED, "N = %d outside [2, 16] for the gradient kernel", N);
    if (T < 0 || B < 0) return fail(MFS_EINVAL, "negative T or B");
    if (!dcoef || !dlik || !m0 || !out_nell || !out_grad || (T > 0 && B > 0 && !ys)) return fail(MFS_EINVAL, "NULL buffer");
    if (mode != MFS_MODE_RAW && !mean0) return fail(MFS_EINVAL, "mean0 is required in central / scaled mode");
    if (mode == MFS_MODE_SCALED && !scale0) return fail(MFS_EINVAL, "scale0 is required in scaled mode");
    if (B == 0) return MFS_OK;
    mfs::Filter1dGradLaunch launch = mfs::g_grad_table[N][n_par];
    if (!launch) return fail(MFS_EUNSUPPORTED, "no gradient kernel compiled for N = %d, n_par = %d", N, n_par);
    HIP_TRY(hipSetDevice(device));
    mfs::Lease lease(device);
    mfs::CallContext* cx = nullptr;
    hipError_t e = lease.context(&cx);
    if (e != hipSuccess) return fail(MFS_EHIP, "mfs_filter_1d_grad: %s", hipGetErrorString(e));
    hipStream_t s = stream ? (hipStream_t)stream : cx->compute;
    const size_t J1 = (size_t)model->degree + 1, ncoef = (size_t)model->n_rows * J1;
    const size_t nbc = model->coef_batched ? B : 1, nbl = model->lik_batched ? B : 1, nb = m0_batched ? B : 1, M2 = 2 * (size_t)N;
    double *d_coef = nullptr, *d_dcoef = nullptr, *d_lik = nullptr, *d_dlik = nullptr, *d_m0 = nullptr, *d_mean0 = nullptr,
           *d_scale0 = nullptr, *d_ys = nullptr, *d_nell = nullptr, *d_grad = nullptr;
    int32_t* d_fn = nullptr;
    auto alloc = [&](auto** d, size_t bytes) { if (e == hipSuccess) e = lease.device_block(d, bytes); };
    auto h2d = [&](void* d, const void* h, size_t bytes) {
        if (e == hipSuccess && bytes && h) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
    };
    alloc(&d_coef, nbc * ncoef * 8); alloc(&d_dcoef, nbc * n_par * ncoef * 8);
    alloc(&d_lik, nbl * model->n_lik * 8); alloc(&d_dlik, nbl * n_par * model->n_lik * 8);
    alloc(&d_m0, nb * M2 * 8); alloc(&d_mean0, nb * 8); alloc(&d_scale0, nb * 8);
    alloc(&d_ys, (size_t)B * T * 8); alloc(&d_nell, (size_t)B * 8); alloc(&d_grad, (size_t)B * n_par * 8);
    alloc(&d_fn, (size_t)B * 4);
    h2d(d_coef, model->coef, nbc * ncoef * 8); h2d(d_dcoef, dcoef, nbc * n_par * ncoef * 8);
    h2d(d_lik, model->lik, nbl * model->n_lik * 8); h2d(d_dlik, dlik, nbl * n_par * model->n_lik * 8);
    h2d(d_m0, m0, nb * M2 * 8); h2d(d_mean0, mean0, nb * 8); h2d(d_scale0, scale0, nb * 8);
    h2d(d_ys, ys, (size_t)B * T * 8);
    if (e == hipSuccess) {
        mfs::Filter1dGradArgs ga;
        memset(&ga, 0, sizeof(ga));
        mfs::Filter1dArgs& a = ga.f;
        a.mode = mode; a.T = T; a.B = B; a.t_begin = 0; a.t_end = T;
        a.trans_kind = model->trans_kind; a.umap = model->umap; a.n_terms = model->n_terms; a.degree = model->degree;
        a.n_rows = model->n_rows; a.coef_batched = model->coef_batched; a.lik_kind = model->lik_kind; a.n_lik = model->n_lik;
        a.lik_batched = model->lik_batched; a.mean_x_coef = model->mean_x_coef; a.coef = d_coef; a.lik = d_lik;
        a.m0 = d_m0; a.m0_batched = m0_batched; a.mean0 = d_mean0; a.scale0 = d_scale0; a.ys = d_ys;
        a.out_nell = d_nell; a.out_first_nan = d_fn;
        ga.n_par = n_par; ga.dcoef = d_dcoef; ga.dlik = d_dlik; ga.out_grad = d_grad;
        e = launch(ga, B, s);      // (the launcher knows its lanes per filter)
    }
    auto d2h = [&](void* h, const void* d, size_t bytes) {
        if (e == hipSuccess && h && bytes) e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    };
    d2h(out_nell, d_nell, (size_t)B * 8); d2h(out_grad, d_grad, (size_t)B * n_par * 8); d2h(out_first_nan, d_fn, (size_t)B * 4);
    const hipError_t es = hipStreamSynchronize(s);
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_filter_1d_grad: %s", hipGetErrorString(e));
    return MFS_OK;
}

int mfs_quadrature_1d(int N, int B, const double* ms, const double* mean, const double* scale, int stable,
                      double* out_weights, double* out_nodes, int device, void* stream) {
    if (N < 2 || N > MFS_MAX_N) return fail(MFS_EUNSUPPORTED, "N = %d outside [2, %d]", N, MFS_MAX_N);
    if (B < 0) return fail(MFS_EINVAL, "negative B");
    if (B == 0) return MFS_OK;
    if (!ms || !out_weights || !out_nodes) return fail(MFS_EINVAL, "NULL buffer");
    const int slot = pick_slot(N, stable);
    const mfs::KernelEntry& ke = mfs::g_table[N][slot];
    if (!ke.quad) return fail(MFS_EUNSUPPORTED, "no kernel compiled for N = %d", N);
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const int G = ke.lanes_per_filter;
    const int fpb = ke.waves_per_block * (64 / G);
    const bool ext = slot >= 3 && stable != 0;     // the completed rule of stable = 1 needs the extended tile
    if (ext && !mfs::g_quad_ext[N][slot - 3]) return fail(MFS_EUNSUPPORTED, "no extended quadrature kernel for N = %d", N);
    const int quad_lds = fpb * (ext ? mfs::g_quad_ext_lds[N][slot - 3] : slot >= 3 ? 2 * N : ke.lds_doubles_per_filter) * 8;
    double *d_ms = nullptr, *d_mean = nullptr, *d_scale = nullptr, *d_w = nullptr, *d_x = nullptr;
    mfs::Lease lease(device);
    hipError_t e = hipSuccess;
    auto alloc = [&](double** d, size_t bytes) { if (e == hipSuccess) e = lease.device_block(d, bytes); };
    alloc(&d_ms, (size_t)B * 2 * N * 8);
    alloc(&d_w, (size_t)B * N * 8);
    alloc(&d_x, (size_t)B * N * 8);
    if (mean) alloc(&d_mean, (size_t)B * 8);
    if (scale) alloc(&d_scale, (size_t)B * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ms, ms, (size_t)B * 2 * N * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && mean) e = hipMemcpyAsync(d_mean, mean, (size_t)B * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && scale) e = hipMemcpyAsync(d_scale, scale, (size_t)B * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        mfs::Quad1dArgs a{B, stable, d_ms, d_mean, d_scale, d_w, d_x};
        e = (ext ? mfs::g_quad_ext[N][slot - 3] : ke.quad)(a, (B + fpb - 1) / fpb, quad_lds, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_weights, d_w, (size_t)B * N * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(out_nodes, d_x, (size_t)B * N * 8, hipMemcpyDeviceToHost, s);
    const hipError_t es = hipStreamSynchronize(s);   // always: the blocks go back to the pool on return
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(MFS_EHIP, "mfs_quadrature_1d: %s", hipGetErrorString(e));
    return MFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// the staging pool (pool.hpp), as far as callers see it
// ---------------------------------------------------------------------------------------------------------------
int mfs_host_alloc(void** ptr, uint64_t bytes, int device) {
    if (!ptr) return fail(MFS_EINVAL, "ptr is NULL");
    *ptr = nullptr;
    if (device < 0 || device >= mfs::kMaxDevices) return fail(MFS_EINVAL, "device %d outside [0, %d)", device, mfs::kMaxDevices);
    HIP_TRY(hipSetDevice(device));
    hipError_t e = mfs::device_state(device).pinned.acquire(ptr, (size_t)bytes);
    if (e != hipSuccess) return fail(MFS_ENOMEM, "hipHostMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    return MFS_OK;
}

int mfs_host_free(void* ptr) {
    if (!ptr) return MFS_OK;
    for (int d = 0; d < mfs::kMaxDevices; ++d)
        if (mfs::device_state(d).pinned.release(ptr)) return MFS_OK;
    return fail(MFS_EINVAL, "pointer was not handed out by mfs_host_alloc");
}

int mfs_pool_trim(int device) {
    if (device < 0 || device >= mfs::kMaxDevices) return fail(MFS_EINVAL, "device %d outside [0, %d)", device, mfs::kMaxDevices);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    mfs::DeviceState& st = mfs::device_state(device);
    st.device.trim();
    st.pinned.trim();
    st.contexts.trim();
    return MFS_OK;
}

int mfs_pool_stats(int device, uint64_t* device_bytes, uint64_t* pinned_bytes, uint64_t* device_allocs, uint64_t* pinned_allocs) {
    if (device < 0 || device >= mfs::kMaxDevices) return fail(MFS_EINVAL, "device %d outside [0, %d)", device, mfs::kMaxDevices);
    mfs::DeviceState& st = mfs::device_state(device);
    const mfs::PoolCounters dc = st.device.counters(), pc = st.pinned.counters();
    if (device_bytes) *device_bytes = dc.cached_bytes;
    if (pinned_bytes) *pinned_bytes = pc.cached_bytes;
    if (device_allocs) *device_allocs = dc.fresh_allocs;
    if (pinned_allocs) *pinned_allocs = pc.fresh_allocs;
    return MFS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// RCCL all-gather of the per-replicate NLL vector.  librccl is dlopen'ed on first use so
// that single-GPU users never pay for loading it.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, mfs_rccl_id, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.handle) return MFS_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
    if (!h) return fail(MFS_ERCCL, "cannot dlopen librccl: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, mfs_rccl_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy)
        return fail(MFS_ERCCL, "librccl lacks an expected symbol");
    g_rccl.handle = h;
    return MFS_OK;
}
#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        int r_ = (expr);                                                                                   \
        if (r_ != 0) return fail(MFS_ERCCL, "%s failed: %s", #expr,                                        \
                                 g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error");        \
    } while (0)
}  // namespace

extern "C" {

int mfs_comm_unique_id(mfs_rccl_id* id) {
    if (!id) return fail(MFS_EINVAL, "id is NULL");
    if (int rc = load_rccl()) return rc;
    RCCL_TRY(g_rccl.GetUniqueId(id));
    return MFS_OK;
}

int mfs_comm_init(void** comm, const mfs_rccl_id* id, int nranks, int rank, int device) {
    if (!comm || !id) return fail(MFS_EINVAL, "NULL argument");
    if (rank < 0 || rank >= nranks) return fail(MFS_EINVAL, "rank %d outside [0, %d)", rank, nranks);
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(device));
    RCCL_TRY(g_rccl.CommInitRank(comm, nranks, *id, rank));
    return MFS_OK;
}

int mfs_allgather_nell(void* comm, const double* d_send, double* d_recv, uint64_t count, void* stream) {
    if (!comm || !d_send || !d_recv) return fail(MFS_EINVAL, "NULL argument");
    if (int rc = load_rccl()) return rc;
    RCCL_TRY(g_rccl.AllGather(d_send, d_recv, (size_t)count, 8 /* ncclFloat64 */, comm, (hipStream_t)stream));
    return MFS_OK;
}

int mfs_comm_destroy(void* comm) {
    if (!comm) return MFS_OK;
    if (int rc = load_rccl()) return rc;
    RCCL_TRY(g_rccl.CommDestroy(comm));
    return MFS_OK;
}

int mfs_memcpy_d2d(void* dst, const void* src, uint64_t bytes, void* stream) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MFS_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// N-D filter (d = 2)
// ---------------------------------------------------------------------------------------------------------------
struct mfs_plan_nd {
    int mode, N, T, B, stable, device, trans_kind, ny;
    bool hi_terms = false;    // operator table with |kappa| > 4 terms (TME order 3): the 29-row layout and its kernel
    mfs::FilterNdArgs args;   // model part filled at create (device pointers), data pointers per run
    double* d_coef = nullptr;
    double* d_lik = nullptr;
    int32_t* d_inds = nullptr;
};

static void destroy_plan_nd(mfs_plan_nd* p, bool quiesced) {
    hipSetDevice(p->device);
    if (!quiesced) hipDeviceSynchronize();   // pool blocks must be idle when they go back (what hipFree did implicitly)
    mfs::BlockPool<false>& pool = mfs::device_state(p->device).device;
    pool.release(p->d_coef); pool.release(p->d_lik); pool.release(p->d_inds);
    delete p;
}

extern "C" int mfs_plan_nd_create(mfs_plan_nd** plan, const mfs_model_nd* model, int mode, int N, int T, int B, int z,
                                  const int32_t* multi_indices, const int32_t* inds, int stable, int device) {
    if (!plan) return fail(MFS_EINVAL, "plan is NULL");
    *plan = nullptr;
    if (!model) return fail(MFS_EINVAL, "model is NULL");
    if (model->d != 2) return fail(MFS_EUNSUPPORTED, "the device N-D path supports d = 2 (got %d)", model->d);
    if (mode != MFS_MODE_RAW && mode != MFS_MODE_CENTRAL && mode != MFS_MODE_SCALED)
        return fail(MFS_EINVAL, "unknown moment mode %d", mode);
    if (N < 2 || N > 7) return fail(MFS_EUNSUPPORTED, "N = %d outside [2, 7] for d = 2", N);
    const mfs::NdEntry& ke = mfs::g_nd_table[N];
    if (!ke.launch) return fail(MFS_EUNSUPPORTED, "no N-D kernel compiled for N = %d", N);
    if (model->trans_kind != MFS_ND_TRANS_OPERATOR && model->trans_kind != MFS_ND_TRANS_GAUSSIAN)
        return fail(MFS_EINVAL, "unknown N-D transition kind %d", model->trans_kind);
    if (model->trans_kind == MFS_ND_TRANS_GAUSSIAN && model->n_terms != 5)
        return fail(MFS_EINVAL, "the Gaussian N-D transition carries 5 polynomials (mu_0, mu_1, S_00, S_01, S_11)");
    if (model->n_terms < 0 || model->n_terms > MFS_ND_TERMS_MAX) return fail(MFS_EINVAL, "bad n_terms %d", model->n_terms);
    const int n_rows = MFS_ND_TABLE_ROWS(model->n_terms);     // 16, or 29 when terms with |kappa| > 4 are present (TME order 3)
    if (z != ke.Z) return fail(MFS_EINVAL, "The size of multi_indices %d must match that of the moments %d.", z, ke.Z);
    const int max_extent = (model->trans_kind == MFS_ND_TRANS_OPERATOR && model->n_terms > MFS_ND_TERMS) ? MFS_ND_MAX_EXTENT_HI : MFS_ND_MAX_EXTENT;
    if (model->extent < 1 || model->extent > max_extent)
        return fail(MFS_EUNSUPPORTED, "coefficient extent %d outside [1, %d]", model->extent, max_extent);
    if (model->n_factors < 1 || model->n_factors > MFS_ND_MAX_FACTORS)
        return fail(MFS_EINVAL, "n_factors %d outside [1, %d]", model->n_factors, MFS_ND_MAX_FACTORS);
    if (model->ny < 1 || model->ny > 2) return fail(MFS_EINVAL, "ny %d outside [1, 2]", model->ny);
    for (int f = 0; f < model->n_factors; ++f) {
        const bool joint = model->fac_kind[f] == MFS_LIK_BEARING_GAUSSIAN;     // a factor of both components: component 2
        if (model->fac_kind[f] < 0 || model->fac_kind[f] > MFS_LIK_BEARING_GAUSSIAN || model->fac_n_par[f] < 1 ||
            model->fac_n_par[f] > MFS_MAX_LIK || model->fac_component[f] < 0 || model->fac_component[f] > 2 ||
            (model->fac_component[f] == 2) != joint || model->fac_ycol[f] < 0 || model->fac_ycol[f] >= model->ny)
            return fail(MFS_EINVAL, "bad description of likelihood factor %d", f);
        if (joint && (model->n_factors != 1 || (model->trans_kind == MFS_ND_TRANS_OPERATOR && (model->n_terms > MFS_ND_TERMS || N > 6))))
            return fail(MFS_EUNSUPPORTED, "a likelihood of both state components: one such factor, with a Normal-closure transition or "
                                          "operator tables of TME order <= 2 at N <= 6 (the other kernels' tiles have no room for the "
                                          "node tables)");
    }
    if (T < 0 || B < 0) return fail(MFS_EINVAL, "negative T or B");
    if (!multi_indices || !inds || !model->coef || !model->lik) return fail(MFS_EINVAL, "NULL buffer");
    {   // The kernels compute the gather of quadratures.py:151-152 arithmetically from the graded-lex order of
        // multi_indices.py:139-229 (d = 2: the index of (a0, a1) is (a0 + a1)(a0 + a1 + 1) / 2 + a0); a caller's table must be that one.
        const int S = ke.S;
        auto deg = [](int i) { int m = 0; while ((m + 1) * (m + 2) / 2 <= i) ++m; return m; };
        for (int t = 0; t < 3; ++t)
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < S; ++j) {
                    const int mi = deg(i), mj = deg(j), ui = i - mi * (mi + 1) / 2, uj = j - mj * (mj + 1) / 2;
                    const int M = mi + mj + (t > 0), want = M * (M + 1) / 2 + ui + uj + (t == 1);
                    if (inds[(size_t)t * S * S + i * S + j] != want)
                        return fail(MFS_EUNSUPPORTED, "inds[%d][%d][%d] = %d is not the graded-lexicographic Gram / Hankel table (expected %d)",
                                    t, i, j, inds[(size_t)t * S * S + i * S + j], want);
                }
    }
    // the kernel derives a moment's multi-index from its position: insist on the graded-lex table
    for (int s = 0, zi = 0; s < 2 * N; ++s)
        for (int n0 = 0; n0 <= s; ++n0, ++zi)
            if (multi_indices[2 * zi] != n0 || multi_indices[2 * zi + 1] != s - n0)
                return fail(MFS_EINVAL, "multi_indices is not the graded-lexicographic table of order 2N-1");
    HIP_TRY(hipSetDevice(device));
    mfs_plan_nd* p = new (std::nothrow) mfs_plan_nd();
    if (!p) return fail(MFS_ENOMEM, "out of host memory");
    p->mode = mode; p->N = N; p->T = T; p->B = B; p->stable = stable; p->device = device;
    p->trans_kind = model->trans_kind; p->ny = model->ny; p->hi_terms = (model->trans_kind == MFS_ND_TRANS_OPERATOR && model->n_terms > MFS_ND_TERMS);
    const size_t S = ke.S, DD = (size_t)model->extent * model->extent;
    const size_t ncoef = (size_t)(model->coef_batched ? B : 1) * n_rows * DD;
    const size_t nlik = (size_t)(model->lik_batched ? B : 1) * model->n_factors * MFS_MAX_LIK;
    mfs::BlockPool<false>& pool = mfs::device_state(device).device;
    hipError_t e = pool.acquire((void**)&p->d_coef, ncoef * 8);
    if (e == hipSuccess) e = pool.acquire((void**)&p->d_lik, nlik * 8);
    if (e == hipSuccess) e = pool.acquire((void**)&p->d_inds, 3 * S * S * 4);
    if (e == hipSuccess && ncoef) e = hipMemcpy(p->d_coef, model->coef, ncoef * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && nlik) e = hipMemcpy(p->d_lik, model->lik, nlik * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_inds, inds, 3 * S * S * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        destroy_plan_nd(p, true);
        return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_plan_nd_create: %s", hipGetErrorString(e));
    }
    mfs::FilterNdArgs& a = p->args;
    memset(&a, 0, sizeof(a));
    a.mode = mode; a.T = T; a.B = B; a.stable = stable;
    a.n_terms_used = model->n_terms; a.D = model->extent;
    a.n_factors = model->n_factors; a.ny = model->ny;
    for (int f = 0; f < model->n_factors; ++f) {
        a.fac_kind[f] = model->fac_kind[f]; a.fac_comp[f] = model->fac_component[f]; a.fac_ycol[f] = model->fac_ycol[f];
    }
    a.coef_batched = model->coef_batched; a.lik_batched = model->lik_batched;
    // stable = 1: the completion on the register front end (a.stable = 1); MFS_ND_STABLE=dense keeps the LDS-tile form (2)
    if (stable) { const char* e = getenv("MFS_ND_STABLE"); a.stable = (e && strcmp(e, "dense") == 0) ? 2 : 1; }
    if (const char* e = getenv("MFS_ND_UPDATE")) {   // A/B switches, like MFS_SOLVER
        a.force_eigen = (strcmp(e, "eigen") == 0);
        a.joint_grid = (strcmp(e, "grid") == 0);
    }
    // true extents of each coefficient block (trailing zero rows / columns cut); the union over replicates when batched
    const size_t ntab = model->coef_batched ? (size_t)B : 1;
    for (int k = 0; k < n_rows; ++k) {
        int ea = 0, eb = 0;
        for (size_t r = 0; r < ntab; ++r) {
            const double* blk = model->coef + (r * n_rows + k) * DD;
            for (int i = 0; i < model->extent; ++i)
                for (int j = 0; j < model->extent; ++j)
                    if (blk[i * model->extent + j] != 0.0) { if (i + 1 > ea) ea = i + 1; if (j + 1 > eb) eb = j + 1; }
        }
        a.ext[k] = (ea == 0) ? 0 : (ea | (eb << 8));
    }
    a.coef = p->d_coef; a.lik = p->d_lik; a.inds = p->d_inds;
    *plan = p;
    return MFS_OK;
}

// one launch over the steps [t0, t1) of the plan's run; `carry` ([B][carry_doubles], or null for a single launch over [0, T))
// takes the state from one chunk to the next
static int plan_nd_launch(mfs_plan_nd* p, const double* d_m0, int m0_batched, const double* d_mean0, const double* d_scale0,
                          const double* d_ys, double* d_out_moments, double* d_out_means, double* d_out_scales,
                          double* d_out_nell, int32_t* d_out_first_nan, int t0, int t1, double* carry, hipStream_t stream) {
    mfs::FilterNdArgs a = p->args;
    a.m0 = d_m0; a.m0_batched = m0_batched; a.mean0 = d_mean0; a.scale0 = d_scale0; a.ys = d_ys;
    a.out_mom = d_out_moments;
    a.out_mean = (p->mode != MFS_MODE_RAW) ? d_out_means : nullptr;
    a.out_scale = (p->mode == MFS_MODE_SCALED) ? d_out_scales : nullptr;
    a.out_nell = d_out_nell; a.out_first_nan = d_out_first_nan;
    a.t_begin = t0; a.t_end = t1; a.carry = carry;
    const mfs::NdEntry& ke = mfs::g_nd_table[p->N];
    const bool joint = a.n_factors == 1 && a.fac_comp[0] == 2;
    hipError_t e = (p->trans_kind == MFS_ND_TRANS_GAUSSIAN ? ke.launch_gauss : p->hi_terms ? ke.launch_hi
                    : joint ? ke.launch_joint : ke.launch)(a, p->B, stream);
    if (e != hipSuccess) return fail(MFS_EHIP, "N-D kernel launch: %s", hipGetErrorString(e));
    return MFS_OK;
}

extern "C" int mfs_plan_nd_run(mfs_plan_nd* p, const double* d_m0, int m0_batched, const double* d_mean0,
                               const double* d_scale0, const double* d_ys, double* d_out_moments, double* d_out_means,
                               double* d_out_scales, double* d_out_nell, int32_t* d_out_first_nan, void* stream) {
    if (!p) return fail(MFS_EINVAL, "plan is NULL");
    if (p->B == 0) return MFS_OK;
    if (!d_m0 || !d_out_nell || (p->T > 0 && !d_ys)) return fail(MFS_EINVAL, "NULL buffer");
    if (p->mode != MFS_MODE_RAW && !d_mean0) return fail(MFS_EINVAL, "mean0 is required in central and scaled modes");
    if (p->mode == MFS_MODE_SCALED && !d_scale0) return fail(MFS_EINVAL, "scale0 is required in scaled mode");
    HIP_TRY(hipSetDevice(p->device));
    return plan_nd_launch(p, d_m0, m0_batched, d_mean0, d_scale0, d_ys, d_out_moments, d_out_means, d_out_scales, d_out_nell,
                          d_out_first_nan, 0, p->T, nullptr, (hipStream_t)stream);
}

extern "C" int mfs_plan_nd_destroy(mfs_plan_nd* p) {
    if (p) destroy_plan_nd(p, false);
    return MFS_OK;
}

extern "C" int mfs_plan_nd_geometry(const mfs_plan_nd* p, int* threads_per_filter, int* grid, int* lds_bytes_per_block) {
    if (!p) return fail(MFS_EINVAL, "plan is NULL");
    if (threads_per_filter) *threads_per_filter = 256;
    if (grid) *grid = p->B;
    if (lds_bytes_per_block) *lds_bytes_per_block = mfs::g_nd_table[p->N].lds_bytes;
    return MFS_OK;
}

extern "C" int mfs_filter_nd(const mfs_model_nd* model, int mode, int N, int T, int B, int z,
                             const int32_t* multi_indices, const int32_t* inds, const double* m0, int m0_batched,
                             const double* mean0, const double* scale0, const double* ys, int stable,
                             double* out_moments, double* out_means, double* out_scales, double* out_nell,
                             int32_t* out_first_nan, int device, void* stream) {
    mfs_plan_nd* plan = nullptr;
    int rc = mfs_plan_nd_create(&plan, model, mode, N, T, B, z, multi_indices, inds, stable, device);
    if (rc != MFS_OK) return rc;
    struct Guard { mfs_plan_nd* p; ~Guard() { destroy_plan_nd(p, true); } } guard{plan};   // (every exit below is quiesced)
    if (!m0 || !out_nell || (T > 0 && B > 0 && !ys)) return fail(MFS_EINVAL, "NULL buffer");
    if (mode != MFS_MODE_RAW && !mean0) return fail(MFS_EINVAL, "mean0 is required in central and scaled modes");
    if (mode == MFS_MODE_SCALED && !scale0) return fail(MFS_EINVAL, "scale0 is required in scaled mode");
    if (B == 0) return MFS_OK;
    const size_t Z = (size_t)z, nb = m0_batched ? B : 1, ny = (size_t)model->ny;
    double *d_m0 = nullptr, *d_mean0 = nullptr, *d_ys = nullptr, *d_mom = nullptr, *d_means = nullptr, *d_nell = nullptr,
           *d_scale0 = nullptr, *d_scales = nullptr;
    int32_t* d_fn = nullptr;
    mfs::Lease lease(device);
    mfs::CallContext* cx = nullptr;
    hipError_t e = lease.context(&cx);
    if (e != hipSuccess) return fail(MFS_EHIP, "mfs_filter_nd: %s", hipGetErrorString(e));
    hipStream_t s = stream ? (hipStream_t)stream : cx->compute;
    auto alloc = [&](auto** d, size_t bytes) { if (e == hipSuccess) e = lease.device_block(d, bytes); };
    auto h2d = [&](void* d, const void* h, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s);
    };
    auto d2h = [&](void* h, const void* d, size_t bytes) {
        if (e == hipSuccess && h && d && bytes) e = hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s);
    };
    alloc(&d_m0, nb * Z * 8);
    alloc(&d_mean0, nb * 2 * 8);
    alloc(&d_scale0, nb * 2 * 8);
    alloc(&d_ys, (size_t)B * T * ny * 8);
    if (out_moments) alloc(&d_mom, (size_t)B * T * Z * 8);
    if (out_means && mode != MFS_MODE_RAW) alloc(&d_means, (size_t)B * T * 2 * 8);
    if (out_scales && mode == MFS_MODE_SCALED) alloc(&d_scales, (size_t)B * T * 2 * 8);
    alloc(&d_nell, (size_t)B * 8);
    alloc(&d_fn, (size_t)B * 4);
    h2d(d_m0, m0, nb * Z * 8);
    if (mean0) h2d(d_mean0, mean0, nb * 2 * 8);
    if (scale0 && mode == MFS_MODE_SCALED) h2d(d_scale0, scale0, nb * 2 * 8);
    h2d(d_ys, ys, (size_t)B * T * ny * 8);
    // with the moments streamed out, T is cut into chunks: chunk k's slice travels to the host (2-D copy: B rows of
    // chunk x z doubles) on the copy stream while the kernel of chunk k + 1 runs; the per-replicate state crosses the
    // launches through a carry block, so the bits are those of the single launch
    const size_t mom_bytes = out_moments ? (size_t)B * T * Z * 8 : 0;
    const int nchunks = (T > 0 && pitch_ok((size_t)T * Z * 8, device)) ? host_chunks(mom_bytes, T) : 1;
    const int chunk = (nchunks > 1) ? (T + nchunks - 1) / nchunks : T;
    double* d_carry = nullptr;
    if (nchunks > 1) alloc(&d_carry, (size_t)B * mfs::g_nd_table[N].carry_doubles * 8);
    if (e == hipSuccess) {
        if (nchunks <= 1) {
            rc = mfs_plan_nd_run(plan, d_m0, m0_batched, d_mean0, d_scale0, d_ys, d_mom, d_means, d_scales, d_nell, d_fn, s);
            if (rc == MFS_OK) d2h(out_moments, d_mom, mom_bytes);
        } else {
            const size_t pitch = (size_t)T * Z * 8;
            for (int k = 0, t0 = 0; t0 < T && rc == MFS_OK && e == hipSuccess; ++k, t0 += chunk) {
                const int t1 = (t0 + chunk < T) ? t0 + chunk : T;
                rc = plan_nd_launch(plan, d_m0, m0_batched, d_mean0, d_scale0, d_ys, d_mom, d_means, d_scales, d_nell, d_fn,
                                    t0, t1, d_carry, s);
                if (rc != MFS_OK) break;
                e = hipEventRecord(cx->ev[k], s);
                if (e == hipSuccess) e = hipStreamWaitEvent(cx->copy, cx->ev[k], 0);
                if (e == hipSuccess && out_moments)
                    e = hipMemcpy2DAsync(out_moments + (size_t)t0 * Z, pitch, d_mom + (size_t)t0 * Z, pitch,
                                         (size_t)(t1 - t0) * Z * 8, (size_t)B, hipMemcpyDeviceToHost, cx->copy);
            }
        }
    }
    if (rc == MFS_OK) {
        d2h(out_means, d_means, (size_t)B * T * 2 * 8);
        d2h(out_scales, d_scales, (size_t)B * T * 2 * 8);
        d2h(out_nell, d_nell, (size_t)B * 8);
        d2h(out_first_nan, d_fn, (size_t)B * 4);
    }
    const hipError_t es = hipStreamSynchronize(s), ec = hipStreamSynchronize(cx->copy);   // always: the pool blocks go back on return
    if (e == hipSuccess) e = (es != hipSuccess) ? es : ec;
    if (rc != MFS_OK) return rc;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_filter_nd: %s", hipGetErrorString(e));
    return MFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// diagnostic: the kernels' elementary functions
// ---------------------------------------------------------------------------------------------------------------
extern "C" int mfs_elementary(int which, int n, const double* x, double* out, int device) {
    if (which < 0 || which > 2) return fail(MFS_EINVAL, "which = %d outside {0 exp, 1 tanh, 2 log}", which);
    if (n < 0) return fail(MFS_EINVAL, "negative n");
    if (n == 0) return MFS_OK;
    if (!x || !out) return fail(MFS_EINVAL, "NULL buffer");
    HIP_TRY(hipSetDevice(device));
    double *d_x = nullptr, *d_o = nullptr;
    mfs::Lease lease(device);
    hipError_t e = lease.device_block(&d_x, (size_t)n * 8);
    if (e == hipSuccess) e = lease.device_block(&d_o, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(d_x, x, (size_t)n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = mfs::launch_elementary(which, n, d_x, d_o, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipDeviceSynchronize();
    if (e != hipSuccess) return fail(MFS_EHIP, "mfs_elementary: %s", hipGetErrorString(e));
    return MFS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// characteristic function from moments, host pointers
// ---------------------------------------------------------------------------------------------------------------
extern "C" int mfs_characteristic_1d(int N, int count, const double* ms, const double* mean, const double* scale,
                                     int nz, const double* zs, double* out, int device, void* stream) {
    if (N < 2 || N > MFS_MAX_N) return fail(MFS_EUNSUPPORTED, "N = %d outside [2, %d]", N, MFS_MAX_N);
    if (count < 0 || nz < 0) return fail(MFS_EINVAL, "negative count or nz");
    if (count == 0 || nz == 0) return MFS_OK;
    if (!ms || !zs || !out) return fail(MFS_EINVAL, "NULL buffer");
    const int gi = (N + 1 <= 8) ? 3 : (N + 1 <= 16) ? 0 : (N + 1 <= 32) ? 1 : 2;
    mfs::Cf1dLaunch launch = mfs::g_cf[N][gi];
    if (!launch) return fail(MFS_EUNSUPPORTED, "no kernel compiled for N = %d", N);
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    const int G = (gi == 3) ? 8 : (gi == 0) ? 16 : (gi == 1) ? 32 : 64, fpb = 64 / G;
    double *d_ms = nullptr, *d_mean = nullptr, *d_scale = nullptr, *d_zs = nullptr, *d_out = nullptr;
    hipError_t e = hipSuccess;
    mfs::Lease lease(device);
    auto alloc = [&](void** d, size_t bytes) { if (e == hipSuccess) e = lease.device_block((char**)d, bytes); };
    alloc((void**)&d_ms, (size_t)count * 2 * N * 8);
    alloc((void**)&d_zs, (size_t)nz * 8);
    alloc((void**)&d_out, (size_t)count * nz * 16);
    if (mean) alloc((void**)&d_mean, (size_t)count * 8);
    if (scale) alloc((void**)&d_scale, (size_t)count * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(d_ms, ms, (size_t)count * 2 * N * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_zs, zs, (size_t)nz * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && mean) e = hipMemcpyAsync(d_mean, mean, (size_t)count * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess && scale) e = hipMemcpyAsync(d_scale, scale, (size_t)count * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        mfs::Cf1dArgs a{count, nz, d_ms, d_mean, d_scale, d_zs, d_out};
        e = launch(a, (count + fpb - 1) / fpb, fpb * 2 * N * 8, s);
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)count * nz * 16, hipMemcpyDeviceToHost, s);
    const hipError_t es = hipStreamSynchronize(s);   // always: the pool blocks go back on return
    if (e == hipSuccess) e = es;
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? MFS_ENOMEM : MFS_EHIP, "mfs_characteristic_1d: %s", hipGetErrorString(e));
    return MFS_OK;
}
