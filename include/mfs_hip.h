/*
 * mfs_hip.h -- C ABI of libmfs_hip.so: the MI355X (gfx950) moment-filter hot path of zgbkdlm/mfs.
 *
 * This is the drop-in boundary.  The reference is pure Python/JAX and has no FFI of its own; the entry points
 * below are what a ctypes binding for its filter loop binds (INTEGRATION.md shows the stub).  Each entry point
 * names the reference interface it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; all floating point is IEEE fp64 (the reference sets jax_enable_x64).
 *   - every function returns MFS_OK (0) or a negative MFS_E* code; mfs_last_error() gives the message of the
 *     calling thread's last failure.  Numerical failure is NEVER an error code: a non-positive-definite moment
 *     matrix poisons that replicate with NaN in-band, exactly as the reference's XLA Cholesky does, and the first
 *     poisoned step is reported in out_first_nan.
 *   - "host" entry points take host pointers and stage through device buffers owned by the library;
 *     "_dev" entry points take device pointers (inputs already resident in HBM) and only enqueue work on `stream`.
 *   - batch axis: B independent replicates (Monte-Carlo keys / parameter points; the reference runs these as
 *     separate calls, dardel/benes_bernoulli/mf.py:70-92).  Layouts are row-major with the replicate axis first.
 */
#ifndef MFS_HIP_H
#define MFS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFS_ABI_VERSION 2   /* 2: mfs_model_nd carries likelihood factors, ny and per-replicate tables; staging pool */

/* return codes */
#define MFS_OK 0
#define MFS_EINVAL (-1)   /* bad argument (mirrors the reference's only raise, multi_dims/filtering.py:159-161) */
#define MFS_EUNSUPPORTED (-2) /* N / degree / model outside what the kernels are compiled for */
#define MFS_EHIP (-3)     /* a HIP runtime call failed (message in mfs_last_error) */
#define MFS_ENOMEM (-4)
#define MFS_ERCCL (-5)

/* moment representation, mfs/one_dim/filtering.py:32 / :92 / :164 */
#define MFS_MODE_RAW 0
#define MFS_MODE_CENTRAL 1
#define MFS_MODE_SCALED 2
/* OR-ed into `mode`: the moment vectors carry 2N + 1 entries instead of 2N.  The reference warns about an odd count and
 * proceeds (mfs/one_dim/filtering.py:65-66): N = floor(M / 2) (mfs/one_dim/quadtures.py:122), every rule is built from
 * the first 2N entries, and the last entry of each output row is the N-node rule's value of the order-2N moment.  m0 and
 * out_moments then have rows of 2N + 1 doubles; the run uses the dense kernel in one launch. */
#define MFS_MODE_ODD_TAIL 0x100

/* how the transition moments E[(X_k - c)^n | X_{k-1} = x] are produced on the device */
#define MFS_TRANS_OPERATOR 0 /* TME without closure: sum_{k<=K} Q_k(u(x)) n!/(n-k)! (x - c)^(n-k);
                                replaces sde_cond_moments_tme, mfs/one_dim/moments.py:141-179 */
#define MFS_TRANS_GAUSSIAN 1 /* normal closure from (mu(x), var(x)): TME-normal (:182-219), Euler (:222-255), or an
                                exact linear-Gaussian step (dardel/convergence/convergence_mf.py:86-107) */

/* the variable the coefficient tables are polynomials in */
#define MFS_U_IDENTITY 0 /* u = x        (polynomial drift, e.g. well_poisson, OU) */
#define MFS_U_TANH 1     /* u = tanh(x)  (benes_bernoulli, mfs/one_dim/ss_models.py:37-38) */

/* measurement likelihood p(y | x) */
#define MFS_LIK_BERNOULLI_LOGISTIC 0 /* p = 1/(1+exp(-(l0 + l1 x + l2 x^2 + l3 x^3))), pmf(y; p), y in {0,1}
                                        (ss_models.py:43-47; multi_dims/ss_models.py:63-67 on x_0) */
#define MFS_LIK_POISSON_SOFTPLUS 1   /* rate = log(1 + exp(l0 x)), Poisson pmf(y; rate) (ss_models.py:80-84) */
#define MFS_LIK_GAUSSIAN 2           /* y ~ N(l0 x + l1, l2) (l2 = variance) (convergence_mf.py:58-61) */
#define MFS_LIK_BEARING_GAUSSIAN 3   /* N-D only, a factor of BOTH state components (fac_component = 2): y ~ N(atan2(x_1, x_0), l0),
                                        l0 = variance (examples/2d_bearing_only.ipynb cell 7); not with TME-order-3 operator tables */

#define MFS_MAX_N 32       /* quadrature order N (2N moments) */
#define MFS_MAX_TERMS 8    /* K <= 2 * tme_order */
#define MFS_MAX_DEGREE 15  /* polynomial degree of a coefficient row */
#define MFS_MAX_LIK 4

/*
 * 1-D model descriptor.  Replaces the Python callables the reference's filters take
 * (state_cond_*_moments, state_cond_mean[_var], measurement_cond_pdf; mfs/one_dim/filtering.py:32-36,92-98,164-172):
 * callables cannot cross a C ABI, so the host side reduces a model to coefficient tables (mfs_amd/tme_poly.py).
 *
 * coef is [n_rows][degree + 1] (ascending powers of u), or [B][n_rows][degree + 1] when coef_batched:
 *   MFS_TRANS_OPERATOR: rows 0..K-1 = Q_1..Q_K (Q_0 = 1 is implicit); row K = conditional variance polynomial
 *                       (tme.mean_and_cov, used by the scaled mode only).  Conditional mean = x + Q_1(u).
 *                       n_rows = K + 1.
 *   MFS_TRANS_GAUSSIAN: row 0 = P_m, row 1 = P_v with mu(x) = mean_x_coef * x + P_m(u), var(x) = P_v(u).
 *                       n_rows = 2.
 */
typedef struct mfs_model_1d {
    int32_t trans_kind;   /* MFS_TRANS_* */
    int32_t umap;         /* MFS_U_* */
    int32_t n_terms;      /* K (MFS_TRANS_OPERATOR); 0 otherwise */
    int32_t degree;       /* J <= MFS_MAX_DEGREE */
    int32_t n_rows;       /* rows per table */
    int32_t coef_batched; /* 0: one table for all replicates; 1: one per replicate */
    int32_t lik_kind;     /* MFS_LIK_* */
    int32_t n_lik;        /* <= MFS_MAX_LIK */
    int32_t lik_batched;  /* 0 / 1 */
    int32_t reserved;
    double mean_x_coef;   /* MFS_TRANS_GAUSSIAN only */
    const double* coef;
    const double* lik;    /* [n_lik] or [B][n_lik] */
} mfs_model_1d;

/* ---- library / device management --------------------------------------------------------------------------- */
int mfs_version(void);                 /* MFS_ABI_VERSION */
const char* mfs_last_error(void);      /* thread-local message of the last failing call ("" if none) */
int mfs_device_count(int* count);
int mfs_set_device(int device);
int mfs_device_synchronize(void);
int mfs_device_name(int device, char* buf, int buflen);

/* device memory and streams for callers that keep data resident (bench harness, sharded driver) */
int mfs_malloc(void** dptr, uint64_t bytes);
int mfs_free(void* dptr);
int mfs_memcpy_h2d(void* dst, const void* src, uint64_t bytes, void* stream);
int mfs_memcpy_d2h(void* dst, const void* src, uint64_t bytes, void* stream);
int mfs_memset(void* dst, int value, uint64_t bytes, void* stream);
int mfs_stream_create(void** stream);
int mfs_stream_destroy(void* stream);
int mfs_stream_synchronize(void* stream);
/*
 * The library's staging pool (SURVEY.md section 8b "Ownership": the caller owns what it passes; the library stages
 * through its own device buffers and pinned pool, reused across calls).  The host-pointer entry points below borrow
 * their device buffers, streams and events from per-device caches, so a steady-state call makes no hipMalloc / hipFree.
 * mfs_host_alloc hands out page-locked host memory from the same pool: results written into it (e.g. out_moments)
 * travel at the PCIe rate instead of the pageable-copy rate, and a block given back with mfs_host_free is reused by
 * the next request of similar size.  mfs_pool_trim returns every unused block (device, pinned, streams) to the driver;
 * mfs_pool_stats reports bytes held and the number of driver allocations made so far (NULL = not wanted).
 */
int mfs_host_alloc(void** ptr, uint64_t bytes, int device);
int mfs_host_free(void* ptr);
int mfs_pool_trim(int device);
int mfs_pool_stats(int device, uint64_t* device_bytes, uint64_t* pinned_bytes, uint64_t* device_allocs,
                   uint64_t* pinned_allocs);
/* HIP-event timing on `stream` (bench.py's live per-launch kernel time) */
int mfs_event_create(void** event);
int mfs_event_destroy(void* event);
int mfs_event_record(void* event, void* stream);
int mfs_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */

/*
 * ---- 1-D moment filter, host pointers ------------------------------------------------------------------------
 * Replaces moment_filter_rms / moment_filter_cms / moment_filter_scms (mfs/one_dim/filtering.py:32-89, 92-161,
 * 164-240) for B replicates at once.
 *
 *   mode        MFS_MODE_*
 *   N           quadrature order; the moment vectors have 2N entries (orders 0..2N-1), 2 <= N <= MFS_MAX_N
 *   T, B        time steps, replicates
 *   m0          initial moments, [2N] (m0_batched = 0) or [B][2N]
 *   mean0       [1] or [B] (same batching as m0); ignored in raw mode (may be NULL)
 *   scale0      likewise; scaled mode only
 *   ys          [B][T], measurements as doubles (Bernoulli y in {0., 1.})
 *   stable      0 = Cholesky; 1 = LDL^T completion (mfs/utils.py:495-538), on the same register-resident kernel (the completed
 *               rule stays tridiagonal)
 *   out_moments [B][T][2N]   out_means [B][T] (NULL in raw mode)   out_scales [B][T] (scaled mode, else NULL)
 *   out_nell    [B]          out_first_nan [B]: first step whose outputs are non-finite, -1 if none (may be NULL)
 *   device      HIP device ordinal; stream: hipStream_t or NULL (a stream of the library's).  Returns after the
 *               results are in the host buffers.
 * When out_moments is requested and large, the run is split into T-chunks (carry state in HBM, bit-identical to one
 * launch) and each chunk's slice of the moments is copied out while the next chunk computes (MFS_HOST_CHUNKS=n
 * overrides the chunk count, 1 = no pipelining).  Per-replicate parameters theta arrive as
 * batched model tables (model->coef_batched / lik_batched), not as a separate argument.
 */
int mfs_filter_1d(const mfs_model_1d* model, int mode, int N, int T, int B,
                  const double* m0, int m0_batched, const double* mean0, const double* scale0,
                  const double* ys, int stable,
                  double* out_moments, double* out_means, double* out_scales, double* out_nell,
                  int32_t* out_first_nan, int device, void* stream);

/*
 * ---- 1-D moment filter, device-resident plan -----------------------------------------------------------------
 * The same computation with every buffer already in HBM.  A plan uploads the model tables once, owns the carry
 * state, splits T into chunks of `chunk` steps (0 = whole T in one launch) and captures the chunk launches into a
 * hipGraph so that a run is one graph launch.  All pointers passed to mfs_plan_1d_run are DEVICE pointers with the
 * layouts documented for mfs_filter_1d; out_moments may be NULL ("NLL only", the parameter-estimation grid of
 * dardel/parameter_estimation/mf.py:37-54 needs nothing else).
 */
typedef struct mfs_plan_1d mfs_plan_1d;

int mfs_plan_1d_create(mfs_plan_1d** plan, const mfs_model_1d* model /* host pointers inside */, int mode, int N,
                       int T, int B, int stable, int chunk, int device);
int mfs_plan_1d_run(mfs_plan_1d* plan, const double* d_m0, int m0_batched, const double* d_mean0,
                    const double* d_scale0, const double* d_ys, double* d_out_moments, double* d_out_means,
                    double* d_out_scales, double* d_out_nell, int32_t* d_out_first_nan, void* stream);
int mfs_plan_1d_destroy(mfs_plan_1d* plan);
/* launch geometry actually used (for DESIGN.md / the bench JSON): lanes per filter, filters per block, grid size */
int mfs_plan_1d_geometry(const mfs_plan_1d* plan, int* lanes_per_filter, int* filters_per_block, int* grid,
                         int* lds_bytes_per_block);

/*
 * ---- negative log-likelihood and its gradient, forward mode inside the time loop ------------------------------------
 * Replaces what dardel/parameter_estimation/mf.py:37-54,70-73 obtains by differentiating obj_func (moment_filter_cms
 * under jax.jit) through the lax.scan with JAX autodiff for jaxopt.ScipyMinimize(L-BFGS-B): the filter's state -- moments,
 * mean, scale, nell -- is carried as dual numbers (value + n_par tangents) through every Cholesky pivot, eigenvalue,
 * weight and quadrature sum (mfs_amd/csrc/filter1d_grad.hpp); no reverse pass.
 *
 *   model       the value tables, as for mfs_filter_1d (coef_batched / lik_batched: one parameter point per replicate)
 *   dcoef       d coef / d theta_p: [n_par][n_rows][degree + 1], or [B][n_par][n_rows][degree + 1] when coef_batched
 *   dlik        d lik / d theta_p:  [n_par][n_lik], or [B][n_par][n_lik] when lik_batched
 *   n_par       1 .. 4;  N  2 .. 10;  m0, mean0, scale0, ys as for mfs_filter_1d (the initial law does not depend on theta)
 *   out_nell    [B];  out_grad [B][n_par] = d nell / d theta_p;  out_first_nan [B] or NULL.  Host pointers.
 * A replicate that NaN-poisons returns NaN in both, as the reference's objective would.
 */
int mfs_filter_1d_grad(const mfs_model_1d* model, const double* dcoef, const double* dlik, int n_par, int mode, int N,
                       int T, int B, const double* m0, int m0_batched, const double* mean0, const double* scale0,
                       const double* ys, double* out_nell, double* out_grad, int32_t* out_first_nan, int device,
                       void* stream);

/*
 * ---- quadrature only -----------------------------------------------------------------------------------------
 * Replaces moment_quadrature (mfs/one_dim/quadtures.py:83-133) for B moment vectors: ms [B][2N], mean/scale [B] or
 * NULL (0 / 1), out weights/nodes [B][N].  Host pointers.  Used by the parity tests to check the Cholesky /
 * triangular-solve / eigensolve stage in isolation, and by characteristic-function post-processing
 * (mfs/one_dim/moments.py:309-337).
 */
int mfs_quadrature_1d(int N, int B, const double* ms, const double* mean, const double* scale, int stable,
                      double* out_weights, double* out_nodes, int device, void* stream);

/*
 * ---- characteristic function from moments ------------------------------------------------------------------------
 * Replaces characteristic_fn (mfs/one_dim/moments.py:309-337) as the post-processing drivers call it
 * (dardel/benes_bernoulli/post_processing_mf.py:37-60: vmapped over the z grid and over every filtering step):
 * out[c][k] = sum_n w_n exp(i zs[k] x_n) with (w, x) the quadrature of ms[c] (mean / scale [count] or NULL).
 * ms [count][2N]; zs [nz]; out [count][nz][2] = (re, im) pairs, i.e. a C-contiguous complex128 array.  Host pointers.
 */
int mfs_characteristic_1d(int N, int count, const double* ms, const double* mean, const double* scale, int nz,
                          const double* zs, double* out, int device, void* stream);

/*
 * ---- diagnostic: the kernels' own elementary functions -------------------------------------------------------------
 * The 1-D kernel evaluates exp / tanh / log with short in-line routines instead of the ocml ones (the reference uses
 * jnp.exp / jnp.tanh / jnp.log inside its model callables, mfs/one_dim/ss_models.py:37-47).  This entry point applies
 * them to an array so that the tests can bound their error against libm.  which: 0 exp, 1 tanh, 2 log.  Host pointers.
 */
int mfs_elementary(int which, int n, const double* x, double* out, int device);

/*
 * ---- N-D moment filter (d = 2), host pointers ------------------------------------------------------------------
 * Replaces moment_filter_nd_rms / moment_filter_nd_cms / moment_filter_nd_scms (mfs/multi_dims/filtering.py:283-344,
 * 210-280, 33-207) for B replicates, with either transition family the reference offers.
 *
 * MFS_ND_TRANS_OPERATOR ('multi-index' signature, sde_cond_moments_tme, mfs/multi_dims/moments.py:414-479):
 * polynomial drift / dispersion reduced on the host to the operator table Q_kappa(x), 1 <= |kappa| <= 2 M for TME order
 * M <= 3, dense per-variable extent D, in graded-lex kappa order (0,1),(1,0),(0,2),(1,1),(2,0),(0,3),..., zeros where the
 * model has no term.  Conditional mean_k = x_k + Q_{e_k}.  Two layouts, chosen by n_terms (MFS_ND_TABLE_ROWS):
 *   n_terms <= MFS_ND_TERMS (|kappa| <= 4, TME order <= 2): coef [16][D][D], rows 14, 15 the conditional variances of
 *     X'_0, X'_1 (diagonal of tme.mean_and_cov, moments.py:469-476; read in scaled mode only);
 *   n_terms <= MFS_ND_TERMS_MAX (|kappa| <= 6, TME order 3): coef [29][D][D], rows 27, 28 the variances.
 *
 * MFS_ND_TRANS_GAUSSIAN ('index' signature, the Normal closures sde_cond_moments_tme_normal / _euler_maruyama,
 * mfs/multi_dims/moments.py:340-411, 257-337, whose moments the reference takes from Kan's formula, :110-154):
 * X' | x ~ N(mu(x), S(x)); rows 0..4 of coef hold the polynomials mu_0, mu_1, S_00, S_01, S_11 (n_terms = 5), the
 * other rows are ignored.  The kernel evaluates E[(X'_0-c_0)^a (X'_1-c_1)^b] by the Stein recursion, which is the
 * same polynomial in (mu - c, S) as Kan's sum.
 * The likelihood is a product of up to MFS_ND_MAX_FACTORS factors, each a function of ONE state component and one
 * column of the measurement: p(y | x) = prod_f lik(kind_f, params_f, y[ycol_f], x[component_f]).  One factor on x_0 is the
 * prey--predator model (mfs/multi_dims/ss_models.py:63-67); two Gaussian factors, one per component, are the reference's
 * measurement_cond_pdf_2d (tests/test_filtering.py:36-46: ys_2d of shape (T, 2), prod(norm.pdf(y, x, sd))).
 *
 *   N              quadrature order per dimension: s = N(N+1)/2 Gram size, z = N(2N+1) moments (|n| <= 2N-1), 2..7
 *   multi_indices  [z][2] int32, must equal the graded-lex table (checked: MFS_EINVAL otherwise, mirroring the
 *                  reference's only raise, multi_dims/filtering.py:238-239)
 *   inds           [3][s][s] int32 Gram / Hankel gather tables (gram_and_hankel_indices_graded_lexico)
 *   m0 [z] or [B][z]; mean0 [2] or [B][2] (central, scaled); scale0 likewise (scaled); ys [B][T][ny]
 *   out_moments [B][T][z]; out_means [B][T][2] (central, scaled; NULL in raw mode); out_scales [B][T][2] (scaled);
 *   out_nell [B]; out_first_nan [B]
 */
#define MFS_ND_TERMS 14     /* kappa terms with |kappa| <= 4 */
#define MFS_ND_TERMS_MAX 27 /* ... with |kappa| <= 6 */
#define MFS_ND_ROWS 16      /* coefficient blocks of the short layout: MFS_ND_TERMS operator terms + 2 variance rows */
#define MFS_ND_ROWS_MAX 29
#define MFS_ND_TABLE_ROWS(n_terms) ((n_terms) > MFS_ND_TERMS ? MFS_ND_ROWS_MAX : MFS_ND_ROWS)
#define MFS_ND_MAX_EXTENT 6    /* per-variable extent (degree + 1) of the coefficient blocks, short layout */
#define MFS_ND_MAX_EXTENT_HI 7 /* ... long layout (TME order 3 of a quadratic drift reaches degree 6) */
#define MFS_ND_MAX_FACTORS 2
#define MFS_ND_TRANS_OPERATOR 0
#define MFS_ND_TRANS_GAUSSIAN 1
typedef struct mfs_model_nd {
    int32_t d;             /* 2 (d = 1 problems go through the 1-D entry points, which the reference guarantees equal:
                              tests/test_filtering.py:304-329; mfs_amd.multi_dims.filtering routes them) */
    int32_t trans_kind;    /* MFS_ND_TRANS_* */
    int32_t n_terms;       /* MFS_ND_TABLE_ROWS(n_terms) blocks are passed; operator terms >= n_terms are known to be zero */
    int32_t extent;        /* D <= MFS_ND_MAX_EXTENT (MFS_ND_MAX_EXTENT_HI with the long layout) */
    int32_t n_factors;     /* 1 .. MFS_ND_MAX_FACTORS likelihood factors */
    int32_t ny;            /* measurement columns per step (1 or 2) */
    int32_t fac_kind[MFS_ND_MAX_FACTORS];      /* MFS_LIK_* */
    int32_t fac_component[MFS_ND_MAX_FACTORS]; /* state component the factor reads */
    int32_t fac_ycol[MFS_ND_MAX_FACTORS];      /* measurement column the factor reads (< ny) */
    int32_t fac_n_par[MFS_ND_MAX_FACTORS];     /* parameters used (<= MFS_MAX_LIK) */
    int32_t coef_batched;  /* 0: one table; 1: one per replicate (per-replicate drift / dispersion parameters) */
    int32_t lik_batched;   /* likewise for the likelihood parameters */
    const double* coef;    /* [rows][D][D] or [B][rows][D][D], rows = MFS_ND_TABLE_ROWS(n_terms) */
    const double* lik;     /* [n_factors][MFS_MAX_LIK] or [B][n_factors][MFS_MAX_LIK], unused entries 0 */
} mfs_model_nd;

int mfs_filter_nd(const mfs_model_nd* model, int mode, int N, int T, int B, int z, const int32_t* multi_indices,
                  const int32_t* inds, const double* m0, int m0_batched, const double* mean0, const double* scale0,
                  const double* ys, int stable, double* out_moments, double* out_means, double* out_scales,
                  double* out_nell, int32_t* out_first_nan, int device, void* stream);

/*
 * ---- N-D plan: device pointers ------------------------------------------------------------------------------------
 * The same filter for callers that keep data resident in HBM (benchmark harness, repeated calls of a jitted filter as
 * in dardel/prey_predator/mf.py:54-65): model tables and gather indices are uploaded once at create; run() only
 * enqueues the kernel on `stream`.  Device buffers have the layouts documented for mfs_filter_nd; d_out_moments,
 * d_out_means, d_out_scales, d_out_first_nan may be NULL.
 */
typedef struct mfs_plan_nd mfs_plan_nd;

int mfs_plan_nd_create(mfs_plan_nd** plan, const mfs_model_nd* model /* host pointers inside */, int mode, int N,
                       int T, int B, int z, const int32_t* multi_indices, const int32_t* inds, int stable, int device);
int mfs_plan_nd_run(mfs_plan_nd* plan, const double* d_m0, int m0_batched, const double* d_mean0,
                    const double* d_scale0, const double* d_ys, double* d_out_moments, double* d_out_means,
                    double* d_out_scales, double* d_out_nell, int32_t* d_out_first_nan, void* stream);
int mfs_plan_nd_destroy(mfs_plan_nd* plan);
/* launch geometry: threads per filter (one workgroup each), grid size, dynamic LDS bytes per workgroup */
int mfs_plan_nd_geometry(const mfs_plan_nd* plan, int* threads_per_filter, int* grid, int* lds_bytes_per_block);

/*
 * ---- multi-GPU: one process per GPU, replicates sharded, NLL all-gather over RCCL / xGMI -----------------------
 * The reference has no multi-device code (its Monte-Carlo runs are separate OS processes,
 * dardel/run_benes_bernoulli_mf.sh:26-31); replicates share nothing, so the data path needs no collective and the
 * only exchange is one ncclAllGather of the per-replicate negative log-likelihoods after the kernel
 * (SURVEY.md section 8e).  Rank 0 creates an id, the host side distributes its 128 bytes by any means
 * (mfs_amd/dist.py sends it over its TCP control plane, mfs_amd/rdzv.py), every rank calls mfs_comm_init.
 */
typedef struct mfs_rccl_id { char bytes[128]; } mfs_rccl_id; /* = ncclUniqueId */
int mfs_comm_unique_id(mfs_rccl_id* id);
int mfs_comm_init(void** comm, const mfs_rccl_id* id, int nranks, int rank, int device);
/* d_send [count], d_recv [nranks * count], DEVICE pointers, rank-major; enqueued on `stream` */
int mfs_allgather_nell(void* comm, const double* d_send, double* d_recv, uint64_t count, void* stream);
int mfs_comm_destroy(void* comm);
int mfs_memcpy_d2d(void* dst, const void* src, uint64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MFS_HIP_H */
