// filter1d_grad.hpp -- forward-mode derivative of the 1-D moment filter's negative log-likelihood with respect to model
// parameters, propagated INSIDE the time-step loop.
//
// Reference: dardel/parameter_estimation/mf.py:37-54,70-73 -- obj_func(params, ys) runs moment_filter_cms and
// jaxopt.ScipyMinimize(L-BFGS-B) differentiates through the lax.scan by JAX autodiff.  Here every quantity of the scan
// body (mfs/one_dim/filtering.py:73-86, 140-158, 217-237) is a dual number (value, P tangents) and the whole recursion
// runs on them: no reverse pass, no tape, P extra "lanes of state" per filter.
//
// Per rule, with the moments m_n as duals:
//   recurrence coefficients  the Chebyshev algorithm on the moments,
//                                sigma_{k,l} = sigma_{k-1,l+1} - alpha_{k-1} sigma_{k-1,l} - beta_{k-1} sigma_{k-2,l},
//                            whose diagonal sigma_{k,k} is the k-th pivot of the Hankel Cholesky (quadtures.py:127) and
//                            alpha_k = sigma_{k,k+1}/sigma_{k,k} - sigma_{k-1,k}/sigma_{k-1,k-1}, beta_k = sigma_{k,k}/sigma_{k-1,k-1}
//                            the entries of the Jacobi matrix K = R^-1 H R^-T (:128-131).  O(N^2) dual operations, done
//                            redundantly by every lane (no cross-lane traffic); a pivot that is not > 0 poisons, as upstream.
//   eigenvalues              lane k finds the k-th root of the monic p_N by Sturm bisection + Newton polish in plain fp64,
//                            then ONE dual evaluation of the three-term recurrence at it gives the tangent by the implicit
//                            function theorem, d lambda = -(d_theta p_N)(lambda) / p_N'(lambda)
//   weights                  w_k = 1 / sum_j p_j(lambda_k)^2 / h_j (h_j = beta_0 ... beta_j), evaluated on duals (= V[0,k]^2, :133)
// Prediction and update are the filter's own sums with dual nodes, weights and model tables (value tables + tangent tables
// d coef / d theta_p supplied by the host tracer); exp / log / tanh / sqrt carry their derivatives.
//
// Mapping: one filter per 16-lane group (N <= 15), lane = node; group sums by shuffles.  This kernel serves optimiser
// loops (a few to a few thousand filters per launch), not the throughput benchmark.
#pragma once
#include "filter1d_kernel.hpp"
#include "filter1d_fast.hpp"     // static_for

namespace mfs {

template <int P>
struct Dual {
    double v;
    double d[P];
};

template <int P> __device__ __forceinline__ Dual<P> dconst(const double v) {
    Dual<P> r; r.v = v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = 0.0;
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator+(const Dual<P>& a, const Dual<P>& b) {
    Dual<P> r; r.v = a.v + b.v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = a.d[p] + b.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator-(const Dual<P>& a, const Dual<P>& b) {
    Dual<P> r; r.v = a.v - b.v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = a.d[p] - b.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator-(const Dual<P>& a) {
    Dual<P> r; r.v = -a.v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = -a.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator*(const Dual<P>& a, const Dual<P>& b) {
    Dual<P> r; r.v = a.v * b.v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = fma(a.v, b.d[p], a.d[p] * b.v);
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator*(const double a, const Dual<P>& b) {
    Dual<P> r; r.v = a * b.v;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = a * b.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> operator+(const Dual<P>& a, const double b) { Dual<P> r = a; r.v += b; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator-(const Dual<P>& a, const double b) { Dual<P> r = a; r.v -= b; return r; }
template <int P> __device__ __forceinline__ Dual<P> operator/(const Dual<P>& a, const Dual<P>& b) {
    const double ib = 1.0 / b.v;
    Dual<P> r; r.v = a.v * ib;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = (a.d[p] - r.v * b.d[p]) * ib;
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> drecip(const Dual<P>& b) {
    const double ib = 1.0 / b.v;
    Dual<P> r; r.v = ib;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = -ib * ib * b.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> dchain(const Dual<P>& a, const double f, const double fp) {   // f(a), f'(a)
    Dual<P> r; r.v = f;
#pragma unroll
    for (int p = 0; p < P; ++p) r.d[p] = fp * a.d[p];
    return r;
}
template <int P> __device__ __forceinline__ Dual<P> dexp(const Dual<P>& a) { const double e = exp(a.v); return dchain(a, e, e); }
template <int P> __device__ __forceinline__ Dual<P> dlog(const Dual<P>& a) { return dchain(a, log(a.v), 1.0 / a.v); }
template <int P> __device__ __forceinline__ Dual<P> dsqrt(const Dual<P>& a) { const double s = sqrt(a.v); return dchain(a, s, 0.5 / s); }
template <int P> __device__ __forceinline__ Dual<P> dtanh(const Dual<P>& a) { const double t = tanh(a.v); return dchain(a, t, 1.0 - t * t); }

template <int P, int G>
__device__ __forceinline__ Dual<P> dgroup_sum(Dual<P> a) {
    a.v = group_sum<G>(a.v);
#pragma unroll
    for (int p = 0; p < P; ++p) a.d[p] = group_sum<G>(a.d[p]);
    return a;
}

struct Filter1dGradArgs {
    Filter1dArgs f;          // the plain filter's arguments (model tables, inputs; out_mom / out_mean unused)
    int n_par;               // P
    const double* dcoef;     // [P][n_rows][degree + 1] (or [B][P][...] when coef_batched): d coef / d theta_p
    const double* dlik;      // [P][n_lik] (or [B][P][n_lik]): d (likelihood parameters) / d theta_p
    double* out_grad;        // [B][P]: d nell / d theta_p
};

// polynomial with dual coefficients at a dual argument
template <int P>
__device__ __forceinline__ Dual<P> dhorner(const double* __restrict__ c, const double* __restrict__ dc, const int stride_p,
                                           const int degree, const Dual<P>& u) {
    Dual<P> acc;
    acc.v = c[degree];
#pragma unroll
    for (int p = 0; p < P; ++p) acc.d[p] = dc[p * stride_p + degree];
    for (int j = degree - 1; j >= 0; --j) {
        Dual<P> cj;
        cj.v = c[j];
#pragma unroll
        for (int p = 0; p < P; ++p) cj.d[p] = dc[p * stride_p + j];
        acc = acc * u + cj;
    }
    return acc;
}

template <int P>
__device__ __forceinline__ Dual<P> dlikelihood(const int kind, const double* __restrict__ lp, const double* __restrict__ dlp,
                                               const int n_lik, const double y, const Dual<P>& x) {
    auto par = [&](int i) {
        Dual<P> r; r.v = lp[i];
#pragma unroll
        for (int p = 0; p < P; ++p) r.d[p] = dlp[p * n_lik + i];
        return r;
    };
    if (kind == MFS_LIK_BERNOULLI_LOGISTIC) {
        Dual<P> z = par(0);
        Dual<P> xp = x;
        for (int i = 1; i < 4; ++i) { if (i < n_lik) z = z + par(i) * xp; xp = xp * x; }
        const Dual<P> pr = drecip(dexp(-z) + 1.0);
        return (y > 0.5) ? pr : (-pr + 1.0);
    } else if (kind == MFS_LIK_POISSON_SOFTPLUS) {
        const Dual<P> rate = dlog(dexp(par(0) * x) + 1.0);
        return dexp(y * dlog(rate) - rate - log_factorial(y));
    } else {
        const Dual<P> r = -(par(0) * x + par(1)) + y;
        const Dual<P> var = par(2);
        return dexp(-0.5 * (r * r / var)) * drecip(dsqrt(6.283185307179586476925 * var));
    }
}

// Gauss rule on duals.  The Chebyshev algorithm on the 2N dual moments runs LANE-PARALLEL in LDS (round 3): row k of the
// sigma table is sigma_{k,l} = sigma_{k-1,l+1} - alpha_{k-1} sigma_{k-1,l} - beta_{k-1} sigma_{k-2,l}, l = k .. 2N-1-k, one or two
// entries per lane, written over row k - 2 (which only the writing lane has read), one wavefront fence per row.  Round 2 kept
// two full rows of duals redundantly in every lane's registers: 2 x 2N x (1 + P) doubles -- 480 registers at N = 15, P = 3,
// i.e. 864 spilled at N = 15, P = 2.  alpha_k, beta_k (group-uniform duals) go to LDS as they are found; their value parts are
// read back into registers for the eigenvalue search, their tangents stream through the two dual recurrences.  Lane l < N
// then owns root l.
//   qs: per-group scratch, [2][1 + P][2N] rows + [2][1 + P][N] coefficients.
#ifdef MFS_GRAD_WARM_DEBUG
__device__ unsigned long long g_grad_warm_dbg[4];
#endif
template <int N, int G, int P>
__device__ __forceinline__ bool dual_quadrature(const double* __restrict__ mom /* [1 + P][2N] */, double* __restrict__ qs, const int l,
                                                const Dual<P>& mean, const Dual<P>& scale, Dual<P>& x_out, Dual<P>& w_out,
                                                double& warm) {
    constexpr int M2 = 2 * N, DW = (1 + P) * M2;
    double* X = qs;                      // row k - 1
    double* Y = qs + DW;                 // row k - 2
    double* AL = qs + 2 * DW;            // [1 + P][N]
    double* BE = AL + (1 + P) * N;
    auto dload = [&](const double* base, const int stride, const int n) {
        Dual<P> r; r.v = base[n];
#pragma unroll
        for (int p = 0; p < P; ++p) r.d[p] = base[(1 + p) * stride + n];
        return r;
    };
    auto dstore = [&](double* base, const int stride, const int n, const Dual<P>& c) {
        base[n] = c.v;
#pragma unroll
        for (int p = 0; p < P; ++p) base[(1 + p) * stride + n] = c.d[p];
    };
    bool poisoned = false;
    constexpr bool kLds = (N > 8);      // registers hold the work arrays up to N = 8 (faster there: no LDS round trips per row; N = 10, P = 2 already spills 52)
    Dual<P> ra[kLds ? 1 : N], rb[kLds ? 1 : N];      // alpha, beta as duals in registers (register form)
    double av[N], bv[N];                              // their value parts (both forms)
    if constexpr (!kLds) {
        Dual<P> (&alpha)[N] = ra;
        Dual<P> (&beta)[N] = rb;
        {
            Dual<P> prev[M2], cur[M2], piv_prev, sub_prev_over_piv;
#pragma unroll
            for (int n = 0; n < M2; ++n) {
                cur[n].v = mom[n];
#pragma unroll
                for (int p = 0; p < P; ++p) cur[n].d[p] = mom[(1 + p) * M2 + n];
                prev[n] = dconst<P>(0.0);
            }
            beta[0] = cur[0];
            poisoned |= !(cur[0].v > 0.0);
            alpha[0] = cur[1] / cur[0];
            piv_prev = cur[0];
            sub_prev_over_piv = alpha[0];
            // sigma_{k, l}, l = k .. 2N - 1 - k, from rows k - 1 (cur) and k - 2 (prev), overwriting in place
#pragma unroll
            for (int k = 1; k < N; ++k) {
#pragma unroll
                for (int ll = k; ll < M2 - k; ++ll) {
                    const Dual<P> s = cur[ll + 1] - alpha[k - 1] * cur[ll] - ((k >= 2) ? beta[k - 1] * prev[ll] : dconst<P>(0.0));
                    prev[ll] = cur[ll];      // row k - 1 entry l, needed as "k - 2" in the next round
                    cur[ll] = s;
                }
                // (entries below k of cur / prev are stale and never read again)
                const Dual<P> piv = cur[k];
                poisoned |= !(piv.v > 0.0);
                beta[k] = piv / piv_prev;
                const Dual<P> ratio = cur[k + 1] / piv;
                alpha[k] = ratio - sub_prev_over_piv;
                piv_prev = piv;
                sub_prev_over_piv = ratio;
            }
        }

#pragma unroll
        for (int j = 0; j < N; ++j) { av[j] = ra[j].v; bv[j] = rb[j].v; }
    } else {
        for (int e = l; e < DW; e += G) { X[e] = mom[e]; Y[e] = 0.0; }
        wave_sync();
        {
            const Dual<P> c0 = dload(X, M2, 0), c1 = dload(X, M2, 1);
            poisoned |= !(c0.v > 0.0);
            Dual<P> a_prev = c1 / c0, b_prev = c0, piv_prev = c0, sub_prev_over_piv = a_prev;
            if (l == 0) { dstore(AL, N, 0, a_prev); dstore(BE, N, 0, c0); }      // beta[0] = m_0
            for (int k = 1; k < N; ++k) {
                for (int ll = k + l; ll < M2 - k; ll += G) {
                    Dual<P> sg = dload(X, M2, ll + 1) - a_prev * dload(X, M2, ll);
                    if (k >= 2) sg = sg - b_prev * dload(Y, M2, ll);
                    dstore(Y, M2, ll, sg);
                }
                wave_sync();
                double* t = X; X = Y; Y = t;
                const Dual<P> piv = dload(X, M2, k), nxt = dload(X, M2, k + 1);
                poisoned |= !(piv.v > 0.0);
                b_prev = piv / piv_prev;
                const Dual<P> ratio = nxt / piv;
                a_prev = ratio - sub_prev_over_piv;
                if (l == 0) { dstore(AL, N, k, a_prev); dstore(BE, N, k, b_prev); }
                piv_prev = piv;
                sub_prev_over_piv = ratio;
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < N; ++j) { av[j] = AL[j]; bv[j] = BE[j]; }
    }
    struct { double v; } alpha[N], beta[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { alpha[j].v = av[j]; beta[j].v = bv[j]; }
    auto dalpha = [&](const int j) { if constexpr (kLds) return dload(AL, N, j); else return ra[j]; };
    auto dbeta = [&](const int j) { if constexpr (kLds) return dload(BE, N, j); else return rb[j]; };
    // ---- root l by Sturm bisection (plain fp64), Gershgorin interval of the symmetric tridiagonal
    double lo = 1.79e308, hi = -1.79e308;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double r = ((j > 0) ? sqrt(fabs(beta[j].v)) : 0.0) + ((j + 1 < N) ? sqrt(fabs(beta[j + 1].v)) : 0.0);
        lo = fmin(lo, alpha[j].v - r);
        hi = fmax(hi, alpha[j].v + r);
    }
    const int k = (l < N) ? l : N - 1;
    // number of eigenvalues < x: sign changes of the Sturm sequence P_j = (alpha_j - x) P_{j-1} - beta_j P_{j-2}, P_0 = 1 -- the
    // pivots q_j = P_j / P_{j-1} of the LDL^T of T - x I without forming them (the quotient form is one fp64 DIVISION per j and
    // bisection step: 64 steps x N divisions were 15 k of the 19 k instructions of a filter step at N = 7).  A zero P_{j-1}
    // counts as positive, which gives the same total as the pivot form's 1e-300 substitute.  Magnitudes: |P_j| <= (spectral
    // width)^j, N <= 16 -- far inside the fp64 range for the central / scaled moments this kernel is used on; the sequence is
    // rescaled by 2^-512 should it ever pass 2^512.
    auto count_below = [&](const double x) {
        double p0 = 1.0, p1 = alpha[0].v - x;
        int cnt = (p1 < 0.0);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            double pn = fma(alpha[j].v - x, p1, -beta[j].v * p0);
            cnt += ((pn < 0.0) != (p1 < 0.0));
            if constexpr (N > 8) if (j == 8 && fabs(pn) > 1.3407807929942597e154) { pn *= 7.458340731200207e-155; p1 *= 7.458340731200207e-155; }
            p0 = p1; p1 = pn;
        }
        return cnt;
    };
    // 44 halvings of the Gershgorin interval isolate root k to 6e-14 of the spectral width; the Newton steps below finish it
    // (they were already there after 64 halvings, which the fp64 mantissa cannot even resolve)
    // Warm start (MFS_GRAD_WARM_ROOTS): root k of the same rule one time step ago, three Newton steps, and TWO Sturm counts that
    // prove the point sits within 1e-4 of the spectral width of root k and of no other root; the halvings run only where that
    // proof fails in some lane of the wave (the first step, a root that moved too far, roots within 1e-4 of each other).
    // Measured (Benes, P = 3): 95 % of the lane-rules pass, 53 % of the wave-rules skip the halvings.  The three Newton steps
    // that finish a root are the same either way.
#ifndef MFS_GRAD_WARM_ROOTS
#define MFS_GRAD_WARM_ROOTS 1
#endif
#ifndef MFS_GRAD_HALVINGS
#define MFS_GRAD_HALVINGS 32      // 2e-10 of the spectral width: roots closer than that are one root to the rule; three Newton steps finish
#endif
#ifndef MFS_GRAD_WARM_NEWTON
#define MFS_GRAD_WARM_NEWTON 3
#endif
#ifndef MFS_GRAD_WARM_DEL
#define MFS_GRAD_WARM_DEL 1e-4
#endif
    double a = lo, b = hi;
    bool bracketed = false;
    if (MFS_GRAD_WARM_ROOTS && warm == warm) {
        double x = warm;
#pragma unroll
        for (int it = 0; it < MFS_GRAD_WARM_NEWTON; ++it) {
            double p0 = 1.0, p1 = x - alpha[0].v, d0 = 0.0, d1 = 1.0;
#pragma unroll
            for (int j = 1; j < N; ++j) {
                const double pn = (x - alpha[j].v) * p1 - beta[j].v * p0;
                const double dn = p1 + (x - alpha[j].v) * d1 - beta[j].v * d0;
                p0 = p1; p1 = pn; d0 = d1; d1 = dn;
            }
            const double step = p1 / d1;
            if (finite(step)) x -= step;
        }
        const double del = MFS_GRAD_WARM_DEL * (hi - lo);
        if (x - del > lo && x + del < hi && count_below(x - del) == k && count_below(x + del) == k + 1) {
            a = x - del; b = x + del; bracketed = true;
        }
    }
#ifdef MFS_GRAD_WARM_DEBUG
    if (l < N) { atomicAdd(&g_grad_warm_dbg[0], 1ull); if (bracketed) atomicAdd(&g_grad_warm_dbg[1], 1ull); if (__builtin_amdgcn_ballot_w64(!bracketed) == 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_grad_warm_dbg[2], 1ull); if ((threadIdx.x & 63) == 0) atomicAdd(&g_grad_warm_dbg[3], 1ull); }
#endif
    if (!bracketed) {
        for (int it = 0; it < MFS_GRAD_HALVINGS; ++it) {
            const double mid = 0.5 * (a + b);
            if (count_below(mid) > k) b = mid; else a = mid;
        }
    }
    double lam = 0.5 * (a + b);
    for (int it = 0; it < 3; ++it) {           // Newton polish on the monic p_N
        double p0 = 1.0, p1 = lam - alpha[0].v, d0 = 0.0, d1 = 1.0;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            const double pn = (lam - alpha[j].v) * p1 - beta[j].v * p0;
            const double dn = p1 + (lam - alpha[j].v) * d1 - beta[j].v * d0;
            p0 = p1; p1 = pn; d0 = d1; d1 = dn;
        }
        const double step = p1 / d1;
        if (finite(step) && lam - step > a - (b - a) && lam - step < b + (b - a)) lam -= step;
    }
    warm = lam;
    // ---- tangent of the root: one dual evaluation of the recurrence at (lam, 0)
    Dual<P> L0 = dconst<P>(lam);
    {
        Dual<P> p0 = dconst<P>(1.0), p1 = L0 - dalpha(0);
        double d0 = 0.0, d1 = 1.0;
#pragma unroll
        for (int j = 1; j < N; ++j) {
            const Dual<P> pn = (L0 - dalpha(j)) * p1 - dbeta(j) * p0;
            const double dn = p1.v + (lam - alpha[j].v) * d1 - beta[j].v * d0;
            p0 = p1; p1 = pn; d0 = d1; d1 = dn;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) L0.d[p] = -p1.d[p] / d1;
    }
    // ---- weight on duals: 1 / sum_j p_j(lam)^2 / h_j
    Dual<P> w;
    {
        Dual<P> p0 = dconst<P>(1.0), p1 = L0 - dalpha(0);
        const Dual<P> b0 = dbeta(0);
        Dual<P> h = b0;
        Dual<P> acc = drecip(h);
#pragma unroll
        for (int j = 1; j < N; ++j) {
            const Dual<P> bj = dbeta(j);
            h = h * bj;
            acc = acc + p1 * p1 / h;
            const Dual<P> pn = (L0 - dalpha(j)) * p1 - bj * p0;
            p0 = p1; p1 = pn;
        }
        w = drecip(acc * b0);     // V[0, :]**2 of the normalised eigenvectors (quadtures.py:133): sums to 1 whatever m_0 is
    }
    const double qnan = __builtin_nan("");
    x_out = scale * L0 + mean;
    w_out = (l < N) ? w : dconst<P>(0.0);
    if (poisoned) { x_out = dconst<P>(qnan); w_out = dconst<P>(qnan); }
    return poisoned;
}

// One filter per G-lane group.  LDS per filter: dual moments [1 + P][2N], contribution table [G][1 + P][2N], the quadrature's
// scratch (two rows of the sigma table, alpha / beta).
// (A/B switch: two waves per SIMD for N <= 8, P <= 2 is a 256-register build that spills 135 registers at N = 7, P = 2 and
//  measured slower -- 154 against 139 ms per host call of 16 384 filters, round 3.)
#ifndef MFS_GRAD_OCC2
#define MFS_GRAD_OCC2 0
#endif
template <int N, int P> constexpr int grad_occ() { return (MFS_GRAD_OCC2 && N <= 8 && P <= 2) ? 2 : 1; }
template <int N, int G, int P>
__global__ __launch_bounds__(64, (grad_occ<N, P>())) void filter1d_grad_kernel(const Filter1dGradArgs ga) {
    const Filter1dArgs& a = ga.f;
    constexpr int M2 = 2 * N, FPW = 64 / G, DW = (1 + P) * M2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, l = lane - grp * G;
    const int b = blockIdx.x * FPW + grp;
    if (b >= a.B) return;
    constexpr int kQs = (N > 8) ? 2 * DW + 2 * (1 + P) * N : 0;     // (the LDS form of the Chebyshev algorithm only)
    double* S = smem + (size_t)grp * (DW + G * DW + kCoefDoubles * (1 + P) + (MFS_MAX_LIK) * (1 + P) + kQs);
    double* mom = S;                              // [1 + P][2N]
    double* TAB = S + DW;                         // [G][1 + P][2N]
    double* coef = TAB + G * DW;                  // [n_rows][J + 1]
    double* dcoef = coef + kCoefDoubles;          // [P][n_rows][J + 1]
    double* lp = dcoef + kCoefDoubles * P;        // [n_lik]
    double* dlp = lp + MFS_MAX_LIK;               // [P][n_lik]
    double* qs = dlp + MFS_MAX_LIK * P;           // quadrature scratch
    const int J1 = a.degree + 1, ncoef = a.n_rows * J1;
    {
        const double* src = a.coef + (a.coef_batched ? (size_t)b * ncoef : 0);
        for (int e = l; e < ncoef; e += G) coef[e] = src[e];
        const double* dsrc = ga.dcoef + (a.coef_batched ? (size_t)b * P * ncoef : 0);
        for (int e = l; e < P * ncoef; e += G) dcoef[e] = dsrc[e];
        const double* ls = a.lik + (a.lik_batched ? (size_t)b * a.n_lik : 0);
        const double* dls = ga.dlik + (a.lik_batched ? (size_t)b * P * a.n_lik : 0);
        for (int e = l; e < a.n_lik; e += G) lp[e] = ls[e];
        for (int e = l; e < P * a.n_lik; e += G) dlp[e] = dls[e];
        const double* m0 = a.m0 + (a.m0_batched ? (size_t)b * M2 : 0);
        for (int e = l; e < DW; e += G) mom[e] = (e < M2) ? m0[e] : 0.0;     // initial moments do not depend on theta
    }
    Dual<P> mean = dconst<P>(0.0), scale = dconst<P>(1.0), nell = dconst<P>(0.0);
    if (a.mode != MFS_MODE_RAW) mean.v = a.mean0[a.m0_batched ? b : 0];
    if (a.mode == MFS_MODE_SCALED) scale.v = a.scale0[a.m0_batched ? b : 0];
    wave_sync();
    const double* yrow = a.ys + (size_t)b * a.T;
    bool dead = false;
    int first_nan = -1;
    double warm_predict = __builtin_nan(""), warm_update = __builtin_nan("");     // this lane's root of the two rules one step ago (in the rule's own units)
    double* myrow = TAB + l * DW;

    auto store_contrib = [&](const int n, const Dual<P>& c) {
        myrow[n] = c.v;
#pragma unroll
        for (int p = 0; p < P; ++p) myrow[(1 + p) * M2 + n] = c.d[p];
    };
    auto reduce_moments = [&](const Dual<P>& py) {
        wave_sync();
        const Dual<P> ipy = drecip(py);
        bool bad = false;
        for (int n = l; n < M2; n += G) {
            Dual<P> acc = dconst<P>(0.0);
            for (int i = 0; i < N; ++i) {
                acc.v += TAB[i * DW + n];
#pragma unroll
                for (int p = 0; p < P; ++p) acc.d[p] += TAB[i * DW + (1 + p) * M2 + n];
            }
            acc = acc * ipy;
            mom[n] = acc.v;
#pragma unroll
            for (int p = 0; p < P; ++p) mom[(1 + p) * M2 + n] = acc.d[p];
            bad |= !finite(acc.v);
        }
        wave_sync();
        return bad;
    };

    for (int t = 0; t < a.T && !dead; ++t) {
        const double y = yrow[t];
        int bad = 0;
        // ---- prediction
        {
            Dual<P> x, w;
            bad |= dual_quadrature<N, G, P>(mom, qs, l, mean, scale, x, w, warm_predict);
            const Dual<P> u = (a.umap == MFS_U_TANH) ? dtanh(x) : x;
            Dual<P> c = dconst<P>(0.0), inv_sc = dconst<P>(1.0);
            if (a.trans_kind == MFS_TRANS_GAUSSIAN) {
                const Dual<P> mu = a.mean_x_coef * x + dhorner<P>(coef, dcoef, ncoef, a.degree, u);
                const Dual<P> var = dhorner<P>(coef + J1, dcoef + J1, ncoef, a.degree, u);
                if (a.mode != MFS_MODE_RAW) { mean = dgroup_sum<P, G>(w * mu); c = mean; }
                if (a.mode == MFS_MODE_SCALED) { scale = dsqrt(dgroup_sum<P, G>(w * var)); inv_sc = drecip(scale); }
                // E_0 = 1, E_1 = m, E_n = m E_{n-1} + (n - 1) v E_{n-2}, m = mu - c
                const Dual<P> m = mu - c;
                Dual<P> e2 = dconst<P>(1.0), e1 = m, scn = w;
                store_contrib(0, scn);
                scn = scn * inv_sc;
                store_contrib(1, scn * m);
                for (int n = 2; n < M2; ++n) {
                    const Dual<P> e = m * e1 + (double)(n - 1) * (var * e2);
                    scn = scn * inv_sc;
                    store_contrib(n, scn * e);
                    e2 = e1; e1 = e;
                }
            } else {
                // E[(X' - c)^n | x] = sum_k Q_k(u) n!/(n-k)! (x - c)^(n-k), Q_0 = 1
                // (every index into Q and D below is a compile-time constant: with the run-time bound `k <= a.n_terms` on a loop
                //  over them the two arrays lived in scratch memory -- 144 (1 + P) bytes per lane, a load and a `s_waitcnt vmcnt(0)`
                //  per access)
                Dual<P> Q[MFS_MAX_TERMS + 1];
                Q[0] = dconst<P>(1.0);
                static_for<1, MFS_MAX_TERMS + 1>([&](auto Kc) {
                    constexpr int k = Kc;
                    if (k <= a.n_terms) Q[k] = dhorner<P>(coef + (k - 1) * J1, dcoef + (k - 1) * J1, ncoef, a.degree, u);
                    else Q[k] = dconst<P>(0.0);
                });
                if (a.mode != MFS_MODE_RAW) { mean = dgroup_sum<P, G>(w * (x + Q[1])); c = mean; }
                if (a.mode == MFS_MODE_SCALED) {
                    const Dual<P> var = dhorner<P>(coef + a.n_terms * J1, dcoef + a.n_terms * J1, ncoef, a.degree, u);
                    scale = dsqrt(dgroup_sum<P, G>(w * var));
                    inv_sc = drecip(scale);
                }
                const Dual<P> dx = x - c;
                Dual<P> D[MFS_MAX_TERMS + 1];
                D[0] = dconst<P>(1.0);
                static_for<1, MFS_MAX_TERMS + 1>([&](auto Kc) { D[Kc] = dconst<P>(0.0); });
                Dual<P> scn = w;
                for (int n = 0; n < M2; ++n) {
                    Dual<P> val = dconst<P>(0.0);
                    static_for<0, MFS_MAX_TERMS + 1>([&](auto Kc) {
                        constexpr int k = MFS_MAX_TERMS - Kc;       // from k = n_terms down to 0, as the plain filter sums them
                        if (k <= a.n_terms) val = val + Q[k] * D[k];
                    });
                    store_contrib(n, scn * val);
                    scn = scn * inv_sc;
                    static_for<0, MFS_MAX_TERMS>([&](auto Kc) {
                        constexpr int k = MFS_MAX_TERMS - Kc;
                        if (k <= a.n_terms) D[k] = dx * D[k] + (double)k * D[k - 1];
                    });
                    D[0] = D[0] * dx;
                }
            }
            bad |= reduce_moments(dconst<P>(1.0));
        }
        // ---- update
        {
            Dual<P> x, w;
            bad |= dual_quadrature<N, G, P>(mom, qs, l, mean, scale, x, w, warm_update);
            const Dual<P> wl = w * dlikelihood<P>(a.lik_kind, lp, dlp, a.n_lik, y, x);
            const Dual<P> py = dgroup_sum<P, G>(wl);
            Dual<P> c = dconst<P>(0.0), inv_sc = dconst<P>(1.0);
            if (a.mode != MFS_MODE_RAW) { mean = dgroup_sum<P, G>(wl * x) / py; c = mean; }
            const Dual<P> dxc = x - c;
            if (a.mode == MFS_MODE_SCALED) { scale = dsqrt(dgroup_sum<P, G>(wl * dxc * dxc) / py); inv_sc = drecip(scale); }
            const Dual<P> dx = dxc * inv_sc;
            Dual<P> pw = wl;
            for (int n = 0; n < M2; ++n) { store_contrib(n, pw); pw = pw * dx; }
            bad |= reduce_moments(py);
            nell = nell - dlog(py);
        }
        bad |= (int)(!finite(nell.v) || !finite(mean.v) || !finite(scale.v));
        bad = group_or<G>(bad);
        if (bad) { dead = true; first_nan = t; }
    }
    if (l == 0) {
        const double qnan = __builtin_nan("");
        a.out_nell[b] = dead ? qnan : nell.v;
        for (int p = 0; p < P; ++p) ga.out_grad[(size_t)b * P + p] = dead ? qnan : nell.d[p];
        if (a.out_first_nan) a.out_first_nan[b] = first_nan;
    }
}

using Filter1dGradLaunch = hipError_t (*)(const Filter1dGradArgs&, int n_filters, hipStream_t);

}  // namespace mfs
