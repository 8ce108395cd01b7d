// filter1d_kernel.hpp -- hand-written HIP for gfx950 (MI355X): the 1-D moment-filter time-step loop.
//
// One filter (replicate) is owned by a group of G lanes of a 64-lane wavefront (G = 16, 32 or 64; 64/G filters per
// wave); the whole sequential k-loop runs inside the kernel with the filter state on chip.  Per time step
// (reference mfs/one_dim/filtering.py:73-86, 140-158, 217-237):
//
//   quadrature   Hankel tiles G[i][j] = m[i+j], H[i][j] = m[i+j+1] gathered from the LDS-staged moment vector,
//                Cholesky G = R R^T, K = R^-1 H R^-T by two triangular solves, symmetrise, cyclic Jacobi eigensolve
//                keeping only the eigenvalues and the first row of the eigenvector matrix
//                (mfs/one_dim/quadtures.py:122-133: weights V[0,:]^2, nodes scale * lambda + mean)
//   predict      m_n <- sum_i w_i E[(X' - c)^n | x_i]   (transition-moment map from coefficient tables)
//   quadrature   again on the predicted moments
//   update       p_y = sum_i w_i l(y, x_i);  m_n <- sum_i w_i (x_i - c)^n l(y, x_i) / p_y;  nell -= log p_y
//
// No MFMA: there is no dense contraction here (N <= 32, strictly sequential dependency chain); the bound is fp64
// VALU issue + LDS latency.  HBM traffic is one y in and 2N(+2) doubles out per step.
//
// Everything a group shares goes through its private LDS region; groups never talk to each other, so the only
// synchronisation is the wave-level ordering of LDS instructions (wave_sync() below), never s_barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mfs_hip.h"

namespace mfs {

struct Filter1dArgs {
    int mode, T, B, stable;
    int extra;   // 1: the moment vectors carry 2N + 1 entries (dense path only; the order-2N moment is an output, never an input)
    int t_begin, t_end;
    // model
    int trans_kind, umap, n_terms, degree, n_rows, coef_batched, lik_kind, n_lik, lik_batched;
    double mean_x_coef;
    const double* coef;
    const double* lik;
    // inputs
    const double* m0;
    int m0_batched;
    const double* mean0;
    const double* scale0;
    const double* ys;
    // carry between chunks: [B][2N], [B], [B], [B], [B]
    double* c_mom;
    double* c_mean;
    double* c_scale;
    double* c_nell;
    int32_t* c_first_nan;
    double* c_lam;  // [B][2][G] nodes (in rule units) and weights of the posterior atoms, i.e. the next predict-half rule (fast
                    // path, chunked runs only)
    int recompute_rule;   // 1: the predict-half rule is recomputed from the posterior moments (MFS_PREDICT_RULE=recompute)
    // outputs (any may be null except out_nell)
    double* out_mom;
    double* out_mean;
    double* out_scale;
    double* out_nell;
    int32_t* out_first_nan;
};

struct Quad1dArgs {
    int B, stable;
    const double* ms;
    const double* mean;
    const double* scale;
    double* out_w;
    double* out_x;
};

constexpr int kMaxSweeps = 40;
constexpr int kCoefDoubles = (MFS_MAX_TERMS + 1) * (MFS_MAX_DEGREE + 1);

template <int N>
struct Tile {
    static constexpr int NP = N + (N & 1);         // padded to even for the Jacobi tournament
    static constexpr int LD = NP + 1;              // odd leading dimension: conflict-free column access
    static constexpr int M2 = 2 * N;
    // per-filter LDS layout (in doubles)
    static constexpr int oMom = 0;                 // [2N] moment vector (Hankel source)
    static constexpr int oA = M2;                  // [NP][LD] G -> R (lower Cholesky factor)
    static constexpr int oK = oA + NP * LD;        // [NP][LD] H -> X -> K -> eigen iteration
    static constexpr int oCs = oK + NP * LD;       // [NP/2][3] rotation (c, s, t) per pair
    static constexpr int oV0 = oCs + 3 * (NP / 2); // [NP] first row of the eigenvector matrix
    static constexpr int oX = oV0 + NP;            // [NP] nodes
    static constexpr int oW = oX + NP;             // [NP] weights
    static constexpr int oCoef = oW + NP;          // model table
    static constexpr int oLik = oCoef + kCoefDoubles;
    static constexpr int kDoubles = (oLik + MFS_MAX_LIK + 1) & ~1;
    static constexpr int oTab = oA;                // [N][2N] per-node moment contributions, aliases A and K tiles
    static_assert(N * M2 <= 2 * NP * LD, "contribution table must fit in the two tiles");
};

__device__ __forceinline__ void wave_sync() {
    // Orders this wave's LDS traffic for cross-lane hand-offs inside one wavefront: LDS executes a wave's
    // instructions in order, so all that is needed is to stop the compiler from moving accesses across this point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

template <int G>
__device__ __forceinline__ int group_or(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, G);
    return v;
}

__device__ __forceinline__ bool finite(double v) { return fabs(v) <= 1.79769313486231570e308; }

// ---------------------------------------------------------------------------------------------------------------
// Gauss quadrature from 2N moments: fills S[oX] (nodes) and S[oW] (weights).  l = lane within the group.
// ---------------------------------------------------------------------------------------------------------------
template <int N, int G>
__device__ __forceinline__ void quadrature(double* __restrict__ S, const int l, const double mean, const double scale,
                           const int stable) {
    using L = Tile<N>;
    constexpr int NP = L::NP, LD = L::LD;
    double* mom = S + L::oMom;
    double* A = S + L::oA;
    double* K = S + L::oK;

    // -- Hankel gather (mfs/one_dim/quadtures.py:124-125); the pad row/column of K is zero
    for (int e = l; e < NP * NP; e += G) {
        const int i = e / NP, j = e - i * NP;
        const bool in = (i < N) && (j < N);
        A[i * LD + j] = in ? mom[i + j] : 0.0;
        K[i * LD + j] = in ? mom[i + j + 1] : 0.0;
    }
    wave_sync();

    bool poisoned = false;  // group-uniform: every lane sees the same pivots / norms
    if (!stable) {
        // -- lower Cholesky, left-looking by column (quadtures.py:127).  LAPACK potrf stops at the first pivot that
        //    is not > 0 and XLA then returns an all-NaN factor: track that and poison the whole rule below.
        for (int j = 0; j < N; ++j) {
            for (int i = j + l; i < N; i += G) {
                double s = A[i * LD + j];
                for (int k = 0; k < j; ++k) s -= A[i * LD + k] * A[j * LD + k];
                A[i * LD + j] = s;
            }
            wave_sync();
            const double d = A[j * LD + j];
            poisoned |= !(d > 0.0);
            const double r = sqrt(d);
            const double inv = 1.0 / r;
            wave_sync();
            for (int i = j + l; i < N; i += G) A[i * LD + j] = (i == j) ? r : A[i * LD + j] * inv;
            wave_sync();
        }
    } else {
        // -- LDL^T completion (mfs/utils.py:495-538): R = L diag(d < 0 ? eps : sqrt(d)), eps = 1e-8 ||G||_F
        double fro = 0.0;
        for (int e = l; e < N * N; e += G) {
            const int i = e / N, j = e - i * N;
            const double v = A[i * LD + j];
            fro += v * v;
        }
        fro = group_sum<G>(fro);
        const double eps = 1e-8 * sqrt(fro);
        // column j: d_j = a_jj - sum_k L_jk^2 d_k ; L_ij = (a_ij - sum_k L_ik L_jk d_k) / d_j.
        // d_k is kept on the diagonal of A until the final scaling pass; L's unit diagonal is implicit.
        for (int j = 0; j < N; ++j) {
            for (int i = j + l; i < N; i += G) {
                double s = A[i * LD + j];
                for (int k = 0; k < j; ++k) s -= A[i * LD + k] * (A[j * LD + k] * A[k * LD + k]);
                A[i * LD + j] = s;
            }
            wave_sync();
            const double dj = A[j * LD + j];
            wave_sync();
            for (int i = j + 1 + l; i < N; i += G) A[i * LD + j] = A[i * LD + j] / dj;
            wave_sync();
        }
        // R[i][j] = L[i][j] * f_j, f_j = d_j < 0 ? eps : sqrt(d_j)
        for (int j = 0; j < N; ++j) {
            const double dj = A[j * LD + j];
            const double fj = (dj < 0.0) ? eps : sqrt(dj);
            wave_sync();
            for (int i = j + l; i < N; i += G) A[i * LD + j] = (i == j) ? fj : A[i * LD + j] * fj;
            wave_sync();
        }
    }

    // -- X = R^-1 H: forward substitution, one column per lane (quadtures.py:128 inner solve)
    for (int c = l; c < N; c += G) {
        for (int i = 0; i < N; ++i) {
            double s = K[i * LD + c];
            for (int k = 0; k < i; ++k) s -= A[i * LD + k] * K[k * LD + c];
            K[i * LD + c] = s / A[i * LD + i];
        }
    }
    wave_sync();
    // -- K = X R^-T: one row per lane (quadtures.py:128-129 outer solve, left_side=False, transpose_a=True)
    for (int i = l; i < N; i += G) {
        for (int j = 0; j < N; ++j) {
            double s = K[i * LD + j];
            for (int k = 0; k < j; ++k) s -= K[i * LD + k] * A[j * LD + k];
            K[i * LD + j] = s / A[j * LD + j];
        }
    }
    wave_sync();
    // -- symmetrise (jax.lax.linalg.eigh symmetrize_input=True) and set up V's first row
    for (int e = l; e < N * N; e += G) {
        const int i = e / N, j = e - i * N;
        if (i > j) {
            const double s = 0.5 * (K[i * LD + j] + K[j * LD + i]);
            K[i * LD + j] = s;
            K[j * LD + i] = s;
        }
    }
    double* cs = S + L::oCs;
    double* v0 = S + L::oV0;
    for (int i = l; i < NP; i += G) v0[i] = (i == 0) ? 1.0 : 0.0;
    wave_sync();

    // -- cyclic Jacobi, round-robin tournament: NP-1 rounds of NP/2 disjoint rotations per sweep.
    //    Pair P of round r: P = 0 -> (NP-1, r); P = k -> ((r+k) mod (NP-1), (r-k) mod (NP-1)).
    constexpr int HP = NP / 2;
    double prev_off = 1.79e308;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int e = l; e < N * N; e += G) {
            const int i = e / N, j = e - i * N;
            const double v = K[i * LD + j];
            if (i == j) dia += v * v; else off += v * v;
        }
        off = group_sum<G>(off);
        dia = group_sum<G>(dia);
        if (!finite(off + dia)) { poisoned = true; break; }  // NaN / inf reached K: the rule is poisoned
        if (!(off > 1e-31 * dia)) break;  // converged
        if (off < 1e-26 * dia && off > 0.25 * prev_off) break;  // at the rounding floor: further sweeps only churn
        prev_off = off;

        for (int r = 0; r < NP - 1; ++r) {
            for (int P = l; P < HP; P += G) {
                int p, q;
                if (P == 0) { p = NP - 1; q = r; }
                else { p = r + P; if (p >= NP - 1) p -= NP - 1; q = r - P; if (q < 0) q += NP - 1; }
                const double app = K[p * LD + p], aqq = K[q * LD + q], apq = K[p * LD + q];
                double c = 1.0, s = 0.0, t = 0.0;
                if (apq != 0.0) {
                    const double theta = (aqq - app) / (2.0 * apq);
                    t = copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    c = 1.0 / sqrt(t * t + 1.0);
                    s = t * c;
                }
                cs[3 * P] = c; cs[3 * P + 1] = s; cs[3 * P + 2] = t;
                const double vp = v0[p], vq = v0[q];
                v0[p] = c * vp - s * vq;
                v0[q] = s * vp + c * vq;
            }
            wave_sync();
            for (int blk = l; blk < HP * HP; blk += G) {
                const int P = blk / HP, Q = blk - P * HP;
                int p1, p2, q1, q2;
                if (P == 0) { p1 = NP - 1; p2 = r; }
                else { p1 = r + P; if (p1 >= NP - 1) p1 -= NP - 1; p2 = r - P; if (p2 < 0) p2 += NP - 1; }
                if (Q == 0) { q1 = NP - 1; q2 = r; }
                else { q1 = r + Q; if (q1 >= NP - 1) q1 -= NP - 1; q2 = r - Q; if (q2 < 0) q2 += NP - 1; }
                const double a11 = K[p1 * LD + q1], a12 = K[p1 * LD + q2];
                const double a21 = K[p2 * LD + q1], a22 = K[p2 * LD + q2];
                double b11, b12, b21, b22;
                if (P == Q) {
                    const double t = cs[3 * P + 2];
                    b11 = a11 - t * a12; b22 = a22 + t * a12; b12 = 0.0; b21 = 0.0;
                } else {
                    const double cP = cs[3 * P], sP = cs[3 * P + 1], cQ = cs[3 * Q], sQ = cs[3 * Q + 1];
                    const double r11 = cP * a11 - sP * a21, r21 = sP * a11 + cP * a21;
                    const double r12 = cP * a12 - sP * a22, r22 = sP * a12 + cP * a22;
                    b11 = cQ * r11 - sQ * r12; b12 = sQ * r11 + cQ * r12;
                    b21 = cQ * r21 - sQ * r22; b22 = sQ * r21 + cQ * r22;
                }
                K[p1 * LD + q1] = b11; K[p1 * LD + q2] = b12;
                K[p2 * LD + q1] = b21; K[p2 * LD + q2] = b22;
            }
            wave_sync();
        }
    }
    // -- weights and nodes (quadtures.py:133)
    double* X = S + L::oX;
    double* W = S + L::oW;
    const double qnan = __builtin_nan("");
    for (int i = l; i < N; i += G) {
        const double v = v0[i];
        W[i] = poisoned ? qnan : v * v;
        X[i] = poisoned ? qnan : scale * K[i * LD + i] + mean;
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// model pieces
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double horner(const double* __restrict__ row, const int degree, const double u) {
    double acc = row[degree];
    for (int j = degree - 1; j >= 0; --j) acc = fma(acc, u, row[j]);
    return acc;
}

// log(y!) for a Poisson count y (a non-negative integer stored as a double).  ocml's lgamma costs ~130 VGPRs when
// inlined into the hot loop; counts are small, so a short sum of logs (exact to rounding) does the job, with the
// Stirling series taking over for large y where it is accurate to < 1 ulp.
__device__ __forceinline__ double log_factorial(const double y) {
    if (y < 32.5) {
        double acc = 0.0;
        for (double k = 2.0; k <= y + 0.5; k += 1.0) acc += log(k);
        return acc;
    }
    const double z = y + 1.0, iz = 1.0 / z, iz2 = iz * iz;
    const double series = iz * (1.0 / 12.0 - iz2 * (1.0 / 360.0 - iz2 * (1.0 / 1260.0 - iz2 * (1.0 / 1680.0 - iz2 * (1.0 / 1188.0)))));
    return (z - 0.5) * log(z) - z + 0.91893853320467274178 + series;
}

__device__ __forceinline__ double likelihood(const int kind, const double* __restrict__ lp, const double y,
                                             const double x) {
    if (kind == MFS_LIK_BERNOULLI_LOGISTIC) {
        const double z = lp[0] + x * (lp[1] + x * (lp[2] + x * lp[3]));
        const double p = 1.0 / (1.0 + exp(-z));
        return (y > 0.5) ? p : 1.0 - p;
    } else if (kind == MFS_LIK_POISSON_SOFTPLUS) {
        const double rate = log(1.0 + exp(lp[0] * x));
        return exp(y * log(rate) - rate - log_factorial(y));
    } else {
        const double r = y - (lp[0] * x + lp[1]);
        return exp(-0.5 * r * r / lp[2]) * rsqrt(6.283185307179586476925 * lp[2]);
    }
}

// Per-node transition moments, written as weighted contributions TAB[i][n] = w_i E[((X' - c) / sc)^n | x_i].
// Also returns this lane's partial sums of w mu(x) and w var(x) through mu_part / var_part when asked.
template <int N, int G>
__device__ __forceinline__ void node_cond_mean_var(const Filter1dArgs& a, const double* __restrict__ S, const int l,
                                   double& mu_part, double& var_part) {
    using L = Tile<N>;
    const double* coef = S + L::oCoef;
    const int J1 = a.degree + 1;
    mu_part = 0.0;
    var_part = 0.0;
    for (int i = l; i < N; i += G) {
        const double x = S[L::oX + i], w = S[L::oW + i];
        const double u = (a.umap == MFS_U_TANH) ? tanh(x) : x;
        double mu, var;
        if (a.trans_kind == MFS_TRANS_OPERATOR) {
            mu = x + horner(coef, a.degree, u);
            var = horner(coef + a.n_terms * J1, a.degree, u);
        } else {
            mu = a.mean_x_coef * x + horner(coef, a.degree, u);
            var = horner(coef + J1, a.degree, u);
        }
        mu_part += w * mu;
        var_part += w * var;
    }
}

template <int N, int G>
__device__ __forceinline__ void predict_contributions(const Filter1dArgs& a, double* __restrict__ S, const int l, const double c,
                                      const double inv_sc) {
    using L = Tile<N>;
    constexpr int M2 = L::M2;
    const double* coef = S + L::oCoef;
    double* TAB = S + L::oTab;
    const int J1 = a.degree + 1;
    for (int i = l; i < N; i += G) {
        const double x = S[L::oX + i], w = S[L::oW + i];
        const double u = (a.umap == MFS_U_TANH) ? tanh(x) : x;
        if (a.trans_kind == MFS_TRANS_OPERATOR) {
            // E[(X' - c)^n | x] = sum_{k=0}^{K} Q_k(u) n!/(n-k)! (x - c)^(n-k), Q_0 = 1
            double Q[MFS_MAX_TERMS + 1];
            Q[0] = 1.0;
#pragma unroll
            for (int k = 1; k <= MFS_MAX_TERMS; ++k) Q[k] = (k <= a.n_terms) ? horner(coef + (k - 1) * J1, a.degree, u) : 0.0;
            const double dx = x - c;
            // D[k] = d^k/dx^k (dx^n) = n!/(n-k)! dx^(n-k), advanced in n by the Leibniz rule
            //   D_k(n+1) = dx D_k(n) + k D_{k-1}(n)
            double D[MFS_MAX_TERMS + 1];
            D[0] = 1.0;
#pragma unroll
            for (int k = 1; k <= MFS_MAX_TERMS; ++k) D[k] = 0.0;
            double sc_n = w;  // w / sc^n
            for (int n = 0; n < M2; ++n) {
                double val = 0.0;
#pragma unroll
                for (int k = MFS_MAX_TERMS; k >= 0; --k) val = fma(Q[k], D[k], val);
                TAB[i * M2 + n] = sc_n * val;
                sc_n *= inv_sc;
#pragma unroll
                for (int k = MFS_MAX_TERMS; k >= 1; --k) D[k] = fma(dx, D[k], (double)k * D[k - 1]);
                D[0] *= dx;
            }
        } else {
            // normal closure: E_0 = 1, E_1 = m, E_n = m E_{n-1} + (n-1) v E_{n-2}, m = mu(x) - c
            const double m = a.mean_x_coef * x + horner(coef, a.degree, u) - c;
            const double v = horner(coef + J1, a.degree, u);
            double e2 = 1.0, e1 = m;
            double sc_n = w;
            TAB[i * M2] = sc_n;
            sc_n *= inv_sc;
            TAB[i * M2 + 1] = sc_n * m;
            for (int n = 2; n < M2; ++n) {
                const double e = fma(m, e1, (double)(n - 1) * v * e2);
                sc_n *= inv_sc;
                TAB[i * M2 + n] = sc_n * e;
                e2 = e1;
                e1 = e;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// the filter kernel
// ---------------------------------------------------------------------------------------------------------------
template <int N, int G, int WPB>
__global__ __launch_bounds__(WPB * 64, 2) void filter1d_kernel(const Filter1dArgs a) {
    using L = Tile<N>;
    constexpr int M2 = L::M2;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.B) return;  // whole groups leave together; no block-level barrier is ever used
    double* S = smem + (size_t)slot * L::kDoubles;
    double* mom = S + L::oMom;
    double* TAB = S + L::oTab;

    // -- stage the model tables and the carry
    {
        const double* src = a.coef + (a.coef_batched ? (size_t)b * a.n_rows * (a.degree + 1) : 0);
        for (int e = l; e < a.n_rows * (a.degree + 1); e += G) S[L::oCoef + e] = src[e];
        const double* ls = a.lik + (a.lik_batched ? (size_t)b * a.n_lik : 0);
        for (int e = l; e < MFS_MAX_LIK; e += G) S[L::oLik + e] = (e < a.n_lik) ? ls[e] : 0.0;
    }
    double mean = 0.0, scale = 1.0, nell = 0.0;
    int first_nan = -1;
    if (a.t_begin == 0) {
        const double* src = a.m0 + (a.m0_batched ? (size_t)b * (M2 + a.extra) : 0);
        for (int n = l; n < M2; n += G) mom[n] = src[n];
        if (a.mode != MFS_MODE_RAW) mean = a.mean0[a.m0_batched ? b : 0];
        if (a.mode == MFS_MODE_SCALED) scale = a.scale0[a.m0_batched ? b : 0];
    } else {
        for (int n = l; n < M2; n += G) mom[n] = a.c_mom[(size_t)b * M2 + n];
        mean = a.c_mean[b];
        scale = a.c_scale[b];
        nell = a.c_nell[b];
        first_nan = a.c_first_nan[b];
    }
    wave_sync();
    const double* lp = S + L::oLik;
    const double* yrow = a.ys + (size_t)b * a.T;
    bool dead = (first_nan >= 0);
    const double qnan = __builtin_nan("");
    // An odd number of moments (the reference warns and proceeds, mfs/one_dim/filtering.py:65-66): N = floor(M / 2)
    // (mfs/one_dim/quadtures.py:122), the rules are built from the first 2N moments, and the order-2N entry of every step is
    // the N-node rule's value of that moment -- computed and carried upstream, but never read by a quadrature.
    double tail = 0.0;

    for (int t = a.t_begin; t < a.t_end; ++t) {
        const double y = yrow[t];
        if (!dead) {
            // Two half-steps share ONE inlined quadrature (halves the code size of the hot loop):
            //   half 0 = prediction (filtering.py:76-79 / 144-148 / 221-225)
            //   half 1 = update     (filtering.py:82-85 / 151-157 / 228-236)
            int bad = 0;
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
                quadrature<N, G>(S, l, mean, scale, a.stable);
                double c = 0.0, inv_sc = 1.0, py = 1.0;
                if (half == 0) {
                    if (a.mode != MFS_MODE_RAW) {
                        double mu_part, var_part;
                        node_cond_mean_var<N, G>(a, S, l, mu_part, var_part);
                        mean = group_sum<G>(mu_part);
                        c = mean;
                        if (a.mode == MFS_MODE_SCALED) {
                            scale = sqrt(group_sum<G>(var_part));
                            inv_sc = 1.0 / scale;
                        }
                    }
                    predict_contributions<N, G>(a, S, l, c, inv_sc);
                } else {
                    double pp = 0.0, mx = 0.0;
                    for (int i = l; i < N; i += G) {
                        const double x = S[L::oX + i];
                        const double wl = S[L::oW + i] * likelihood(a.lik_kind, lp, y, x);
                        S[L::oW + i] = wl;  // weights now carry the likelihood
                        pp += wl;
                        mx += wl * x;
                    }
                    py = group_sum<G>(pp);
                    if (a.mode != MFS_MODE_RAW) {
                        mean = group_sum<G>(mx) / py;
                        c = mean;
                    }
                    if (a.mode == MFS_MODE_SCALED) {
                        double v = 0.0;
                        for (int i = l; i < N; i += G) {
                            const double dx = S[L::oX + i] - c;
                            v += S[L::oW + i] * dx * dx;
                        }
                        scale = sqrt(group_sum<G>(v) / py);
                        inv_sc = 1.0 / scale;
                    }
                    for (int i = l; i < N; i += G) {
                        const double dx = (S[L::oX + i] - c) * inv_sc;
                        double p = S[L::oW + i];
                        for (int n = 0; n < M2; ++n) {
                            TAB[i * M2 + n] = p;
                            p *= dx;
                        }
                        if (a.extra) S[L::oV0 + i] = p;      // wl dx^(2N): the first eigenvector row is free after the quadrature
                    }
                    nell -= log(py);
                }
                wave_sync();
                for (int n = l; n < M2; n += G) {
                    double acc = 0.0;
                    for (int i = 0; i < N; ++i) acc += TAB[i * M2 + n];
                    acc /= py;
                    mom[n] = acc;
                    bad |= !finite(acc);
                }
                if (a.extra && half == 1) {
                    double acc = 0.0;
                    for (int i = 0; i < N; ++i) acc += S[L::oV0 + i];
                    tail = acc / py;
                }
                wave_sync();
            }
            bad |= (int)(!finite(nell) || !finite(mean) || !finite(scale));
            bad = group_or<G>(bad);
            if (bad) { dead = true; first_nan = t; }
        } else {
            // NaN-poisoned replicate: the reference keeps emitting NaN for every later step
            for (int n = l; n < M2; n += G) mom[n] = qnan;
            mean = qnan; scale = qnan; nell = qnan; tail = qnan;
            wave_sync();
        }
        // ---- stream the step's results to HBM: 2N contiguous doubles per filter (+ mean, scale)
        if (a.out_mom) {
            double* dst = a.out_mom + ((size_t)b * a.T + t) * (M2 + a.extra);
            for (int n = l; n < M2; n += G) dst[n] = mom[n];
            if (a.extra && l == 0) dst[M2] = dead ? qnan : tail;
        }
        if (l == 0) {
            if (a.out_mean) a.out_mean[(size_t)b * a.T + t] = mean;
            if (a.out_scale) a.out_scale[(size_t)b * a.T + t] = scale;
        }
    }
    // -- carry / final results
    if (a.t_end >= a.T) {
        if (l == 0) {
            a.out_nell[b] = nell;
            if (a.out_first_nan) a.out_first_nan[b] = first_nan;
        }
    } else {
        for (int n = l; n < M2; n += G) a.c_mom[(size_t)b * M2 + n] = mom[n];
        if (l == 0) {
            a.c_mean[b] = mean; a.c_scale[b] = scale; a.c_nell[b] = nell; a.c_first_nan[b] = first_nan;
        }
    }
}

template <int N, int G, int WPB>
__global__ __launch_bounds__(WPB * 64) void quadrature1d_kernel(const Quad1dArgs a) {
    using L = Tile<N>;
    constexpr int M2 = L::M2;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.B) return;
    double* S = smem + (size_t)slot * L::kDoubles;
    for (int n = l; n < M2; n += G) S[L::oMom + n] = a.ms[(size_t)b * M2 + n];
    wave_sync();
    const double mean = a.mean ? a.mean[b] : 0.0;
    const double scale = a.scale ? a.scale[b] : 1.0;
    quadrature<N, G>(S, l, mean, scale, a.stable);
    for (int i = l; i < N; i += G) {
        a.out_w[(size_t)b * N + i] = S[L::oW + i];
        a.out_x[(size_t)b * N + i] = S[L::oX + i];
    }
}

// launcher signature shared by the per-N translation units
using Filter1dLaunch = hipError_t (*)(const Filter1dArgs&, int grid, int lds_bytes, hipStream_t);
using Quad1dLaunch = hipError_t (*)(const Quad1dArgs&, int grid, int lds_bytes, hipStream_t);

struct KernelEntry {
    Filter1dLaunch filter;
    Quad1dLaunch quad;
    int lds_doubles_per_filter;  // dense: complete; fast: fixed part, the model table is added at launch
    int waves_per_block;
    int lanes_per_filter;
};

// registry slots per N: [0..2] dense path with G = 16 / 32 / 64, [3..5] fast path with G = 16 / 32 / 64,
// [6] fast path with G = 8 (N <= 7: eight filters per wavefront)
constexpr int kSlots = 7;
using Filter1dFastLaunch = hipError_t (*)(const Filter1dArgs&, int grid, int lds_doubles_per_filter, hipStream_t);

}  // namespace mfs
