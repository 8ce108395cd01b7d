// filternd_kernel.hpp -- hand-written HIP for gfx950: the N-D (d = 2) moment-filter time-step loop.
//
// Reference: mfs/multi_dims/filtering.py:210-344 (scan bodies :258-277, :326-341, :181-204),
// mfs/multi_dims/quadratures.py:120-178 (moment_quadrature_nd), mfs/multi_dims/moments.py:414-479 (TME transition).
//
// One filter per 256-thread workgroup (4 waves); the k-loop runs inside the kernel.  Per half step:
//   quadrature_nd  G = ms[inds[0]], H_k = ms[inds[1+k]] gathered from the LDS moment vector; Cholesky and both triangular
//                  solves on one wave in registers (columns of L reach the other lanes as DPP operands after a lane-swap
//                  duplication); K_k' = V_prev^T K_k V_prev warm start; both K_k diagonalised TOGETHER by cyclic Jacobi
//                  (round-robin tournament from an LDS index table, one item = a 2x2 block of K_m + two eigenvector
//                  row-pairs: 2 (s/2)^2 items <= 256 threads at s = 21) with the eigenvector matrices accumulated;
//                  s^2 tensor-product nodes with weights <v0_i, v1_j> v0_i[0] v1_j[0]      (quadratures.py:165-170)
//   predict        per node, sum_kappa Q_kappa(x) prod_k n_k!/(n_k-kappa_k)! (x_k-c_k)^(n_k-kappa_k) in its factorised
//                  form (operator tables, moments.py:414-479) or the Stein recursion of a Normal closure (:257-411),
//                  unrolled over the z moments so that every power / coefficient index is a compile-time constant
//   update         likelihood-weighted raw / central / scaled-central moments
// with the z moment sums formed 16 at a time through a transposing DPP row reduction into a [16 rows][z] LDS table.
// No MFMA: s <= 28, fp64, sequential.
#pragma once
#include "filter1d_fast.hpp"

namespace mfs {

#ifdef MFS_ND_STAMPS
// diagnostic build (tools/diag/nd_stamps.hip): cycles per phase, accumulated by thread 0 of block 0
__device__ unsigned long long g_nd_stamps[16];
__device__ unsigned long long g_nd_hist[8][40];   // [test index][-log10(off / dia), clamped]: Jacobi convergence tests of block 0
#define ND_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_nd_stamps[slot] += now_ - t_last_; t_last_ = now_; } } while (0)
#define ND_STAMP_BEGIN unsigned long long t_last_ = clock64()
#else
#define ND_STAMP(slot) do {} while (0)
#define ND_STAMP_BEGIN do {} while (0)
#endif

struct FilterNdArgs {
    int mode, T, B, stable;
    int n_terms_used, D;      // coefficient block extent per variable (degree + 1)
    int lik_kind, n_lik, lik_component;
    int ext[16];              // per-block true extents (ea | eb << 8) of the coefficient blocks; 0 = empty block
    const double* coef;       // [kNdRows][D][D]: rows 0..13 Q_kappa in the fixed kappa order below (zeros where the
                              // model has no term), rows 14, 15 the conditional variances of X'_0, X'_1 (scaled mode)
    const double* lik;        // [n_lik]
    const int32_t* inds;      // [3][s][s]
    const double* m0;         // [z] or [B][z]
    int m0_batched;
    const double* mean0;      // [2] or [B][2]
    const double* scale0;     // [2] or [B][2] (scaled mode)
    const double* ys;         // [B][T]
    double* out_mom;          // [B][T][z]
    double* out_mean;         // [B][T][2]
    double* out_scale;        // [B][T][2] (scaled mode)
    double* out_nell;
    int32_t* out_first_nan;
};

// derivative multi-indices kappa with 1 <= |kappa| <= 4, graded-lex order (the order the host fills `coef` in)
constexpr int kNdTerms = 14;
constexpr int kNdRows = 16;
#ifndef MFS_ND_JACOBI_TOL
#define MFS_ND_JACOBI_TOL 1e-31
#endif
// off-diagonal / diagonal mass (squared Frobenius norms) at which the Jacobi sweeps stop
constexpr double kNdJacobiTol = MFS_ND_JACOBI_TOL;
#ifndef MFS_ND_FINISH_X2
#define MFS_ND_FINISH_X2 1e-14
#endif
// squared Frobenius size of the first-order eigenvector correction below which it replaces further sweeps (0: never)
constexpr double kNdFinishX2 = MFS_ND_FINISH_X2;
constexpr int kNdMaxD = 6;
__device__ constexpr int kKap0[kNdTerms] = {0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
__device__ constexpr int kKap1[kNdTerms] = {1, 0, 2, 1, 0, 3, 2, 1, 0, 4, 3, 2, 1, 0};

__device__ __forceinline__ constexpr double ffact(int n, int k) {
    double r = 1.0;
    for (int j = 0; j < k; ++j) r *= (double)(n - j);
    return r;
}

template <int N>
struct NdTile {
    static constexpr int S = N * (N + 1) / 2;        // Gram size
    static constexpr int Z = N * (2 * N + 1);        // number of moments, |n| <= 2N - 1
    static constexpr int P = 2 * N;                  // powers 0..2N-1 per coordinate
    static constexpr int NP = S + (S & 1);
    static constexpr int HP = NP / 2;
    static constexpr int LD = NP + 1;
    static constexpr int R = S * S;                  // tensor-product nodes
    static constexpr int oMom = 0;
    static constexpr int oA = (Z + 1) & ~1;          // [NP][LD] G -> R
    static constexpr int oK = oA + NP * LD;          // [2][NP][LD]
    static constexpr int oV = oK + 2 * NP * LD;      // [2][NP][LD]
    static constexpr int oCs = oV + 2 * NP * LD;     // [2][HP][3]
    static constexpr int oW = oCs + 2 * HP * 3 + 2;  // [S][S] node weights
    static constexpr int oLam = oW + ((R + 1) & ~1); // [2][NP]
    static constexpr int ZB = (Z + 15) / 16;         // batches of 16 moments in the transposing reduction
    static constexpr int RW = 16 * ZB + 6;           // reduction row: moments (padded), flag, 5 scalar sums
    static constexpr int oRed = oLam + 2 * NP;       // [4 waves x 4 DPP rows][RW]
    static constexpr int oCoef = oRed + 16 * RW;     // [kNdRows][D][D]
    static constexpr int oMisc = oCoef + kNdRows * kNdMaxD * kNdMaxD;  // lik params [4], flags [4]
    // Tournament index tables, built once per launch: which rows / columns a work item touches in round r depends on
    // (r, item) only, and recomputing it cost ~50 integer instructions per thread per round next to ~30 flops.
    //   KT[r][P * HP + Q] = p1 | p2 << 8 | q1 << 16 | q2 << 24  (u32)
    static constexpr bool kTables = (2 * HP * HP <= 256) && (2 * S * HP <= 512);
    static constexpr int oIdxK = oMisc + 8;
    static constexpr int nIdxK = kTables ? ((NP - 1) * HP * HP * 4 + 7) / 8 : 0;
    static constexpr int oIdxV = oIdxK + nIdxK;
    static constexpr int nIdxV = 0;
    static constexpr int kDoubles = oIdxV + nIdxV;
};

__device__ __forceinline__ double wave_sum64(double v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    // every lane of a 16-row now holds its row sum; add the four rows through scalar registers
    double tot = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * r);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * r);
        tot += __hiloint2double(hi, lo);
    }
    return tot;
}

// Transposing reduction of 16 values per lane across each 16-lane DPP row: at every step a lane keeps one value of a
// pair and sends the other to its partner, so after four steps it holds ONE of the 16 values summed over its row --
// value number bitrev4(lane & 15).  ~7 instructions per pair, 15 pairs, against 16 x (4 DPP steps + 4 readlane pairs)
// for value-at-a-time wave sums.  Partners (row_mirror, row_half_mirror, quad [3,2,1,0], quad [1,0,3,2]) are chosen so
// that the two lanes of a pair always hold the same subset of values.
__device__ __forceinline__ double row_reduce16(const double (&b)[16], const int lane) {
    const bool s3 = (lane & 8) != 0, s2 = (lane & 4) != 0, s1 = (lane & 2) != 0, s0 = (lane & 1) != 0;
    double r1[8], r2[4], r3[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double keep = s3 ? b[2 * k + 1] : b[2 * k], send = s3 ? b[2 * k] : b[2 * k + 1];
        r1[k] = keep + dpp_move<0x140>(send);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double keep = s2 ? r1[2 * k + 1] : r1[2 * k], send = s2 ? r1[2 * k] : r1[2 * k + 1];
        r2[k] = keep + dpp_move<0x141>(send);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double keep = s1 ? r2[2 * k + 1] : r2[2 * k], send = s1 ? r2[2 * k] : r2[2 * k + 1];
        r3[k] = keep + dpp_move<0x1B>(send);
    }
    const double keep = s0 ? r3[1] : r3[0], send = s0 ? r3[0] : r3[1];
    return keep + dpp_move<0xB1>(send);
}

__device__ __forceinline__ double poly2d(const double* __restrict__ c, const int D, const int ext, const double x0,
                                         const double x1) {
    // sum_{a < ea, b < eb} c[a][b] x0^a x1^b by nested Horner; ext = ea | eb << 8 are the true extents of this block
    // (most Q_kappa of a low-order TME have degree <= 2 although the common block is D x D)
    const int ea = ext & 0xff, eb = ext >> 8;
    double acc = 0.0;
    for (int a = ea - 1; a >= 0; --a) {
        double row = c[a * D + eb - 1];
        for (int b = eb - 2; b >= 0; --b) row = fma(row, x1, c[a * D + b]);
        acc = fma(acc, x0, row);
    }
    return acc;
}

// the Bernoulli-logistic likelihood (mfs/multi_dims/ss_models.py:63-67) with the in-line exponential; other kinds through
// the generic routine
__device__ __forceinline__ double likelihood_nd(const int kind, const double* __restrict__ lp, const double y,
                                                const double x) {
    if (kind == MFS_LIK_BERNOULLI_LOGISTIC) {
        const double z = lp[0] + x * (lp[1] + x * (lp[2] + x * lp[3]));
        const double p = rcp_sat(1.0 + fast_exp(-z));
        return (y > 0.5) ? p : 1.0 - p;
    }
    return likelihood(kind, lp, y, x);
}

// pair p of round r in the round-robin tournament on NP players
template <int NP>
__device__ __forceinline__ void tournament_pair(const int r, const int P, int& p, int& q) {
    if (P == 0) { p = NP - 1; q = r; }
    else { p = r + P; if (p >= NP - 1) p -= NP - 1; q = r - P; if (q < 0) q += NP - 1; }
}

// Fills lam[2][NP] and W[S][S]; returns block-uniform poison flag.
template <int N>
__device__ bool quadrature_nd(double* __restrict__ Sm, const int32_t* __restrict__ inds, const int stable,
                              const bool warm) {
    using L = NdTile<N>;
    constexpr int S = L::S, NP = L::NP, HP = L::HP, LD = L::LD;
    const int tid = threadIdx.x, nthr = blockDim.x;
    double* mom = Sm + L::oMom;
    double* A = Sm + L::oA;
    double* K = Sm + L::oK;
    double* V = Sm + L::oV;
    double* cs = Sm + L::oCs;
    double* flags = Sm + L::oMisc + 4;

    ND_STAMP_BEGIN;
    // -- gather (quadratures.py:151-152); pad rows / columns are zero, V = I
    for (int e = tid; e < NP * NP; e += nthr) {
        const int i = e / NP, j = e - i * NP;
        const bool in = (i < S) && (j < S);
        A[i * LD + j] = in ? mom[inds[i * S + j]] : 0.0;
        K[i * LD + j] = in ? mom[inds[S * S + i * S + j]] : 0.0;
        K[NP * LD + i * LD + j] = in ? mom[inds[2 * S * S + i * S + j]] : 0.0;
        if (!warm) {
            V[i * LD + j] = (i == j) ? 1.0 : 0.0;
            V[NP * LD + i * LD + j] = (i == j) ? 1.0 : 0.0;
        }
    }
    if (tid == 0) flags[0] = 0.0;
    __syncthreads();

    ND_STAMP(0);
    // -- Cholesky (quadratures.py:154) or LDL^T completion (mfs/utils.py:495-538)
    if (!stable) {
        // Register-resident front end on ONE wave, no barriers: lane i (< S) owns row i of G -> L; lane c (< 2S) owns
        // column c of [H_0 | H_1].  Column step j needs L[j][k] in every lane, twice: for the Cholesky dot products
        // and for the forward substitution X = R^-1 [H_0 | H_1] (quadratures.py:156-161, inner solve) fused into
        // the same sweep.  As soon as column k of L is final, gfx950's lane swaps (v_permlane16_swap / 32_swap on
        // (v, copy of v)) spread DPP row 0 and DPP row 1 of it over all four rows of the wave, once; every later use
        // is then the row_newbcast operand of the multiply-add itself (one instruction per term instead of two
        // v_readlane + a multiply-add).  The outer solve K^T = R^-1 X^T reuses them after a transpose through the K tiles.
        if (tid < 64) {
            const int li = (tid < S) ? tid : S - 1;
            const int hc = (tid < 2 * S) ? tid : 0, hm = hc / S, hj = hc - hm * S;
            static_assert(S <= 32, "rows of L live in DPP rows 0 and 1");
            constexpr int S0 = (S < 16) ? S : 16;     // columns whose entries are needed from DPP row 0 (rows j < 16)
            double Lr[S], xc[S];
            double* rinv_lds = Sm + L::oRed;          // 1 / L_jj, parked in LDS between the two solves (registers are short)
            double D0[S0], D1[(S > 16) ? S : 1];      // column k of L: DPP row 0 / DPP row 1 of it in all four rows
            static_for<0, S>([&](auto Jc) { Lr[Jc] = A[li * LD + Jc]; xc[Jc] = K[hm * NP * LD + Jc * LD + hj]; });
            bool bad = false;
            // right-looking: a final column is subtracted from every later column at once (blocks of eight fused DPP
            // multiply-adds behind one hazard nop, independent accumulators) -- each column still receives its terms in
            // the order k = 0, 1, ..., so the numbers are those of the column-by-column form
            constexpr int loE = (S < 16) ? S : 16;
            static_for<0, S>([&](auto Jc) {
                constexpr int j = Jc;
                const double s = Lr[j], xa = xc[j];
                const double pj = bcast<64, j>(s);
                bad |= !(pj > 0.0);
                const double rinv = rsq_nr(pj);
                if (tid == 0) rinv_lds[j] = rinv;
                Lr[j] = s * rinv;       // row j itself gets sqrt(piv) = piv * rinv
                xc[j] = xa * rinv;
                if constexpr (j + 1 < S) {   // column j is final: spread its two DPP rows over the wave
                    double ev, od, lo, up;
                    row_dup(Lr[j], ev, od);
                    if constexpr (j < S0) { half_dup(ev, lo, up); D0[j] = lo; }
                    if constexpr (S > 16) { half_dup(od, lo, up); D1[j] = lo; }
                    if constexpr (j + 1 < loE) {
                        fnma_bcast_range<j + 1, loE, 0>(Lr, Lr[j], D0[j]);
                        fnma_bcast_range<j + 1, loE, 0>(xc, xc[j], D0[j]);
                    }
                    if constexpr (S > 16) {
                        constexpr int hi0 = (j + 1 > 16) ? j + 1 : 16;
                        fnma_bcast_range<hi0, S, 16>(Lr, Lr[j], D1[j]);
                        fnma_bcast_range<hi0, S, 16>(xc, xc[j], D1[j]);
                    }
                }
            });
            if (tid == 0 && bad) flags[0] = 1.0;
            // transpose X through the K tiles: column hc of X_m -> K_m[:, hj]
            static_for<0, S>([&](auto Jc) { if (tid < 2 * S) K[hm * NP * LD + Jc * LD + hj] = xc[Jc]; });
            wave_sync();
            // row hj of X_m is column hj of X_m^T: forward-substitute it, K_m[hj][:] = (R^-1 X_m^T)[:, hj]
            double yr[S];
            static_for<0, S>([&](auto Jc) { yr[Jc] = K[hm * NP * LD + hj * LD + Jc]; });
            static_for<0, S>([&](auto Ic) {
                constexpr int i = Ic;
                yr[i] = yr[i] * rinv_lds[i];
                if constexpr (i + 1 < loE) fnma_bcast_range<i + 1, loE, 0>(yr, yr[i], D0[i]);
                if constexpr (S > 16 && i + 1 < S) {
                    constexpr int hi0 = (i + 1 > 16) ? i + 1 : 16;
                    fnma_bcast_range<hi0, S, 16>(yr, yr[i], D1[i]);
                }
            });
            wave_sync();
            static_for<0, S>([&](auto Jc) { if (tid < 2 * S) K[hm * NP * LD + hj * LD + Jc] = yr[Jc]; });
        }
        __syncthreads();
    } else {
        double fro = 0.0;
        for (int e = tid; e < S * S; e += nthr) { const double v = A[(e / S) * LD + (e % S)]; fro += v * v; }
        fro = wave_sum64(fro);
        if ((tid & 63) == 0) Sm[L::oRed + (tid >> 6)] = fro;
        __syncthreads();
        const double eps = 1e-8 * sqrt(Sm[L::oRed] + Sm[L::oRed + 1] + Sm[L::oRed + 2] + Sm[L::oRed + 3]);
        __syncthreads();
        for (int j = 0; j < S; ++j) {
            if (tid >= j && tid < S) {
                double s = A[tid * LD + j];
                for (int k = 0; k < j; ++k) s -= A[tid * LD + k] * (A[j * LD + k] * A[k * LD + k]);
                A[tid * LD + j] = s;
            }
            __syncthreads();
            const double dj = A[j * LD + j];
            __syncthreads();
            if (tid > j && tid < S) A[tid * LD + j] = A[tid * LD + j] / dj;
            __syncthreads();
        }
        for (int j = 0; j < S; ++j) {
            const double dj = A[j * LD + j];
            const double fj = (dj < 0.0) ? eps : sqrt(dj);
            __syncthreads();
            if (tid >= j && tid < S) A[tid * LD + j] = (tid == j) ? fj : A[tid * LD + j] * fj;
            __syncthreads();
        }
        // -- K_k = R^-1 H_k R^-T (quadratures.py:156-161): columns then rows, both matrices at once
        if (tid < 2 * S) {
            double* Kk = K + (tid / S) * NP * LD;
            const int c = tid % S;
            for (int i = 0; i < S; ++i) {
                double s = Kk[i * LD + c];
                for (int k = 0; k < i; ++k) s -= A[i * LD + k] * Kk[k * LD + c];
                Kk[i * LD + c] = s / A[i * LD + i];
            }
        }
        __syncthreads();
        if (tid < 2 * S) {
            double* Kk = K + (tid / S) * NP * LD;
            const int i = tid % S;
            for (int j = 0; j < S; ++j) {
                double s = Kk[i * LD + j];
                for (int k = 0; k < j; ++k) s -= Kk[i * LD + k] * A[j * LD + k];
                Kk[i * LD + j] = s / A[j * LD + j];
            }
        }
        __syncthreads();
    }

    ND_STAMP(1);
    for (int e = tid; e < 2 * S * S; e += nthr) {
        double* Kk = K + (e / (S * S)) * NP * LD;
        const int f = e % (S * S), i = f / S, j = f - i * S;
        if (i > j) {
            const double s = 0.5 * (Kk[i * LD + j] + Kk[j * LD + i]);
            Kk[i * LD + j] = s;
            Kk[j * LD + i] = s;
        }
    }
    __syncthreads();

    ND_STAMP(2);
    // -- warm start: the eigenvector matrices of the previous rule are still in LDS.  K changes little between
    //    consecutive rules, so V_prev^T K V_prev is already nearly diagonal and the sweeps below converge
    //    quadratically from there (2-3 sweeps instead of 8-9); V then accumulates on top of V_prev.  The tile of the
    //    Cholesky factor is free by now and serves as the temporary.
    if (warm) {
        // both matrices in the same two stages (the weight tile, free until the end of the rule, is the second scratch)
        double* A1 = Sm + L::oW;   // [S][S]
        for (int e = tid; e < 2 * S * S; e += nthr) {   // A_m = K_m V_m
            const int m = e / (S * S), f = e - m * S * S, i = f / S, j = f - i * S;
            const double* Kk = K + m * NP * LD;
            const double* Vk = V + m * NP * LD;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) acc = fma(Kk[i * LD + k], Vk[k * LD + j], acc);
            if (m == 0) A[i * LD + j] = acc; else A1[i * S + j] = acc;
        }
        __syncthreads();
        constexpr int TRI = S * (S + 1) / 2;
        for (int e = tid; e < 2 * TRI; e += nthr) {     // K_m = V_m^T A_m on the lower triangle, symmetrised on the fly
            const int m = e / TRI, t = e - m * TRI;
            int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            i += ((i + 1) * (i + 2) / 2 <= t) ? 1 : 0;
            i -= (i * (i + 1) / 2 > t) ? 1 : 0;
            const int j = t - i * (i + 1) / 2;
            double* Kk = K + m * NP * LD;
            const double* Vk = V + m * NP * LD;
            const double* Am = (m == 0) ? A : A1;
            const int lda = (m == 0) ? LD : S;
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) {
                a1 = fma(Vk[k * LD + i], Am[k * lda + j], a1);
                a2 = fma(Vk[k * LD + j], Am[k * lda + i], a2);
            }
            const double v = 0.5 * (a1 + a2);
            Kk[i * LD + j] = v;
            Kk[j * LD + i] = v;
        }
        __syncthreads();
    }

    ND_STAMP(3);
    // -- cyclic Jacobi on both matrices with eigenvectors (quadratures.py:163)
    double prev_off = 1.79e308;
    const int kw_slot = (tid < 2 * HP * HP) ? tid % (HP * HP) : 0;
    unsigned kw_next = 0;
    if constexpr (L::kTables) kw_next = reinterpret_cast<const unsigned*>(Sm + L::oIdxK)[kw_slot];
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        // (after a warm start the first two sweeps are always needed -- the off-diagonal mass goes 1e-6 -> 1e-12 -> 1e-24
        //  of the diagonal's -- so the convergence test, a pass over both matrices and two barriers, starts at the third)
        if (!(warm && sweep < 2)) {
        // off / dia: off-diagonal and diagonal mass; xsq = sum_{i != j} (K_ij / (K_jj - K_ii))^2, the squared size of the
        // first-order eigenvector correction (below)
        double off = 0.0, dia = 0.0, xsq = 0.0;
        for (int e = tid; e < 2 * S * S; e += nthr) {
            const double* Kk = K + (e / (S * S)) * NP * LD;
            const int f = e % (S * S), i = f / S, j = f - i * S;
            const double v = Kk[i * LD + j];
            const double dd = Kk[j * (LD + 1)] - Kk[i * (LD + 1)];
            if (i == j) dia += v * v;
            else { off += v * v; xsq += (v * v) * __builtin_amdgcn_rcp(dd * dd); }
        }
        off = wave_sum64(off);
        dia = wave_sum64(dia);
        xsq = wave_sum64(xsq);
        __syncthreads();  // previous readers of the scratch slots are done
        if ((tid & 63) == 0) {
            Sm[L::oRed + 3 * (tid >> 6)] = off; Sm[L::oRed + 3 * (tid >> 6) + 1] = dia; Sm[L::oRed + 3 * (tid >> 6) + 2] = xsq;
        }
        __syncthreads();
        off = Sm[L::oRed] + Sm[L::oRed + 3] + Sm[L::oRed + 6] + Sm[L::oRed + 9];
        dia = Sm[L::oRed + 1] + Sm[L::oRed + 4] + Sm[L::oRed + 7] + Sm[L::oRed + 10];
        xsq = Sm[L::oRed + 2] + Sm[L::oRed + 5] + Sm[L::oRed + 8] + Sm[L::oRed + 11];
#ifdef MFS_ND_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            int bin = (off > 0.0 && dia > 0.0) ? (int)(-log10(off / dia)) : 39;
            bin = bin < 0 ? 0 : bin > 39 ? 39 : bin;
            g_nd_hist[sweep < 8 ? sweep : 7][bin] += 1;
        }
#endif
        if (!finite(off + dia)) { if (tid == 0) flags[0] = 1.0; break; }
        if (!(off > kNdJacobiTol * dia)) break;
        if (off < 1e-26 * dia && off > 0.25 * prev_off) break;
        if (xsq <= kNdFinishX2) {
            // Nearly diagonal, K = D + E with every |E_ij / (d_j - d_i)| < 1e-7: the eigenvectors are the columns of
            // I + X, X_ij = E_ij / (d_j - d_i) (antisymmetric, so I + X is orthogonal up to X^T X <= 1e-14), the
            // eigenvalues d_j + sum_i E_ij X_ij -- as accurate as one more sweep at the price of one S x S x S product
            // per matrix instead of S - 1 rounds of rotations (close eigenvalues make X large and take the sweep)
            for (int e = tid; e < 2 * S * S; e += nthr) {      // X over the off-diagonal of K (the diagonal stays)
                double* Kk = K + (e / (S * S)) * NP * LD;
                const int f = e % (S * S), i = f / S, j = f - i * S;
                if (i != j) Kk[i * LD + j] = Kk[i * LD + j] / (Kk[j * (LD + 1)] - Kk[i * (LD + 1)]);
            }
            __syncthreads();
            double* T1 = Sm + L::oW;
            for (int e = tid; e < 2 * S * S; e += nthr) {      // T_m = V_m X_m
                const int m = e / (S * S), f = e - m * S * S, k = f / S, j = f - k * S;
                const double* Kk = K + m * NP * LD;
                const double* Vk = V + m * NP * LD;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < S; ++i) acc = fma(Vk[k * LD + i], (i == j) ? 0.0 : Kk[i * LD + j], acc);
                if (m == 0) A[k * LD + j] = acc; else T1[k * S + j] = acc;
            }
            for (int e = tid; e < 2 * S; e += nthr) {          // second-order eigenvalues, parked in the rotation records
                const int m = e / S, j = e - m * S;
                const double* Kk = K + m * NP * LD;
                const double dj = Kk[j * (LD + 1)];
                double acc = dj;
                for (int i = 0; i < S; ++i) {
                    const double x = (i == j) ? 0.0 : Kk[i * LD + j];
                    acc = fma((dj - Kk[i * (LD + 1)]) * x, x, acc);
                }
                cs[e] = acc;
            }
            __syncthreads();
            for (int e = tid; e < 2 * S * S; e += nthr) {
                const int m = e / (S * S), f = e - m * S * S, k = f / S, j = f - k * S;
                V[m * NP * LD + k * LD + j] += (m == 0) ? A[k * LD + j] : T1[k * S + j];
            }
            for (int e = tid; e < 2 * S; e += nthr) K[(e / S) * NP * LD + (e % S) * (LD + 1)] = cs[e];
            break;
        }
        prev_off = off;
        }
#ifdef MFS_ND_STAMPS
        if (blockIdx.x == 0 && threadIdx.x == 0) g_nd_stamps[8] += 1;
#endif

        for (int r = 0; r < NP - 1; ++r) {
            // this round's index word is fetched and unpacked before the rotations / the barrier, off the update's path
            // (the word itself was requested a round ago: its latency would otherwise stand in front of every other load)
            const unsigned kw = kw_next;
            if constexpr (L::kTables) kw_next = reinterpret_cast<const unsigned*>(Sm + L::oIdxK)[((r + 1 < NP - 1) ? r + 1 : 0) * (HP * HP) + kw_slot];
            const int p1 = kw & 255, p2 = (kw >> 8) & 255, q1 = (kw >> 16) & 255, q2 = kw >> 24;
            const int o11 = p1 * LD + q1, o12 = p1 * LD + q2, o21 = p2 * LD + q1, o22 = p2 * LD + q2;
            // ... and so are the block and the eigenvector entries the item will rotate: the rotations (below) only read
            // K and write their own records, so these loads are in flight while wave 0 works through the rotation chain
            double a11 = 0.0, a12 = 0.0, a21 = 0.0, a22 = 0.0, x0 = 0.0, x1 = 0.0, y0 = 0.0, y1 = 0.0;
            if constexpr (L::kTables) {
                const int mB = (tid < 2 * HP * HP) ? tid / (HP * HP) : 0;
                const int Q = ((tid < 2 * HP * HP) ? tid % (HP * HP) : 0) % HP;
                const double* Kk = K + mB * NP * LD;
                const double* Va = V + mB * NP * LD + Q * LD;
                const double* Vb = V + mB * NP * LD + (Q + HP) * LD;
                a11 = Kk[o11]; a12 = Kk[o12]; a21 = Kk[o21]; a22 = Kk[o22];
                x0 = Va[p1]; x1 = Va[p2]; y0 = Vb[p1]; y1 = Vb[p2];
            }
            if (tid < 2 * HP) {
                const int m = tid / HP, P = tid - m * HP;
                double* Kk = K + m * NP * LD;
                int p, q;
                tournament_pair<NP>(r, P, p, q);
                const double app = Kk[p * LD + p], aqq = Kk[q * LD + q], apq = Kk[p * LD + q];
                // t = tan of the Jacobi angle = 2 a_pq / (d + sgn(d) sqrt(d^2 + 4 a_pq^2)), d = a_qq - a_pp, through
                // reciprocal / reciprocal-square-root seeds: t only decides how well a_pq is annihilated (one Newton
                // step is plenty), c = (1 + t^2)^-1/2 decides the orthogonality of V and gets two.
                const double d = aqq - app, h = apq + apq;
                const double rr = fma(d, d, h * h);
                double c = 1.0, s = 0.0, t = 0.0;
                if (apq != 0.0 && rr > 0.0) {
                    double y = __builtin_amdgcn_rsq(rr);
                    y = fma(y, fma(-0.5 * rr * y, y, 0.5), y);
                    const double den = d + copysign(rr * y, d);
                    double z = __builtin_amdgcn_rcp(den);
                    z = fma(fma(-den, z, 1.0), z, z);
                    t = h * z;
                    t = finite(t) ? t : 0.0;
                    c = rsq_nr(fma(t, t, 1.0));
                    s = t * c;
                }
                cs[(m * HP + P) * 3] = c; cs[(m * HP + P) * 3 + 1] = s; cs[(m * HP + P) * 3 + 2] = t;
            }
            __syncthreads();
            // One straight-line section per thread: its 2x2 block of A <- J^T A J and its two eigenvector row-pairs of
            // V <- V J.  All LDS reads are issued before any arithmetic so that one latency is exposed, not three.
            if constexpr (2 * HP * HP <= 256 && 2 * S * HP <= 512) {
                // item (m, P, Q): the 2x2 block (pair P, pair Q) of K_m and the eigenvector entries of rows Q and Q + HP in
                // the two columns of pair P -- both need the rotation of pair P, so an item reads two rotations, not four
                const bool hasB = tid < 2 * HP * HP;
                const int mB = hasB ? tid / (HP * HP) : 0, blk = hasB ? tid - mB * HP * HP : 0;
                const int P = blk / HP, Q = blk - P * HP;
                double* Kk = K + mB * NP * LD;
                double* Va = V + mB * NP * LD + Q * LD;
                double* Vb = V + mB * NP * LD + (Q + HP) * LD;
                const double* csB = cs + mB * HP * 3;
                // ---- the two rotations
                const double cP = csB[3 * P], sP = csB[3 * P + 1], tP = csB[3 * P + 2];
                const double cQ = csB[3 * Q], sQ = csB[3 * Q + 1];
                // ---- arithmetic
                double b11, b12, b21, b22;
                if (P == Q) {
                    b11 = a11 - tP * a12; b22 = a22 + tP * a12; b12 = 0.0; b21 = 0.0;
                } else {
                    const double r11 = cP * a11 - sP * a21, r21 = sP * a11 + cP * a21;
                    const double r12 = cP * a12 - sP * a22, r22 = sP * a12 + cP * a22;
                    b11 = cQ * r11 - sQ * r12; b12 = sQ * r11 + cQ * r12;
                    b21 = cQ * r21 - sQ * r22; b22 = sQ * r21 + cQ * r22;
                }
                // ---- stores (every item touches only its own elements)
                if (hasB) {
                    Kk[o11] = b11; Kk[o12] = b12;
                    Kk[o21] = b21; Kk[o22] = b22;
                    Va[p1] = cP * x0 - sP * x1; Va[p2] = sP * x0 + cP * x1;
                    Vb[p1] = cP * y0 - sP * y1; Vb[p2] = sP * y0 + cP * y1;
                }
            } else {  // larger N: the same work in strided loops
                for (int e = tid; e < 2 * HP * HP; e += nthr) {
                    const int m = e / (HP * HP), blk = e - m * HP * HP, P = blk / HP, Q = blk - P * HP;
                    double* Kk = K + m * NP * LD;
                    const double* csm = cs + m * HP * 3;
                    int p1, p2, q1, q2;
                    tournament_pair<NP>(r, P, p1, p2);
                    tournament_pair<NP>(r, Q, q1, q2);
                    const double a11 = Kk[p1 * LD + q1], a12 = Kk[p1 * LD + q2];
                    const double a21 = Kk[p2 * LD + q1], a22 = Kk[p2 * LD + q2];
                    double b11, b12, b21, b22;
                    if (P == Q) {
                        const double t = csm[3 * P + 2];
                        b11 = a11 - t * a12; b22 = a22 + t * a12; b12 = 0.0; b21 = 0.0;
                    } else {
                        const double cP = csm[3 * P], sP = csm[3 * P + 1], cQ = csm[3 * Q], sQ = csm[3 * Q + 1];
                        const double r11 = cP * a11 - sP * a21, r21 = sP * a11 + cP * a21;
                        const double r12 = cP * a12 - sP * a22, r22 = sP * a12 + cP * a22;
                        b11 = cQ * r11 - sQ * r12; b12 = sQ * r11 + cQ * r12;
                        b21 = cQ * r21 - sQ * r22; b22 = sQ * r21 + cQ * r22;
                    }
                    Kk[p1 * LD + q1] = b11; Kk[p1 * LD + q2] = b12;
                    Kk[p2 * LD + q1] = b21; Kk[p2 * LD + q2] = b22;
                }
                for (int e = tid; e < 2 * S * HP; e += nthr) {
                    const int m = e / (S * HP), f = e - m * S * HP, row = f / HP, P = f - row * HP;
                    double* Vk = V + m * NP * LD;
                    const double* csm = cs + m * HP * 3;
                    int p, q;
                    tournament_pair<NP>(r, P, p, q);
                    const double c = csm[3 * P], sn = csm[3 * P + 1];
                    const double vp = Vk[row * LD + p], vq = Vk[row * LD + q];
                    Vk[row * LD + p] = c * vp - sn * vq;
                    Vk[row * LD + q] = sn * vp + c * vq;
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    ND_STAMP(4);
    const bool poisoned = flags[0] != 0.0;

    // -- eigenvalues and tensor-product weights (quadratures.py:165-170)
    double* lam = Sm + L::oLam;
    double* W = Sm + L::oW;
    for (int e = tid; e < 2 * S; e += nthr) lam[(e / S) * NP + (e % S)] = K[(e / S) * NP * LD + (e % S) * (LD + 1)];
    const double* V0 = V;
    const double* V1 = V + NP * LD;
    for (int e = tid; e < S * S; e += nthr) {
        const int i0 = e / S, i1 = e - i0 * S;
        double dot = 0.0;
        for (int row = 0; row < S; ++row) dot = fma(V0[row * LD + i0], V1[row * LD + i1], dot);
        W[e] = dot * V0[i0] * V1[i1];
    }
    __syncthreads();
    ND_STAMP(5);
    return poisoned;
}

// TK = 0: operator-table transition (sde_cond_moments_tme); TK = 1: Normal closure (tme_normal / Euler--Maruyama)
template <int N, int TK>
__global__ __launch_bounds__(256, 2) void filternd_kernel(const FilterNdArgs a) {
    using L = NdTile<N>;
    constexpr int S = L::S, Z = L::Z, P = L::P, NP = L::NP, R = L::R, RW = L::RW, ZB = L::ZB;
    extern __shared__ __attribute__((aligned(16))) double Sm[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const bool scaled = a.mode == MFS_MODE_SCALED;
    double* mom = Sm + L::oMom;
    const double* coef = Sm + L::oCoef;
    const double* lp = Sm + L::oMisc;
    const int DD = a.D * a.D;

    for (int e = tid; e < kNdRows * DD; e += 256) Sm[L::oCoef + e] = a.coef[e];
    if constexpr (L::kTables) {
        constexpr int HP = L::HP;
        unsigned* kt = reinterpret_cast<unsigned*>(Sm + L::oIdxK);
        for (int e = tid; e < (NP - 1) * HP * HP; e += 256) {
            const int r = e / (HP * HP), blk = e - r * HP * HP, Pq = blk / HP, Qq = blk - Pq * HP;
            int p1, p2, q1, q2;
            tournament_pair<NP>(r, Pq, p1, p2);
            tournament_pair<NP>(r, Qq, q1, q2);
            kt[e] = (unsigned)p1 | ((unsigned)p2 << 8) | ((unsigned)q1 << 16) | ((unsigned)q2 << 24);
        }
    }
    if (tid < 4) Sm[L::oMisc + tid] = (tid < a.n_lik) ? a.lik[tid] : 0.0;
    {
        const double* src = a.m0 + (a.m0_batched ? (size_t)b * Z : 0);
        for (int e = tid; e < Z; e += 256) mom[e] = src[e];
    }
    // every thread carries an identical copy of the block-uniform state (mean, nell): they are all computed from the
    // same LDS-reduced sums, so no broadcast is ever needed
    double mean0 = 0.0, mean1 = 0.0, nell = 0.0;
    double scale0 = 1.0, scale1 = 1.0;
    if (a.mode != MFS_MODE_RAW) { const double* m = a.mean0 + (a.m0_batched ? 2 * b : 0); mean0 = m[0]; mean1 = m[1]; }
    if (scaled) { const double* m = a.scale0 + (a.m0_batched ? 2 * b : 0); scale0 = m[0]; scale1 = m[1]; }
    double* red = Sm + L::oRed;
    if (tid == 0) { red[16 * ZB] = 0.0; Sm[L::oMisc + 5] = __hiloint2double(0, -1); }   // flag slot 1: step of the first non-finite result (an int in the low word)
    __syncthreads();
    bool dead = false;
    bool warm = false;
    const double qnan = __builtin_nan("");
    const double* yrow = a.ys + (size_t)b * a.T;

    for (int t = 0; t < a.T; ++t) {
        const double y = yrow[t];
        if (!dead) {
            bool bad = false;
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
                // half 0 = prediction (filtering.py:262-266 / :330-331), half 1 = update (:268-275 / :333-339)
                const bool poisoned = quadrature_nd<N>(Sm, a.inds, a.stable, warm);
                warm = !poisoned;
                ND_STAMP_BEGIN;
                const double* lam = Sm + L::oLam;
                const double* W = Sm + L::oW;
                const double qm0 = mean0, qm1 = mean1;  // the centre this quadrature's nodes are built around
                const double qs0 = scale0, qs1 = scale1; // and their scales (1 unless scaled mode)
                // ---- pass 1: the scalar sums (conditional means, or p_y and the posterior mean)
                double s0 = 0.0, s1 = 0.0, s4 = 0.0, s2 = 0.0, s3 = 0.0;
                for (int e = tid; e < R; e += 256) {
                    const int i0 = e / S, i1 = e - i0 * S;
                    const double w = W[e];
                    const double x0 = fma(lam[i0], qs0, qm0), x1 = fma(lam[NP + i1], qs1, qm1);
                    if (half == 0) {
                        if constexpr (TK == 0) {
                            s0 = fma(w, x0 + poly2d(coef + 1 * DD, a.D, a.ext[1], x0, x1), s0);  // kappa = (1, 0): E[X'_0 | x]
                            s1 = fma(w, x1 + poly2d(coef + 0 * DD, a.D, a.ext[0], x0, x1), s1);  // kappa = (0, 1)
                        } else {
                            s0 = fma(w, poly2d(coef + 0 * DD, a.D, a.ext[0], x0, x1), s0);       // mu_0(x)
                            s1 = fma(w, poly2d(coef + 1 * DD, a.D, a.ext[1], x0, x1), s1);       // mu_1(x)
                        }
                        if (scaled) {  // scale <- sqrt(sum w var_k(x)), filtering.py:186
                            constexpr int v0 = (TK == 0) ? 14 : 2, v1 = (TK == 0) ? 15 : 4;
                            s2 = fma(w, poly2d(coef + v0 * DD, a.D, a.ext[v0], x0, x1), s2);
                            s3 = fma(w, poly2d(coef + v1 * DD, a.D, a.ext[v1], x0, x1), s3);
                        }
                    } else {
                        const double wl = w * likelihood_nd(a.lik_kind, lp, y, a.lik_component == 0 ? x0 : x1);
                        s0 = fma(wl, x0, s0); s1 = fma(wl, x1, s1); s4 += wl;
                    }
                }
                s0 = wave_sum64(s0); s1 = wave_sum64(s1); s4 = wave_sum64(s4);
                if (scaled && half == 0) { s2 = wave_sum64(s2); s3 = wave_sum64(s3); }
                if ((tid & 63) == 0) {
                    double* r3 = red + 4 * RW * (tid >> 6) + 16 * ZB;
                    r3[1] = s0; r3[2] = s1; r3[3] = s4; r3[4] = s2; r3[5] = s3;
                }
                __syncthreads();
                {
                    const double* q = red + 16 * ZB;
                    s0 = q[1] + q[4 * RW + 1] + q[8 * RW + 1] + q[12 * RW + 1];
                    s1 = q[2] + q[4 * RW + 2] + q[8 * RW + 2] + q[12 * RW + 2];
                    s4 = q[3] + q[4 * RW + 3] + q[8 * RW + 3] + q[12 * RW + 3];
                    s2 = q[4] + q[4 * RW + 4] + q[8 * RW + 4] + q[12 * RW + 4];
                    s3 = q[5] + q[4 * RW + 5] + q[8 * RW + 5] + q[12 * RW + 5];
                }
                double c0 = 0.0, c1 = 0.0, py = 1.0;
                if (half == 0) {
                    if (a.mode != MFS_MODE_RAW) { c0 = s0; c1 = s1; }
                } else {
                    py = s4;
                    if (a.mode != MFS_MODE_RAW) { c0 = s0 / py; c1 = s1 / py; }
                    nell -= fast_log(py);
                }
                ND_STAMP(6);
                // ---- pass 2: every moment about the new centre.  One node per thread per sweep of the node list; the
                //      per-node integrands are formed 16 moments at a time (compile-time multi-indices) and each batch
                //      goes through the transposing row reduction, after which every lane owns one moment summed over
                //      its 16-lane row and adds it to its slot of the [16 rows][moments] LDS table -- no per-thread
                //      accumulator array (it would spill) and no value-at-a-time wave sums.
                const int lane16 = tid & 15;
                const int cls = ((lane16 & 1) << 3) | ((lane16 & 2) << 1) | ((lane16 & 4) >> 1) | ((lane16 & 8) >> 3);
                double* myred = red + (tid >> 4) * RW + cls;
                const bool gauss_pred = (TK == 1) && (half == 0);
                if (gauss_pred) {
                    // Normal closure: per node, E[(X'_0-c_0)^a (X'_1-c_1)^b] for a + b <= 2N-1 by the Stein recursion
                    //   M(0,b) = m_1 M(0,b-1) + (b-1) S_11 M(0,b-2)
                    //   M(a,b) = m_0 M(a-1,b) + (a-1) S_00 M(a-2,b) + b S_01 M(a-1,b-1)
                    // (equal to raw_moments_mvn_kan(mu(x) - c, S(x), (a, b)), mfs/multi_dims/moments.py:110-154); three
                    // rows of the table live at a time.  Entries are emitted row by row: slot e(a, b) = a P - a(a-1)/2 + b.
                    for (int base = 0; base < R; base += 256) {
                        double wA, mA0, mA1, sA00, sA01, sA11;
                        {
                            const int eA = base + tid;
                            const bool okA = eA < R;
                            const int iA0 = okA ? eA / S : 0, iA1 = okA ? eA - iA0 * S : 0;
                            wA = okA ? W[eA] : 0.0;
                            const double xA0 = fma(lam[iA0], qs0, qm0), xA1 = fma(lam[NP + iA1], qs1, qm1);
                            mA0 = poly2d(coef + 0 * DD, a.D, a.ext[0], xA0, xA1) - c0;
                            mA1 = poly2d(coef + 1 * DD, a.D, a.ext[1], xA0, xA1) - c1;
                            sA00 = poly2d(coef + 2 * DD, a.D, a.ext[2], xA0, xA1);
                            sA01 = poly2d(coef + 3 * DD, a.D, a.ext[3], xA0, xA1);
                            sA11 = poly2d(coef + 4 * DD, a.D, a.ext[4], xA0, xA1);
                        }
                        double MA[3][P], bt[16];
                        static_for<0, P>([&](auto N0c) {
                            constexpr int n0 = N0c, r = n0 % 3, r1 = (n0 + 2) % 3, r2 = (n0 + 1) % 3;   // rows n0, n0-1, n0-2
                            static_for<0, P - n0>([&](auto N1c) {
                                constexpr int n1 = N1c;
                                double vA;
                                if constexpr (n0 == 0) {
                                    if constexpr (n1 == 0) vA = 1.0;
                                    else {
                                        vA = mA1 * MA[0][n1 > 0 ? n1 - 1 : 0];
                                        if constexpr (n1 >= 2) vA = fma((double)(n1 - 1) * sA11, MA[0][n1 - 2], vA);
                                    }
                                } else {
                                    vA = mA0 * MA[r1][n1];
                                    if constexpr (n0 >= 2) vA = fma((double)(n0 - 1) * sA00, MA[r2][n1], vA);
                                    if constexpr (n1 >= 1) vA = fma((double)n1 * sA01, MA[r1][n1 - 1], vA);
                                }
                                MA[r][n1] = vA;
                                constexpr int e = n0 * P - n0 * (n0 - 1) / 2 + n1;
                                bt[e % 16] = wA * vA;
                                if constexpr (e % 16 == 15 || e == Z - 1) {
                                    if constexpr (e % 16 != 15) static_for<e % 16 + 1, 16>([&](auto Jc) { bt[Jc] = 0.0; });
                                    const double v = row_reduce16(bt, lane16);
                                    double* slot = myred + 16 * (e / 16);
                                    *slot = (base == 0) ? v : *slot + v;
                                }
                            });
                        });
                    }
                } else if (TK == 0 && half == 0) {
                    // Operator-table prediction: f_n(x) = sum_kappa Q_kappa(x) d^kappa (x - c)^n factorises over the two
                    // coordinates,  f_n = sum_{k0 <= 4} [n0!/(n0-k0)! dx0^(n0-k0)] g_{k0}(n1),
                    //               g_{k0}(n1) = sum_{k1 <= 4-k0} Q_(k0,k1) n1!/(n1-k1)! dx1^(n1-k1)   (Q_(0,0) = 1),
                    // so a column of moments sharing n1 costs 15 FMAs once plus 5 per moment instead of 15 per moment.
                    // Entries are emitted column by column: slot e(n1, n0) = n1 P - n1(n1-1)/2 + n0.
                    for (int base = 0; base < R; base += 256) {
                        double wA, pxA0[P], pxA1[P], QA[kNdTerms + 1], bt[16];
                        {
                            const int eA = base + tid;
                            const bool okA = eA < R;
                            const int iA0 = okA ? eA / S : 0, iA1 = okA ? eA - iA0 * S : 0;
                            wA = okA ? W[eA] : 0.0;
                            const double xA0 = fma(lam[iA0], qs0, qm0), xA1 = fma(lam[NP + iA1], qs1, qm1);
                            pxA0[0] = pxA1[0] = 1.0;
#pragma unroll
                            for (int p = 1; p < P; ++p) { pxA0[p] = pxA0[p - 1] * (xA0 - c0); pxA1[p] = pxA1[p - 1] * (xA1 - c1); }
                            QA[0] = 1.0;   // kappa = (0, 0)
#pragma unroll
                            for (int k = 0; k < kNdTerms; ++k)
                                QA[k + 1] = (k < a.n_terms_used) ? poly2d(coef + k * DD, a.D, a.ext[k], xA0, xA1) : 0.0;
                        }
                        static_for<0, P>([&](auto N1c) {
                            constexpr int n1 = N1c;
                            double t1[5], g[5];
                            static_for<0, 5>([&](auto K1c) {
                                constexpr int k1 = K1c;
                                if constexpr (k1 <= n1) t1[k1] = ffact(n1, k1) * pxA1[n1 - k1];
                            });
                            static_for<0, 5>([&](auto K0c) {
                                constexpr int k0 = K0c;
                                double acc = 0.0;
                                static_for<0, 5 - k0>([&](auto K1c) {
                                    constexpr int k1 = K1c, sk = k0 + k1;
                                    if constexpr (k1 <= n1) acc = fma(QA[sk * (sk + 1) / 2 + k0], t1[k1], acc);  // (0,0) -> slot 0
                                });
                                g[k0] = acc;
                            });
                            static_for<0, P - n1>([&](auto N0c) {
                                constexpr int n0 = N0c;
                                double vA = g[0] * pxA0[n0];
                                static_for<1, 5>([&](auto K0c) {
                                    constexpr int k0 = K0c;
                                    if constexpr (k0 <= n0) vA = fma(g[k0] * ffact(n0, k0), pxA0[n0 - k0], vA);
                                });
                                constexpr int e = n1 * P - n1 * (n1 - 1) / 2 + n0;
                                bt[e % 16] = wA * vA;
                                if constexpr (e % 16 == 15 || e == Z - 1) {
                                    if constexpr (e % 16 != 15) static_for<e % 16 + 1, 16>([&](auto Jc) { bt[Jc] = 0.0; });
                                    const double v = row_reduce16(bt, lane16);
                                    double* slot = myred + 16 * (e / 16);
                                    *slot = (base == 0) ? v : *slot + v;
                                }
                            });
                        });
                    }
                } else {
                    // update (either transition kind): prod_k (x_k - c_k)^{n_k} times the likelihood, in table order; two
                    // nodes per thread share each row reduction
                    for (int base = 0; base < R; base += 512) {
                        double wA, wB, pxA0[P], pxA1[P], pxB0[P], pxB1[P], bt[16];
                        {
                            // (the thread index is laundered so that these per-thread indices are recomputed every step:
                            // hoisted out of the time loop they do not fit in the register budget and come back from
                            // scratch -- a s_waitcnt vmcnt(0) each)
                            int tv = tid;
                            asm volatile("" : "+v"(tv));
                            const int eA = base + tv, eB = base + 256 + tv;
                            const bool okA = eA < R, okB = eB < R;
                            const int iA0 = okA ? eA / S : 0, iA1 = okA ? eA - iA0 * S : 0;
                            const int iB0 = okB ? eB / S : 0, iB1 = okB ? eB - iB0 * S : 0;
                            wA = okA ? W[eA] : 0.0;
                            wB = okB ? W[eB] : 0.0;
                            const double xA0 = fma(lam[iA0], qs0, qm0), xA1 = fma(lam[NP + iA1], qs1, qm1);
                            const double xB0 = fma(lam[iB0], qs0, qm0), xB1 = fma(lam[NP + iB1], qs1, qm1);
                            pxA0[0] = pxA1[0] = pxB0[0] = pxB1[0] = 1.0;
#pragma unroll
                            for (int p = 1; p < P; ++p) {
                                pxA0[p] = pxA0[p - 1] * (xA0 - c0); pxA1[p] = pxA1[p - 1] * (xA1 - c1);
                                pxB0[p] = pxB0[p - 1] * (xB0 - c0); pxB1[p] = pxB1[p - 1] * (xB1 - c1);
                            }
                            wA *= likelihood_nd(a.lik_kind, lp, y, a.lik_component == 0 ? xA0 : xA1);
                            wB *= likelihood_nd(a.lik_kind, lp, y, a.lik_component == 0 ? xB0 : xB1);
                        }
                        static_for<0, 2 * N>([&](auto Sc) {
                            constexpr int sd = Sc;
                            static_for<0, sd + 1>([&](auto N0c) {
                                constexpr int n0 = N0c, n1 = sd - n0, zi = sd * (sd + 1) / 2 + n0;
                                bt[zi % 16] = fma(wA, pxA0[n0] * pxA1[n1], wB * (pxB0[n0] * pxB1[n1]));
                                if constexpr (zi % 16 == 15 || zi == Z - 1) {
                                    if constexpr (zi % 16 != 15) static_for<zi % 16 + 1, 16>([&](auto Jc) { bt[Jc] = 0.0; });
                                    const double v = row_reduce16(bt, lane16);
                                    double* slot = myred + 16 * (zi / 16);
                                    *slot = (base == 0) ? v : *slot + v;
                                }
                            });
                        });
                    }
                }
                __syncthreads();
                const double ipy = 1.0 / py;
                // scaled mode: the sums above are central moments about the new mean; the new scales are
                // sqrt(sum w var_k) on prediction (filtering.py:186) and the posterior standard deviations on update
                // (:195-197), and every moment is divided by prod_k scale_k^{n_k} -- the same numbers as forming
                // ((x - mean) / scale)^n per node (:177, :187, :198-200)
                double ns0 = 1.0, ns1 = 1.0;
                if (scaled) {
                    if (half == 0) { ns0 = sqrt(s2); ns1 = sqrt(s3); }
                    else {
                        double v5 = 0.0, v3 = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) { v5 += red[q * RW + 5]; v3 += red[q * RW + 3]; }
                        ns0 = sqrt(v5 * ipy);   // (2, 0)
                        ns1 = sqrt(v3 * ipy);   // (0, 2)
                    }
                }
                for (int zi = tid; zi < Z; zi += 256) {
                    int sd = 0;
                    while ((sd + 1) * (sd + 2) / 2 <= zi) ++sd;
                    const int n0 = zi - sd * (sd + 1) / 2, n1 = sd - n0;
                    // where pass 2 left this moment: predictions emit row by row (Normal closure) or column by column
                    const int e = (half != 0) ? zi : (TK == 1) ? n0 * P - n0 * (n0 - 1) / 2 + n1
                                                              : n1 * P - n1 * (n1 - 1) / 2 + n0;
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) v += red[q * RW + e];
                    v = (half == 0) ? v : v * ipy;
                    if (scaled) {
                        const double i0 = 1.0 / ns0, i1 = 1.0 / ns1;
                        double f = 1.0;
                        for (int q = 0; q < n0; ++q) f *= i0;
                        for (int q = 0; q < n1; ++q) f *= i1;
                        v *= f;
                    }
                    mom[zi] = v;
                    if (!finite(v)) red[16 * ZB] = 1.0;  // slot Z of the first row flags a non-finite moment
                }
                if (a.mode != MFS_MODE_RAW) { mean0 = c0; mean1 = c1; }
                if (scaled) { scale0 = ns0; scale1 = ns1; }
                __syncthreads();
                bad = bad || poisoned || (red[16 * ZB] != 0.0);
                __syncthreads();
                if (tid == 0) {   // (a laundered zero: the hoisted constant was living in scratch)
                    double zero = 0.0;
                    asm volatile("" : "+v"(zero));
                    red[16 * ZB] = zero;
                }
                ND_STAMP(7);
#ifdef MFS_ND_STAMPS
                if (blockIdx.x == 0 && threadIdx.x == 0) g_nd_stamps[9] += 1;
#endif
            }
            bad = bad || !finite(nell) || !finite(mean0) || !finite(mean1) || !finite(scale0) || !finite(scale1);
            if (bad) { dead = true; if (tid == 0) Sm[L::oMisc + 5] = __hiloint2double(0, t); }
        } else {
            for (int zi = tid; zi < Z; zi += 256) mom[zi] = qnan;
            mean0 = mean1 = nell = qnan;
            if (scaled) scale0 = scale1 = qnan;
        }
        __syncthreads();
        if (a.out_mom) {
            double* dst = a.out_mom + ((size_t)b * a.T + t) * Z;
            int tz = tid;                       // (laundered, as above: no per-thread pointer kept across steps)
            asm volatile("" : "+v"(tz));
            for (int zi = tz; zi < Z; zi += 256) dst[zi] = mom[zi];
        }
        if (tid == 0 && a.out_mean) {
            a.out_mean[((size_t)b * a.T + t) * 2] = mean0;
            a.out_mean[((size_t)b * a.T + t) * 2 + 1] = mean1;
        }
        if (tid == 0 && a.out_scale) {
            a.out_scale[((size_t)b * a.T + t) * 2] = scale0;
            a.out_scale[((size_t)b * a.T + t) * 2 + 1] = scale1;
        }
    }
    if (tid == 0) {
        a.out_nell[b] = nell;
        if (a.out_first_nan) a.out_first_nan[b] = __double2loint(Sm[L::oMisc + 5]);
    }
}

}  // namespace mfs
