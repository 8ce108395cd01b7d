// filternd_kernel.hpp -- hand-written HIP for gfx950: the N-D (d = 2) moment-filter time-step loop.
//
// Reference: mfs/multi_dims/filtering.py:210-344 (scan bodies :258-277, :326-341, :181-204),
// mfs/multi_dims/quadratures.py:120-178 (moment_quadrature_nd), mfs/multi_dims/moments.py:414-479 (TME transition).
//
// One filter per 256-thread workgroup (4 waves); the k-loop runs inside the kernel.
//
// The reference's rule has s^2 tensor-product nodes (lambda0_i, lambda1_j) with weights
// W_ij = v0_i[0] <v0_i, v1_j> v1_j[0] (quadratures.py:165-170), so for a separable integrand
//
//     sum_ij W_ij f(x0_i) g(x1_j) = (f(X_0) e_0)^T (g(X_1) e_0),     X_k = scale_k K_k + mean_k I,        (*)
//
// the bilinear form of two matrix functions of K_k = R^-1 H_k R^-T applied to the first unit vector.  Every sum the
// filter takes is a sum of such terms:
//   predict (operator tables, moments.py:414-479)  the integrand sum_kappa Q_kappa(x) d^kappa (x - c)^n is a polynomial:
//            (*) needs only Krylov vectors and NO eigen-decomposition -- a few powers K_k^p e_0 for the low corner of the rule's
//            moment array (the new means), then (K_k - d_k I)^p e_0 with the matrices shifted to the NEW mean, whose array
//            M[p][q] = ((K_0 - d_0)^p e_0)^T ((K_1 - d_1)^q e_0) is contracted with the coefficient blocks re-centred there.
//            Same numbers as the node sums (1e-10 in an fp64 side-by-side over 150 steps), ~2 % of their flops, and the
//            d eigensolves of quadratures.py:163 disappear from this half-step.
//   update   the likelihood is a product of factors, each a function of ONE state component (ss_models.py:63-67,
//            tests/test_filtering.py:44-46): h_k = lik_k(X_k) e_0 by a CHECKED Chebyshev interpolant of the factor on the
//            spectrum's interval (cheb_h_nd; cyclic Jacobi + spectral evaluation only when the check fails); p_y and the
//            posterior mean from h_k and K_k h_k; posterior moments M[p][q] = ((K_0 - d_0)^p h_0)^T ((K_1 - d_1)^q h_1) / p_y
//            with the matrices shifted to the posterior mean.
//   predict (Normal closures, moments.py:257-411)  E[(X' - c)^n | x] is a polynomial of degree ~ |n| deg(mu) in x, not
//            separable: it is integrated on an NCP x NCP Chebyshev grid with the weights Omega_pq = (l_p(X_0) e_0)^T (l_q(X_1) e_0)
//            of the same bilinear form (cheb_grid_rule_nd) -- exact, because NCP exceeds the integrands' degree per variable --
//            and the Stein recursion runs per grid point, 16 moments at a time through a transposing DPP row reduction.  Only
//            when NCP would not fit its table are the reference's s^2 eigen-nodes formed (both K_k diagonalised by cyclic Jacobi).
//   update, a likelihood of BOTH components (fac_component = 2; examples/2d_bearing_only.ipynb): Normal-closure kernels only --
//            sum_ij W_ij l(y, x_ij) xi_0^a xi_1^b over the reference's own s^2 eigen-nodes (filtering.py:263-275).
// Front end of every rule (front_nd): H_k = P_k G+ with G+ the Gram matrix extended by the degree-N monomials, so
// K_k = R^-1 H_k R^-T = D^-1/2 L^-1 (P_k L+) D^1/2 -- one wave eliminates the rows of G+ in registers (L D L^T, columns of L
// reaching the other lanes as DPP operands after a lane-swap duplication), then solves for the block-tridiagonal band of
// K_k only (two short unit-triangular substitutions per column).  No H gathers, no index tables.
// Matrix core: the three products of GEMM shape in a step -- the Gram product of the two Krylov families (bilinear_moments_nd)
// and, in a Normal-closure prediction, the cardinal vectors and the grid weights (cheb_grid_rule_nd) -- are 16 x 16 tiles of
// v_mfma_f64_16x16x4_f64; everything else is matrix-VECTOR work in a sequential chain (DPP multiply-adds, dpp_matvec).
#pragma once
#include "filter1d_fast.hpp"

namespace mfs {

#ifdef MFS_ND_STAMPS
// diagnostic build (tools/diag/nd_stamps.hip): cycles per phase, accumulated by thread 0 of block 0
__device__ unsigned long long g_nd_stamps[24];
__device__ unsigned long long g_nd_hist[8][40];   // [test index][-log10(off / dia), clamped]: Jacobi convergence tests of block 0
#define ND_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_nd_stamps[slot] += now_ - t_last_; t_last_ = now_; } } while (0)
#define ND_STAMP_BEGIN unsigned long long t_last_ = clock64()
#else
#define ND_STAMP(slot) do {} while (0)
#define ND_STAMP_BEGIN do {} while (0)
#endif

// Work is handed out by "thread id" = 64 x (the role of the wave) + lane; most phases of a step run on roles 0 or 0..1 only, i.e. on
// one or two of a workgroup's four waves, and two or three workgroups share a CU.  The role of a wave is NOT its position in the
// workgroup but a permutation of its SIMD id chosen by the wave slot (MFS_ND_SIMD_ROLES, default on): the waves of a workgroup sit on the four
// SIMDs in no fixed order, all on the same wave slot, and co-resident workgroups on different slots (tools/diag/hwid.hip) -- so the
// busy waves of co-resident workgroups land on DIFFERENT SIMDs instead of queueing on one.  The kernel prologue checks that the four
// roles are a permutation of 0..3 and falls back to the position in the workgroup otherwise (nd_assign_roles).  Measured, round 3:
// config 5 11.87 -> 11.36 ms, tme_normal_2 16.95 -> 16.4 ms, B = 2048 35.4 -> 34.3 ms on real batches (-9..12 % when every
// workgroup runs the same replicate in lock step).  An earlier form that re-read the role from LDS at every use gave the gain back (1 %).
#ifndef MFS_ND_SIMD_ROLES
#define MFS_ND_SIMD_ROLES 1
#endif
// (The empty volatile asm makes every call a fresh value to the optimiser.  Without it each inlined helper's thread-dependent
//  LDS addresses are loop invariants of the time loop, get hoisted out of it by the dozen, do not fit in the register budget of
//  two workgroups per CU and come back as scratch reloads -- a `s_waitcnt vmcnt(0)` in front of every phase -- where
//  recomputing them is two or three integer instructions.)
__device__ __forceinline__ int nd_tid(const int role64) {
    int t = role64 + (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(t));
    return t;
}
#ifdef MFS_ND_ROLE_DEBUG
__device__ unsigned g_nd_role_fallbacks[2];
#endif
// Prologue of the kernel: 64 x (the role of this wave), a scalar.  Two barriers; every thread of the workgroup must call it.
__device__ __forceinline__ int nd_assign_roles(double* misc) {
    const int w = threadIdx.x >> 6;
#if MFS_ND_SIMD_ROLES
    int* roles = reinterpret_cast<int*>(misc);
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // role of the wave on SIMD i of a workgroup on wave slot s: nibble i of kMap[s].  Slots 0 and 1 (two workgroups per CU): the two
    // busiest roles of the two workgroups on four different SIMDs.  Slot 2 (a third workgroup): its role 0 beside a role-1 wave,
    // its role 1 beside the other role-1 wave -- six busy waves on four SIMDs, none of the three role-0 waves beside another.
    const unsigned slot = hw & 3u, simd = (hw >> 4) & 3u;
    const unsigned map = (slot == 0u) ? 0x3210u : (slot == 1u) ? 0x1032u : (slot == 2u) ? 0x1302u : 0x0123u;
    const int cand = (int)((map >> (4u * simd)) & 3u);
    if ((threadIdx.x & 63) == 0) roles[w] = cand;
    __syncthreads();
    const unsigned seen = (1u << roles[0]) | (1u << roles[1]) | (1u << roles[2]) | (1u << roles[3]);
    __syncthreads();
#ifdef MFS_ND_ROLE_DEBUG
    if (threadIdx.x == 0) { atomicAdd(&g_nd_role_fallbacks[0], seen != 15u ? 1u : 0u); atomicAdd(&g_nd_role_fallbacks[1], cand != 0 ? 1u : 0u); }
#endif
    return __builtin_amdgcn_readfirstlane((seen == 15u) ? cand : w) * 64;
#else
    (void)misc;
    return __builtin_amdgcn_readfirstlane(w) * 64;
#endif
}
// exponents (n0, n1) of moment zi in graded lexicographic order: zi = sd (sd + 1) / 2 + n0, sd = n0 + n1  (zi < 2^20)
__device__ __forceinline__ void nd_exponents(const int zi, int& n0, int& n1) {
    int sd = (int)((__builtin_sqrtf(8.0f * (float)zi + 1.0f) - 1.0f) * 0.5f);
    sd += ((sd + 1) * (sd + 2) / 2 <= zi) ? 1 : 0;
    sd -= (sd * (sd + 1) / 2 > zi) ? 1 : 0;
    n0 = zi - sd * (sd + 1) / 2; n1 = sd - n0;
}

struct FilterNdArgs {
    int mode, T, B, stable;
    int t_begin, t_end;       // the steps this launch takes (a chunk of [0, T)); the state crosses launches through `carry`
    double* carry;            // [B][NdTile::kCarry]: moments, means / scales / nell / first-NaN / flags, eigenvector tiles (or null)
    int n_terms_used, D;      // coefficient block extent per variable (degree + 1)
    int n_factors, ny;        // likelihood = prod_f lik(kind_f, params_f, y[ycol_f], x[comp_f]);  ys is [B][T][ny]
    int fac_kind[2], fac_comp[2], fac_ycol[2];
    int coef_batched, lik_batched;
    int force_eigen;          // 1: every update diagonalises K_k (the checked fallback of the Chebyshev evaluation, as the only route)
    int joint_grid;           // 1: a likelihood of both components is integrated over the Chebyshev grid instead of the eigen-nodes (A/B)
    int ext[32];              // per-block true extents (ea | eb << 8) of the coefficient blocks; 0 = empty block
    const double* coef;       // [rows][D][D] (or [B][...]), rows = terms + 2 with terms = 14 or 27 (see nd_terms): Q_kappa in the
                              // fixed kappa order below (zeros where the model has no term), then the conditional variances
                              // of X'_0, X'_1 (scaled mode)
    const double* lik;        // [n_factors][4] (or [B][n_factors][4])
    const int32_t* inds;      // [3][s][s]
    const double* m0;         // [z] or [B][z]
    int m0_batched;
    const double* mean0;      // [2] or [B][2]
    const double* scale0;     // [2] or [B][2] (scaled mode)
    const double* ys;         // [B][T][ny]
    double* out_mom;          // [B][T][z]
    double* out_mean;         // [B][T][2]
    double* out_scale;        // [B][T][2] (scaled mode)
    double* out_nell;
    int32_t* out_first_nan;
};

// derivative multi-indices kappa, graded-lex order (the order the host fills `coef` in): 1 <= |kappa| <= 4 (TME order <= 2:
// 14 terms, TK = 0) or <= 6 (TME order 3: 27 terms, TK = 2).  The coefficient table has two more rows, the conditional
// variances of X'_0, X'_1 (scaled mode).  TK = 1 (Normal closure) uses rows 0..4 of the short layout.
constexpr int kNdTermsLo = 14, kNdTermsHi = 27;
template <int TK> constexpr int nd_terms() { return (TK == 2) ? kNdTermsHi : kNdTermsLo; }
template <int TK> constexpr int nd_rows() { return nd_terms<TK>() + 2; }
template <int TK> constexpr int nd_kmax() { return (TK == 2) ? 6 : 4; }      // highest derivative order per variable
#ifndef MFS_ND_JACOBI_TOL
#define MFS_ND_JACOBI_TOL 1e-31
#endif
// off-diagonal / diagonal mass (squared Frobenius norms) at which the Jacobi sweeps stop
constexpr double kNdJacobiTol = MFS_ND_JACOBI_TOL;
#ifndef MFS_ND_FORCE_JACOBI
#define MFS_ND_FORCE_JACOBI 0   // 1: updates always diagonalise (A/B against the Chebyshev evaluation)
#endif
#ifndef MFS_ND_FINISH_X2
#define MFS_ND_FINISH_X2 1e-14
#endif
// squared Frobenius size of the first-order eigenvector correction below which it replaces further sweeps (0: never)
constexpr double kNdFinishX2 = MFS_ND_FINISH_X2;
#ifndef MFS_ND_FINISH2_X2
#define MFS_ND_FINISH2_X2 1e-9
#endif
// ... below which the SECOND-order correction does (its error is ~ ||X||^3: 3e-14 at 1e-9; 0: never)
constexpr double kNdFinish2X2 = MFS_ND_FINISH2_X2;
template <int TK> constexpr int nd_maxd() { return (TK == 2) ? 7 : 6; }   // per-variable extent bound of the coefficient blocks
__device__ constexpr int kKap0[kNdTermsHi] = {0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5, 0, 1, 2, 3, 4, 5, 6};
__device__ constexpr int kKap1[kNdTermsHi] = {1, 0, 2, 1, 0, 3, 2, 1, 0, 4, 3, 2, 1, 0, 5, 4, 3, 2, 1, 0, 6, 5, 4, 3, 2, 1, 0};

__device__ __forceinline__ constexpr double ffact(int n, int k) {
    double r = 1.0;
    for (int j = 0; j < k; ++j) r *= (double)(n - j);
    return r;
}

// TK selects what the tile must hold: the node tables (weights, the 16-row reduction table) exist for Normal-closure
// predictions only, the operator-term tables for operator predictions only.
template <int N, int TK>
struct NdTile {
    static constexpr int S = N * (N + 1) / 2;        // Gram size
    static constexpr int Z = N * (2 * N + 1);        // number of moments, |n| <= 2N - 1
    static constexpr int P = 2 * N;                  // powers 0..2N-1 per coordinate
    static constexpr int NP = S + (S & 1);
    static constexpr int HP = NP / 2;
    static constexpr int LD = NP + 1;
    static constexpr int R = S * S;                  // tensor-product nodes
    static constexpr int kNcpMax = (N <= 6) ? 34 : 28;     // largest Chebyshev grid per variable; 34 = 3 (2 N - 1) + 1 at N = 6: TME-normal-3 of a quadratic drift
    static constexpr int oMom = 0;
    static constexpr int oK = (Z + 1) & ~1;          // [2][NP][LD]
    static constexpr int oV = oK + 2 * NP * LD;      // [2][NP][LD]
    // (Normal closures: the Chebyshev-grid rule parks its cardinal vectors [2][NCP][NP] in the tiles of K_0, K_1, V_0, V_1; at small
    //  N those are smaller than that, hence the pad)
    static constexpr int kUoPad = (TK == 1 && 2 * kNcpMax * NP > 4 * NP * LD) ? 2 * kNcpMax * NP - 4 * NP * LD : 0;
    // (TK = 0 keeps its tile under a third of a CU's LDS: the rotation records of the Jacobi fallback alias the Chebyshev scratch
    //  of the evaluation that has just been abandoned when it runs, and the eigenvalue lists exist only where a node rule is formed)
    static constexpr int oCsSeq = oV + 2 * NP * LD + kUoPad;     // [2][HP][3]
    static constexpr int nCs = (TK == 0) ? 0 : 2 * HP * 3 + 2;
    // Normal closures integrate over s^2 eigen-nodes (weights W, coordinates lam) or -- when the integrand's degree allows --
    // over an NCP x NCP Chebyshev grid with the weights Omega of the same bilinear form (kNcpMax bounds NCP)
    static constexpr int LS = (NP > kNcpMax) ? NP : kNcpMax;           // stride of the two coordinate lists
    static constexpr int nWt = (R > kNcpMax * kNcpMax) ? R : kNcpMax * kNcpMax;
    static constexpr int oW = oCsSeq + nCs;  // [S][S] / [NCP][NCP] node weights (Normal closures)
    // a likelihood of both state components is integrated over the s^2 eigen-nodes: the kernels that take one hold the
    // node weights [S][S] and the 16-row reduction table too (not the TME-order-3 tables, whose tile has no room left)
    // TK = 3: the operator-table kernel WITH those tiles (picked by the plan for a joint likelihood; TK = 0 does without them
    // and fits three workgroups per CU)
    static constexpr bool kJoint = (TK == 1) || (TK == 3 && N <= 6);     // (N = 7 with operator tables: a second workgroup per CU matters more)
    static constexpr int oLam = oW + ((TK == 1) ? ((nWt + 1) & ~1) : kJoint ? ((R + 1) & ~1) : 0); // [2][LS]
    static constexpr int ZB = (Z + 15) / 16;         // batches of 16 moments in the transposing reduction
    static constexpr int RW = 16 * ZB + 6;           // reduction row: moments (padded), flag, 5 scalar sums
    static constexpr int oRed = oLam + (kJoint ? 2 * LS : 0);       // [4 waves x 4 DPP rows][RW]
    static constexpr int nRed = kJoint ? 16 * RW : 16 * ZB + 16;   // (TK = 2: Jacobi test scratch + the flag slot at [16 ZB] only)
    static constexpr int kTerms = nd_terms<TK>(), kRows = nd_rows<TK>(), kMaxD = nd_maxd<TK>();
    static constexpr bool kOperator = (TK != 1);     // operator-table prediction (TK = 0, 2) or Normal closure (TK = 1)
    static constexpr int FFS = nd_kmax<TK>() + 1;    // falling-factorial table stride: k = 0 .. kmax
    static constexpr int oCoef = oRed + nRed;        // [kRows][D][D]
    static constexpr int oMisc = oCoef + kRows * kMaxD * kMaxD;  // lik params [4], flags [4]
    // Tournament index tables, built once per launch: which rows / columns a work item touches in round r depends on
    // (r, item) only, and recomputing it cost ~50 integer instructions per thread per round next to ~30 flops.
    //   KT[r][P * HP + Q] = p1 | p2 << 8 | q1 << 16 | q2 << 24  (u32)
    // (not for Normal closures: their Jacobi is a fallback now -- strided rounds -- and the 10 KB go to the Chebyshev-grid tables)
    // (round 3: off -- Jacobi is a rarely taken fallback on every path now, and the 10 KB of tables cost the third workgroup per CU)
    static constexpr bool kTables = false;
    static constexpr int oState = oMisc + 8;                 // block-uniform state: mean_0, mean_1, scale_0, scale_1, nell (+ 3 spare)
    static constexpr int oIdxK = oState + 8;
    static constexpr int nIdxK = kTables ? ((NP - 1) * HP * HP * 4 + 7) / 8 : 0;
    // bilinear-form path: Krylov tiles, the moment array of the rule and its shifted copy, re-centred coefficients
    static constexpr int NPW = P + kMaxD;                  // powers 0 .. 2N-1 + (D-1)
    static constexpr int MLD = NPW + 1;
    static constexpr int oPK = oIdxK + nIdxK;                // [2][NPW][NP]
    // free during a front end and during a Jacobi run, the Krylov tiles also serve as: the rows of R+ and 1 / R_jj
    // ([S + N + 1][S + 1] + [S], default front end), the Gram tile of the LDL^T completion and Jacobi's scratch ([NP][LD])
    static constexpr int oRp = oPK;
    static constexpr int oA = oPK;
    static_assert((S + N + 1) * (S + 1) + S <= 2 * NPW * NP && NP * LD <= 2 * NPW * NP, "front-end tiles alias the Krylov tiles");
    static constexpr int oM = oPK + 2 * NPW * NP;            // [NPW][MLD]
    static constexpr int oM2 = oM + NPW * MLD;               // [NPW][MLD]
    static constexpr int kUPad = (TK == 1 && 2 * kNcpMax * NP > 2 * NPW * NP + 2 * NPW * MLD) ? 2 * kNcpMax * NP - (2 * NPW * NP + 2 * NPW * MLD) : 0;
    static constexpr int oQs = oM2 + NPW * MLD + kUPad;      // [kRows][kMaxD * kMaxD] (operator path)
    static constexpr int oBin = oQs + (kOperator ? kRows * kMaxD * kMaxD : 0);  // [NPW][NPW] binomial coefficients
    static constexpr int oBx = oBin + NPW * NPW;             // 8 scalars (sums of the mean / variance rows)
    static constexpr int oLik = oBx + 8;                     // [2][4] likelihood factor parameters
    static constexpr int oPw = oLik + 8;                     // [2][NPW] powers of the centre shift
    // Chebyshev evaluation of lik(X_k) e_0: nodes cos(pi (j + 1/2) / NCH), the cosine table of the coefficient transform,
    // per-matrix coefficients and scratch ([2] x (coefficients NCH, vector ping-pong 2 NP, 8 scalars))
    static constexpr int NCH = 32;
    static constexpr int oChX = (oPw + 2 * NPW + 1) & ~1;    // [NCH]
    static constexpr int oChC = oChX + NCH;                  // [sample j][coefficient i]: cos(pi i (j + 1/2) / NCH)
    // (a 4 NCH-entry table of cos(pi m / (2 NCH)) indexed by i (2j + 1) mod 4 NCH would save 7 KB, but the per-lane addresses
    //  doubled the time of the coefficient transform -- 3.6 k -> 7.8 k cycles per step -- and a third workgroup per CU, which
    //  the 7 KB could buy, needs a 168-register build that spills 234 registers and is slower at every batch size)
    static constexpr int nChV = (2 * NP > NCH) ? 2 * NP : NCH;   // two vectors of NP, or the NCH samples
    static constexpr int oChW = oChC + NCH * NCH;            // [2][NCH + nChV + 8]
    static constexpr int nChW = NCH + nChV + 8;
    // operator terms (kappa, alpha, beta) inside the extents of their blocks, packed once per launch, and n!/(n-k)!
    static constexpr int kMaxTermWords = kTerms * kMaxD * kMaxD;       // 504 / 1323 u16
    static constexpr int oCs = (TK == 0) ? oChW : oCsSeq;
    static_assert(2 * nChW >= 2 * HP * 3 + 2 && 2 * nChW >= 2 * NP, "rotation records / spectral coefficients alias the Chebyshev scratch");
    static constexpr int oTerms = oChW + 2 * nChW;           // [1 + kMaxTermWords] u16 (count first)
    // term words: al | be << 3 | ea << 6 | eb << 9 | row << 12 -- u16 where the table has at most 16 rows, u32 for TME order 3
    using TermWord = std::conditional_t<(kTerms > 16), unsigned, unsigned short>;
    static constexpr int oFf = oTerms + (kOperator ? ((kMaxTermWords + 1) * (int)sizeof(TermWord) + 7) / 8 : 0);   // [P][FFS]
    // gather indices of G, H_0, H_1 as u16 ([3][S][S]), when they fit next to the rest at two workgroups per CU: the
    // per-rule gather then makes no global-memory round trip
    // the non-empty rows of the operator table, one packed word each (the contraction walks them: see there), count in word 31
    static constexpr int oRowTab = oFf + (kOperator ? P * FFS : 0);     // [32] u32
    static constexpr int kDoubles0 = oRowTab + (kOperator ? 16 : 0);
    // scratch of the Jacobi warm-start products / first-order finish ([S][S]): the weight tile where it exists, else the
    // moment-array tiles (free while a Jacobi runs)
    // Chebyshev-grid rule of a Normal closure: the transform table D[a][p] = (2 - [a = 0]) / NCP cos(pi a (p + 1/2) / NCP), the
    // grid's cosines, and the vectors T_a(Khat_k) e_0 -> l_p(X_k) e_0 ([2][NCP][NP], in the Krylov / moment-array tiles,
    // which a Normal-closure prediction does not otherwise use)
    static constexpr int oChD = kDoubles0;
    static constexpr int oChG = oChD + ((TK == 1) ? kNcpMax * kNcpMax : 0);
    static constexpr int kDoubles = oChG + ((TK == 1) ? kNcpMax : 0);
    static constexpr int oU = oPK;
    static_assert(TK != 1 || 2 * kNcpMax * NP <= 2 * NPW * NP + 2 * NPW * MLD + kUPad, "Chebyshev-grid vectors must fit in the Krylov and moment-array tiles");
    static constexpr int kCarry = Z + 8 + 2 * NP * LD;   // per-replicate state between the chunk launches of one run
    static constexpr int oJs = (TK == 1) ? oW : oM;
    static_assert(S * S <= 2 * NPW * MLD, "Jacobi scratch must fit in the moment-array tiles");
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

// wave-wide maximum in every lane, same DPP ladder as wave_sum64 (a shuffle-xor butterfly of doubles is two ds_bpermute
// round trips per step, ~1.5 k cycles for the two extrema of a Gershgorin interval)
__device__ __forceinline__ double wave_max64(double v) {
    v = fmax(v, dpp_move<0xB1>(v));
    v = fmax(v, dpp_move<0x4E>(v));
    v = fmax(v, dpp_move<0x141>(v));
    v = fmax(v, dpp_move<0x140>(v));
    double tot = -1.79e308;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), 16 * r);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 16 * r);
        tot = fmax(tot, __hiloint2double(hi, lo));
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

// The same sum over the whole DD x DD block (entries outside a block's true extents are zero in the table), every
// coefficient read issued before the first multiply-add: with run-time extents the nested Horner loops above are one
// dependent LDS round trip per coefficient, ~900 cycles per polynomial, and a Normal-closure prediction evaluates five
// per node.
template <int DDc>
__device__ __forceinline__ double poly2d_full(const double* __restrict__ c, const double x0, const double x1) {
    double cv[DDc * DDc];
    static_for<0, DDc * DDc>([&](auto Tc) { cv[Tc] = c[Tc]; });
    double acc = 0.0;
    static_for<0, DDc>([&](auto Ar) {
        constexpr int a = DDc - 1 - Ar;
        double row = cv[a * DDc + DDc - 1];
        static_for<0, DDc - 1>([&](auto Bc) { constexpr int b = DDc - 2 - Bc; row = fma(row, x1, cv[a * DDc + b]); });
        acc = fma(acc, x0, row);
    });
    return acc;
}
// f(integral_constant<D>) for the block extent D of this launch (1 <= D <= DMAX): one uniform branch
template <int DMAX, class F>
__device__ __forceinline__ void dispatch_extent(const int D, F&& f) {
    static_for<1, DMAX + 1>([&](auto Dc) { if (D == Dc) f(Dc); });
}

// the likelihood kinds of include/mfs_hip.h with the in-line exponential / logarithm in their scalar-constant form
// (fast_exp<true>: see there); Bernoulli-logistic is mfs/multi_dims/ss_models.py:63-67
__device__ __forceinline__ double likelihood_nd(const int kind, const double* __restrict__ lp, const double y,
                                                const double x) {
    if (kind == MFS_LIK_BERNOULLI_LOGISTIC) {
        const double z = lp[0] + x * (lp[1] + x * (lp[2] + x * lp[3]));
        const double p = rcp_sat(1.0 + fast_exp<true>(-z));
        return (y > 0.5) ? p : 1.0 - p;
    }
    if (kind == MFS_LIK_POISSON_SOFTPLUS) {
        const double rate = fast_log<true>(1.0 + fast_exp<true>(lp[0] * x));
        return fast_exp<true>(y * fast_log<true>(rate) - rate - log_factorial(y));
    }
    const double r = y - fma(lp[0], x, lp[1]);
    return fast_exp<true>(-0.5 * r * r * rcp_nr(lp[2])) * rsq_nr(6.283185307179586476925 * lp[2]);
}

// a factor of BOTH state components (fac_component = 2).  MFS_LIK_BEARING_GAUSSIAN: y ~ N(atan2(x_1, x_0), lp[0]) -- the
// bearing-only measurement of /root/reference/examples/2d_bearing_only.ipynb cell 7, norm.pdf(y, arctan2(x[1], x[0]), sd)
__device__ __forceinline__ double likelihood_joint_nd(const int kind, const double* __restrict__ lp, const double y,
                                                      const double x0, const double x1) {
    if (kind == MFS_LIK_BEARING_GAUSSIAN) {
        const double r = y - atan2(x1, x0);
        return fast_exp<true>(-0.5 * r * r * rcp_nr(lp[0])) * rsq_nr(6.283185307179586476925 * lp[0]);
    }
    return __builtin_nan("");
}

// acc[j] -= u * (v of lane j - OFF of this lane's DPP row) for j = J0 .. J1 - 1 as blocks of 8 / 4 / 2 / 1 fused DPP
// multiply-adds.  Only the FIRST block of a range carries the hazard nop (two wait states between the VALU write of the
// DPP source and its first DPP read); the later blocks read the same, by then long-written source.  The blocks have
// disjoint outputs, so a token operand chains them: without it the scheduler may hoist a nop-less block in front of
// the first one.
template <int CNT, int J0, bool NOP>
__device__ __forceinline__ void fnma_chain_block(double* a, const double u, const double v, int& tok) {
    if constexpr (CNT == 8) {
        if constexpr (NOP) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %9, -%10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %9, -%10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %9, -%10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %9, -%10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %9, -%10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %9, -%10 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %9, -%10 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %9, -%10 row_newbcast:%18 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3), "n"(J0 + 4), "n"(J0 + 5), "n"(J0 + 6), "n"(J0 + 7));
        else asm("v_fmac_f64_dpp %0, %9, -%10 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %9, -%10 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %9, -%10 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %9, -%10 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %9, -%10 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %9, -%10 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %6, %9, -%10 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %9, -%10 row_newbcast:%18 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3), "n"(J0 + 4), "n"(J0 + 5), "n"(J0 + 6), "n"(J0 + 7));
    }
    if constexpr (CNT == 4) {
        if constexpr (NOP) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %5, -%6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %5, -%6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %5, -%6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %5, -%6 row_newbcast:%10 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3));
        else asm("v_fmac_f64_dpp %0, %5, -%6 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %5, -%6 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %2, %5, -%6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %5, -%6 row_newbcast:%10 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3));
    }
    if constexpr (CNT == 2) {
        if constexpr (NOP) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %3, -%4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %3, -%4 row_newbcast:%6 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1));
        else asm("v_fmac_f64_dpp %0, %3, -%4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %3, -%4 row_newbcast:%6 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(a[1]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0), "n"(J0 + 1));
    }
    if constexpr (CNT == 1) {
        if constexpr (NOP) asm("s_nop 1\n\tv_fmac_f64_dpp %0, %2, -%3 row_newbcast:%4 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0));
        else asm("v_fmac_f64_dpp %0, %2, -%3 row_newbcast:%4 row_mask:0xf bank_mask:0xf" : "+v"(a[0]), "+v"(tok) : "v"(v), "v"(u), "n"(J0 + 0));
    }
}
template <int J0, int J1, int OFF>
__device__ __forceinline__ void fnma_chain_range(double* acc, const double u, const double v) {
    if constexpr (J0 < J1) {
        int tok = 0;
        constexpr int n8 = (J1 - J0) / 8;
        static_for<0, n8>([&](auto Pc) { fnma_chain_block<8, J0 - OFF + 8 * Pc, Pc == 0>(acc + J0 + 8 * Pc, u, v, tok); });
        constexpr int j4 = J0 + 8 * n8, rem = J1 - j4;
        if constexpr (rem >= 4) fnma_chain_block<4, j4 - OFF, j4 == J0>(acc + j4, u, v, tok);
        constexpr int j2 = j4 + ((rem >= 4) ? 4 : 0), rem2 = J1 - j2;
        if constexpr (rem2 >= 2) fnma_chain_block<2, j2 - OFF, j2 == J0>(acc + j2, u, v, tok);
        constexpr int j1 = j2 + ((rem2 >= 2) ? 2 : 0);
        if constexpr (j1 < J1) fnma_chain_block<1, j1 - OFF, j1 == J0>(acc + j1, u, v, tok);
    }
}

// total degree of the multi-index with graded-lex number j (d = 2: block m holds the m + 1 indices m (m + 1) / 2 ...)
__device__ __host__ constexpr int nd_degree(int j) {
    int m = 0;
    while ((m + 1) * (m + 2) / 2 <= j) ++m;
    return m;
}

// pair p of round r in the round-robin tournament on NP players
template <int NP>
__device__ __forceinline__ void tournament_pair(const int r, const int P, int& p, int& q) {
    if (P == 0) { p = NP - 1; q = r; }
    else { p = r + P; if (p >= NP - 1) p -= NP - 1; q = r - P; if (q < 0) q += NP - 1; }
}

// Front end of a rule (quadratures.py:151-161): G = ms[inds[0]], H_k = ms[inds[1 + k]], R = chol(G), K_k = R^-1 H_k R^-T.
// Leaves the symmetric K_0, K_1 in their LDS tiles; returns the block-uniform poison flag (a pivot that is not > 0).
// The index tables are the graded-lex ones of multi_indices.py:185-229 (the host checks that `inds` is that table), so
// the default path computes the gather arithmetically; stable = 1 (LDL^T completion) keeps the dense LDS-tile form.
// (COMPLETE: the completion branch as a compile-time switch -- as a run-time flag its selects sat on the elimination's critical
//  chain and cost the plain filter 2 % per pass.  Only one of the two instantiations runs in a launch.)
template <int N, int TK, bool COMPLETE>
__device__ __forceinline__ bool front_nd(double* __restrict__ Sm, const int32_t* __restrict__ inds, const int stable_arg, const int role64) {
    constexpr bool kComplete = COMPLETE;
    const int stable = kComplete ? stable_arg : 0;
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, LD = L::LD;
    const int tid = nd_tid(role64), nthr = blockDim.x;
    double* mom = Sm + L::oMom;
    double* A = Sm + L::oA;
    double* K = Sm + L::oK;
    double* flags = Sm + L::oMisc + 4;

    ND_STAMP_BEGIN;
    if (tid == 0) flags[0] = 0.0;
    const bool dense = stable == 2;      // (A/B: the LDS-tile form of the completion, MFS_ND_STABLE=dense)
    if (dense) {
        // -- gather (quadratures.py:151-152) for the LDL^T completion below; pad rows / columns are zero
        for (int e = tid; e < NP * NP; e += nthr) {
            const int i = e / NP, j = e - i * NP;
            const bool in = (i < S) && (j < S);
            const int o = in ? i * S + j : 0;
            const double v0 = mom[inds[o]], v1 = mom[inds[S * S + o]], v2 = mom[inds[2 * S * S + o]];
            A[i * LD + j] = in ? v0 : 0.0;
            K[i * LD + j] = in ? v1 : 0.0;
            K[NP * LD + i * LD + j] = in ? v2 : 0.0;
        }
        __syncthreads();
    }

    ND_STAMP(0);
    // -- Cholesky (quadratures.py:154) or LDL^T completion (mfs/utils.py:495-538)
    if (!dense) {
        // Front end on ONE wave, in registers.  Two facts about monomial Gram matrices carry it (the N-D form of what the
        // 1-D kernel does with its Hankel matrix):
        //   * H_k is G "shifted by e_k": H_k[i][j] = m[alpha_i + e_k + alpha_j] = G+[sigma_k(i)][j] with G+ the Gram matrix
        //     extended by the N + 1 monomials of degree N (rows only: degree <= 2N - 1 is all it needs) and sigma_k(i) the
        //     index of alpha_i + e_k.  With G = R R^T and R+ = [R; W], W R^T = the extension rows, this is H_k = P_k R+ R^T,
        //     hence   K_k = R^-1 H_k R^-T = R^-1 (P_k R+)  --  ONE triangular solve, and no gather of H_0, H_1 at all.
        //   * K_k is the matrix of "multiply by x_k" in the orthonormal polynomials, graded by degree: block tridiagonal
        //     (blocks of 1, 2, ..., N).  Outside that band the reference's dense K_k holds rounding noise only (2e-15 of
        //     max |K| on config 5; dropping it moves no moment by more than 5e-10 over 200 steps, the same as any two dense
        //     implementations differ).  Column j of K_k (degree c) then needs the rows of degree c - 1 and c only:
        //         K[c-1, j] = R[c-1,c-1]^-1 (P_k R+)[c-1, j],     K[c, j] = R[c,c]^-1 ((P_k R+)[c, j] - R[c,c-1] K[c-1, j]),
        //     two forward substitutions of at most N - 1 and N unknowns; the blocks below the diagonal are the transposes.
        // Lane r < S + N + 1 owns row r of G+.  The elimination is right-looking and square-root free (L D L^T): as soon
        // as column j is final its UNSCALED entries are spread over the four DPP rows by lane swaps while 1 / d_j forms
        // (the two chains run side by side), the lane's own factor u_rj / d_j multiplies them in fused DPP multiply-adds,
        // and 1 / sqrt(d_j) -- needed only for the rows of R+ that go to LDS -- stays off the chain.
        // stable = 1 (`ldl=True`, quadratures.py:154 -> mfs/utils.py:495-538): the SAME elimination -- `ldl` is this L D L^T --
        // and the completed factor R = L diag(f), f_j = d_j < 0 ? eps : sqrt(d_j), eps = 1e-8 ||G||_F, only changes the two
        // diagonal scalings:  K_k = R^-1 H_k R^-T = F^-1 [L^-1 (P_k L+)] (D F^-1),  i.e. row i by 1 / f_i and column j by
        // d_j / f_j (= sqrt(d_j) where the pivot is positive).  T = L^-1 (P_k L+) stays block tridiagonal: T D is the symmetric
        // matrix of <x_k pi_j, pi_i> in the L D L^T-orthogonal polynomials, zero by degree counting outside the band whatever
        // the signs of the pivots (the 1-D kernel's completed rule rests on the same fact).
        constexpr int SP = S + N + 1;                 // rows of G+
        constexpr int LDR = S + 1;
        static_assert(SP <= 64 && S <= 32, "rows of G+ in one wave, columns of L in DPP rows 0 and 1");
        double* Lp = Sm + L::oRp;                     // [SP][LDR] rows of the unit lower factor L+ (the Krylov tiles are free during a front end)
        double* sdv = Lp + SP * LDR;                  // [S] sqrt(d_j), then [S] 1 / sqrt(d_j)
        if (tid < 64) {
            const int r = (tid < SP) ? tid : SP - 1;
            int mr = 0;
            while ((mr + 1) * (mr + 2) / 2 <= r) ++mr;              // degree of alpha_r, position in its block
            const int ur = r - mr * (mr + 1) / 2;
            double v[S];
            static_for<0, S>([&](auto Jc) {                          // G+[r][j] = m[alpha_r + alpha_j], graded-lex index
                constexpr int j = Jc;
                constexpr int mj = nd_degree(j), uj = j - mj * (mj + 1) / 2;
                const int M = mr + mj;
                v[j] = mom[M * (M + 1) / 2 + ur + uj];
            });
            double eps = 0.0;
            if constexpr (kComplete) {       // eps = 1e-8 ||G||_F over the S x S Gram block (the rows of lanes < S)
                double f0 = 0.0, f1 = 0.0;
                static_for<0, S>([&](auto Jc) { if constexpr (Jc % 2 == 0) f0 = fma(v[Jc], v[Jc], f0); else f1 = fma(v[Jc], v[Jc], f1); });
                eps = 1e-8 * sqrt(wave_sum64((tid < S) ? f0 + f1 : 0.0));
            }
            __builtin_amdgcn_sched_barrier(0);
            bool bad = false;
            constexpr int S0 = (S < 16) ? S : 16;
            constexpr bool kFourRows = SP > 32;       // rows of G+ beyond lane 31 (N = 7): the copies must reach DPP rows 2, 3 too
            double dmine = 1.0;                       // d_r of this lane's own row (r < S)
            double tl[S];                             // L_rj: stored together after the loop (one address register, live once)
            static_for<0, S>([&](auto Jc) {
                constexpr int j = Jc;
                const double uj = v[j];
                const double dj = bcast<64, j>(uj);
                bad |= kComplete ? !(dj == dj) : !(dj > 0.0);      // (completion: a negative pivot is kept; a zero one gives f = 0 and a non-finite K, as upstream)
                dmine = (r == j) ? uj : dmine;
                // 1 / d_j: seed + two Newton steps, the second folded into the product (as in the 1-D kernel)
                const double y0 = __builtin_amdgcn_rcp(dj);
                const double y1 = fma(fma(-dj, y0, 1.0), y0, y0);
                const double delta = fma(-dj, y1, 1.0);
                const double ty = uj * y1;
                const double t = fma(ty, delta, ty);                   // u_rj / d_j = L_rj
                tl[j] = t;
                if constexpr (j + 1 < S) {   // the unscaled column to the DPP rows that hold rows of G+ (independent of the reciprocal)
                    double ev, od, lo0, lo1, up;
                    row_dup(uj, ev, od);
                    lo0 = ev; lo1 = od;
                    if constexpr (kFourRows) {
                        if constexpr (j + 1 < S0) half_dup(ev, lo0, up);
                        if constexpr (S > 16) half_dup(od, lo1, up);
                    }
                    if constexpr (j + 1 < S0) fnma_chain_range<j + 1, S0, 0>(v, t, lo0);
                    if constexpr (S > 16) {
                        constexpr int hi0 = (j + 1 > 16) ? j + 1 : 16;
                        fnma_chain_range<hi0, S, 16>(v, t, lo1);
                    }
                }
            });
            if (tid == 0 && bad) flags[0] = 1.0;
            if (tid < SP) static_for<0, S>([&](auto Jc) { Lp[r * LDR + Jc] = tl[Jc]; });      // (the spare lanes hold garbage)
            {   // column scale d_r / f_r and row scale 1 / f_r (sqrt(d_r), 1 / sqrt(d_r) without a completion): one per lane, in parallel
                const double rsd = rsq_nr(dmine);
                double cs = dmine * rsd, rs = rsd;
                if (kComplete && dmine < 0.0) { rs = 1.0 / eps; cs = dmine * rs; }
                if (tid < S) { sdv[tid] = cs; sdv[S + tid] = rs; }
            }
            wave_sync();
            ND_STAMP(21);
            // ---- column j of K_k = D^-1/2 T D^1/2, T = L^-1 (P_k L+): two UNIT lower forward substitutions; lane t = k S + j.
            //      Every operand sits at (a per-lane base) + (a compile-time offset) in the tile of L+; rows beyond the lane's
            //      blocks are read too (finite entries of the same tile) and their unknowns forced to zero instead.
            if (tid < 2 * S) {
                const int k = tid / S, j = tid - k * S, dk = (k == 0) ? 1 : 0;
                int c = 0;
                while ((c + 1) * (c + 2) / 2 <= j) ++c;
                const int b0 = c * (c - 1) / 2, b1 = c * (c + 1) / 2, b2 = (c + 1) * (c + 2) / 2;   // first index of degree c - 1, c, c + 1
                const int uj = j - b1;
                const double* pa = Lp + b0 * LDR + b0;            // L[c-1, c-1]
                const double* pr1 = Lp + (b1 + dk) * LDR + j;     // (P_k L+)[c-1 rows, j]
                const double* pb = Lp + b1 * LDR + b0;            // L[c, c-1]
                const double* pc = Lp + b1 * LDR + b1;            // L[c, c]
                const double* pr2 = Lp + (b2 + dk) * LDR + j;     // (P_k L+)[c rows, j]
                const double sdj = sdv[j];
                double x[(N > 1) ? N - 1 : 1], y[N];
                double rhs1[(N > 1) ? N - 1 : 1], Ra[(N > 1) ? (N - 1) * (N - 2) / 2 + 1 : 1];
                static_for<0, N - 1>([&](auto Qc) {
                    constexpr int q = Qc;
                    rhs1[q] = pr1[q * LDR];
                    static_for<0, q>([&](auto Lc) { Ra[q * (q - 1) / 2 + Lc] = pa[q * LDR + Lc]; });
                });
                double rs1[(N > 1) ? N - 1 : 1], rs2[N];
                static_for<0, N>([&](auto Qc) {
                    constexpr int q = Qc;
                    rs2[q] = sdv[S + ((q <= c) ? b1 + q : 0)];
                    if constexpr (q < N - 1) rs1[q] = sdv[S + ((q < c) ? b0 + q : 0)];
                });
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, N - 1>([&](auto Qc) {
                    constexpr int q = Qc;
                    // row sigma_k(b0 + q) = b1 + q + dk of L+ (unit lower): column j is 1 on the diagonal, 0 above it
                    const int row = b1 + q + dk;
                    double sacc = (j < row) ? rhs1[q] : ((j == row) ? 1.0 : 0.0);
                    static_for<0, q>([&](auto Lc) { sacc = fma(-Ra[q * (q - 1) / 2 + Lc], x[Lc], sacc); });
                    x[q] = (q < c) ? sacc : 0.0;
                });
                // second substitution, its operands in batches of three rows (all of them at once is 57 doubles: spills)
                constexpr int kRowBatch = 3;
                static_for<0, (N + kRowBatch - 1) / kRowBatch>([&](auto Bc) {
                    constexpr int q0 = Bc * kRowBatch, q1 = (q0 + kRowBatch < N) ? q0 + kRowBatch : N;
                    double rhs2[kRowBatch], Rb[kRowBatch][(N > 1) ? N - 1 : 1], Rc[kRowBatch][N];
                    static_for<q0, q1>([&](auto Qc) {
                        constexpr int q = Qc;
                        rhs2[q - q0] = pr2[q * LDR];
                        static_for<0, N - 1>([&](auto Lc) { Rb[q - q0][Lc] = pb[q * LDR + Lc]; });
                        static_for<0, q>([&](auto Lc) { Rc[q - q0][Lc] = pc[q * LDR + Lc]; });
                    });
                    __builtin_amdgcn_sched_barrier(0);
                    static_for<q0, q1>([&](auto Qc) {
                        constexpr int q = Qc;
                        double s0 = rhs2[q - q0], s1 = 0.0;       // (rows of degree c + 1 lie below column j: always stored entries)
                        static_for<0, N - 1>([&](auto Lc) { s1 = fma(Rb[q - q0][Lc], x[Lc], s1); });
                        static_for<0, q>([&](auto Lc) { s0 = fma(-Rc[q - q0][Lc], y[Lc], s0); });
                        y[q] = (q <= c) ? s0 - s1 : 0.0;
                    });
                });
                // K[i][j] = T[i][j] sqrt(d_j) / sqrt(d_i).  The band of K_k: rows of degree c - 1 in column j and their mirror;
                // of the diagonal block the entries on and below the diagonal, mirrored (symmetric in exact arithmetic;
                // quadratures.py:163 averages the two roundings, a difference of the size of the rounding itself)
                //  (unconditional stores: an entry outside the lane's blocks goes to the pad column of row 0, which nobody
                //   reads -- a branch per store cost more than the substitutions)
                const int kOff = L::oK + k * NP * LD, kDummy = kOff + NP;
                static_for<0, N - 1>([&](auto Qc) {
                    const bool in = Qc < c;
                    const double kv = x[Qc] * sdj * rs1[Qc];
                    Sm[in ? kOff + (b0 + Qc) * LD + j : kDummy] = kv;
                    Sm[in ? kOff + j * LD + b0 + Qc : kDummy] = kv;
                });
                static_for<0, N>([&](auto Qc) {
                    const bool in = (Qc <= c) && (Qc >= uj);
                    const double kv = y[Qc] * sdj * rs2[Qc];
                    Sm[in ? kOff + (b1 + Qc) * LD + j : kDummy] = kv;
                    Sm[in ? kOff + j * LD + b1 + Qc : kDummy] = kv;
                });
            }
            ND_STAMP(22);
        } else {
            // meanwhile: zeros outside the band (a Jacobi fallback or a Chebyshev-grid rule may have used the tiles as scratch)
            for (int e = tid - 64; e < 2 * NP * NP; e += nthr - 64) {
                const int k = e / (NP * NP), f = e - k * NP * NP, i = f / NP, j = f - i * NP;
                int di = 0, dj = 0;
                while ((di + 1) * (di + 2) / 2 <= i) ++di;
                while ((dj + 1) * (dj + 2) / 2 <= j) ++dj;
                const bool band = (i < S) && (j < S) && (di - dj <= 1) && (dj - di <= 1);
                if (!band) K[k * NP * LD + i * LD + j] = 0.0;
            }
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
    if (stable) {
        constexpr int NPAIR = S * (S - 1) / 2;
        for (int e = tid; e < 2 * NPAIR; e += nthr) {     // one (i > j) pair per item: pair f of the strict lower triangle, row by row
            double* Kk = K + (e / NPAIR) * NP * LD;
            const int f = e % NPAIR;
            int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)f)) * 0.5f);
            i += ((i + 1) * i / 2 <= f) ? 1 : 0;
            i -= (i * (i - 1) / 2 > f) ? 1 : 0;
            const int j = f - i * (i - 1) / 2;
            const double sy = 0.5 * (Kk[i * LD + j] + Kk[j * LD + i]);
            Kk[i * LD + j] = sy;
            Kk[j * LD + i] = sy;
        }
        __syncthreads();
    }

    ND_STAMP(2);
    return flags[0] != 0.0;
}

// Cyclic Jacobi with eigenvectors (quadratures.py:163) on the matrices m in [mbeg, mend) -- both K_k for a Normal-closure
// prediction, only the components a likelihood factor reads for an update.  warm_mask: bit m set = V_m holds the
// eigenvectors of an earlier rule of this filter.  On return the diagonal of K_m holds the eigenvalues, V_m the vectors.
template <int N, int TK>
__device__ void jacobi_nd(double* __restrict__ Sm, const int mbeg, const int mend, const int warm_mask, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, HP = L::HP, LD = L::LD;
    const int tid = nd_tid(role64), nthr = blockDim.x;
    double* A = Sm + L::oA;
    double* K = Sm + L::oK;
    double* V = Sm + L::oV;
    double* cs = Sm + L::oCs;
    double* flags = Sm + L::oMisc + 4;
    const int nm = mend - mbeg;
    bool warm = true;   // every matrix of the range warm-started
    for (int m = mbeg; m < mend; ++m) warm = warm && ((warm_mask >> m) & 1);
    ND_STAMP_BEGIN;
    for (int e = tid; e < nm * NP * NP; e += nthr) {
        const int m = mbeg + e / (NP * NP), f = e % (NP * NP), i = f / NP, j = f - i * NP;
        if (!((warm_mask >> m) & 1)) V[m * NP * LD + i * LD + j] = (i == j) ? 1.0 : 0.0;
    }
    // -- warm start: the eigenvector matrices of the previous rule are still in LDS.  K changes little between
    //    consecutive rules, so V_prev^T K V_prev is already nearly diagonal and the sweeps below converge
    //    quadratically from there (2-3 sweeps instead of 8-9); V then accumulates on top of V_prev.  The tile of the
    //    Cholesky factor is free by now and serves as the temporary.
    if (warm_mask & (((1 << nm) - 1) << mbeg)) {
        // all warm matrices of the range in the same two stages (the weight tile, free until the end of the rule, is the second scratch)
        double* A1 = Sm + L::oJs;   // [S][S]
        for (int e = tid; e < nm * S * S; e += nthr) {   // A_m = K_m V_m
            const int m = mbeg + e / (S * S), f = e % (S * S), i = f / S, j = f - i * S;
            if (!((warm_mask >> m) & 1)) continue;
            const double* Kk = K + m * NP * LD;
            const double* Vk = V + m * NP * LD;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < S; ++k) acc = fma(Kk[i * LD + k], Vk[k * LD + j], acc);
            if (m == 0) A[i * LD + j] = acc; else A1[i * S + j] = acc;
        }
        __syncthreads();
        constexpr int TRI = S * (S + 1) / 2;
        for (int e = tid; e < nm * TRI; e += nthr) {     // K_m = V_m^T A_m on the lower triangle, symmetrised on the fly
            const int m = mbeg + e / TRI, t = e % TRI;
            if (!((warm_mask >> m) & 1)) continue;
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
    const int kw_slot = (tid < nm * HP * HP) ? tid % (HP * HP) : 0;
    unsigned kw_next = 0;
    if constexpr (L::kTables) kw_next = reinterpret_cast<const unsigned*>(Sm + L::oIdxK)[kw_slot];
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
        // (after a warm start the first two sweeps are always needed -- the off-diagonal mass goes 1e-6 -> 1e-12 -> 1e-24
        //  of the diagonal's -- so the convergence test, a pass over both matrices and two barriers, starts at the third)
        if (!(warm && sweep < 2)) {
        // off / dia: off-diagonal and diagonal mass; xsq = sum_{i != j} (K_ij / (K_jj - K_ii))^2, the squared size of the
        // first-order eigenvector correction (below)
        double off = 0.0, dia = 0.0, xsq = 0.0;
        for (int e = tid; e < nm * S * S; e += nthr) {
            const double* Kk = K + (mbeg + e / (S * S)) * NP * LD;
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
        if (blockIdx.x == 0 && nd_tid(role64) == 0) {
            int bin = (off > 0.0 && dia > 0.0) ? (int)(-log10(off / dia)) : 39;
            bin = bin < 0 ? 0 : bin > 39 ? 39 : bin;
            g_nd_hist[sweep < 8 ? sweep : 7][bin] += 1;
            if (sweep == 2) {      // size of the first-order eigenvector correction at the first test after a warm start
                int bx2 = (xsq > 0.0) ? (int)(-log10(xsq)) : 39;
                bx2 = bx2 < 0 ? 0 : bx2 > 39 ? 39 : bx2;
                g_nd_hist[6][bx2] += 1;
            }
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
            for (int e = tid; e < nm * S * S; e += nthr) {      // X over the off-diagonal of K (the diagonal stays)
                double* Kk = K + (mbeg + e / (S * S)) * NP * LD;
                const int f = e % (S * S), i = f / S, j = f - i * S;
                if (i != j) Kk[i * LD + j] = Kk[i * LD + j] / (Kk[j * (LD + 1)] - Kk[i * (LD + 1)]);
            }
            __syncthreads();
            double* T1 = Sm + L::oJs;
            for (int e = tid; e < nm * S * S; e += nthr) {      // T_m = V_m X_m
                const int m = mbeg + e / (S * S), f = e % (S * S), k = f / S, j = f - k * S;
                const double* Kk = K + m * NP * LD;
                const double* Vk = V + m * NP * LD;
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < S; ++i) acc = fma(Vk[k * LD + i], (i == j) ? 0.0 : Kk[i * LD + j], acc);
                if (m == 0) A[k * LD + j] = acc; else T1[k * S + j] = acc;
            }
            for (int e = tid; e < nm * S; e += nthr) {          // second-order eigenvalues, parked in the rotation records
                const int m = mbeg + e / S, j = e % S;
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
            for (int e = tid; e < nm * S * S; e += nthr) {
                const int m = mbeg + e / (S * S), f = e % (S * S), k = f / S, j = f - k * S;
                V[m * NP * LD + k * LD + j] += (m == 0) ? A[k * LD + j] : T1[k * S + j];
            }
            for (int e = tid; e < nm * S; e += nthr) K[(mbeg + e / S) * NP * LD + (e % S) * (LD + 1)] = cs[e];
            break;
        }
        if (xsq <= kNdFinish2X2) {
            // The same one order further, U = I + X + Y with (Rayleigh-Schroedinger, E_jj = 0)
            //   Y_ij = (E X)_ij / (d_j - d_i)  (i != j),   Y_jj = -1/2 sum_k X_kj^2,   (E X)_ij = sum_k X_ik (d_k - d_i) X_kj,
            //   lambda_j = d_j + sum_k (d_j - d_k) X_kj^2 + sum_k X_kj (E X)_kj,
            // orthogonal and diagonalising up to ||X||^3 <= 3e-14: two S x S x S products per matrix, and it applies
            // right after the SECOND sweep of a warm-started run (||X||^2 is 1e-8 ... 1e-11 there, 1e-14 would need the
            // third sweep -- S - 1 more rounds of rotations with two barriers each).
            double* Z0 = A;                  // [S][LD-strided] (m = 0)
            double* Z1 = Sm + L::oJs;        // [S][S]          (m = 1)
            double* yd = Sm + L::oRed;       // [nm][S] diagonal of Y (the test's scratch slots are free; the flag slot lies beyond)
            static_assert(2 * S <= 16 * L::ZB, "diagonal corrections must stay below the flag slot of the reduction tile");
            for (int e = tid; e < nm * S * S; e += nthr) {      // X over the off-diagonal of K (the diagonal stays)
                double* Kk = K + (mbeg + e / (S * S)) * NP * LD;
                const int f = e % (S * S), i = f / S, j = f - i * S;
                if (i != j) Kk[i * LD + j] = Kk[i * LD + j] / (Kk[j * (LD + 1)] - Kk[i * (LD + 1)]);
            }
            __syncthreads();
            for (int e = tid; e < nm * S * S; e += nthr) {      // Z_m = E_m X_m, Y_jj
                const int m = mbeg + e / (S * S), f = e % (S * S), i = f / S, j = f - i * S;
                const double* Kk = K + m * NP * LD;
                const double di = Kk[i * (LD + 1)];
                double acc = 0.0, ysum = 0.0;
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    const double xik = (k == i) ? 0.0 : Kk[i * LD + k], xkj = (k == j) ? 0.0 : Kk[k * LD + j];
                    acc = fma(xik * (Kk[k * (LD + 1)] - di), xkj, acc);
                    ysum = fma(xkj, xkj, ysum);
                }
                if (m == 0) Z0[i * LD + j] = acc; else Z1[i * S + j] = acc;
                if (i == 0) yd[(m - mbeg) * S + j] = -0.5 * ysum;
            }
            __syncthreads();
            for (int e = tid; e < nm * S; e += nthr) {          // third-order eigenvalues, parked in the rotation records
                const int m = mbeg + e / S, j = e % S;
                const double* Kk = K + m * NP * LD;
                const double dj = Kk[j * (LD + 1)];
                double acc = dj;
                for (int k = 0; k < S; ++k) {
                    const double x = (k == j) ? 0.0 : Kk[k * LD + j];
                    const double z = (m == 0) ? Z0[k * LD + j] : Z1[k * S + j];
                    acc = fma((dj - Kk[k * (LD + 1)]) * x, x, acc);
                    acc = fma(x, z, acc);
                }
                cs[e] = acc;
            }
            __syncthreads();
            for (int e = tid; e < nm * S * S; e += nthr) {      // off-diagonal of U - I in place of X
                const int m = mbeg + e / (S * S), f = e % (S * S), i = f / S, j = f - i * S;
                double* Kk = K + m * NP * LD;
                if (i != j) {
                    const double z = (m == 0) ? Z0[i * LD + j] : Z1[i * S + j];
                    Kk[i * LD + j] += z / (Kk[j * (LD + 1)] - Kk[i * (LD + 1)]);
                }
            }
            __syncthreads();
            for (int e = tid; e < nm * S * S; e += nthr) {      // T_m = V_m (U_m - I)
                const int m = mbeg + e / (S * S), f = e % (S * S), k = f / S, j = f - k * S;
                const double* Kk = K + m * NP * LD;
                const double* Vk = V + m * NP * LD;
                const double yjj = yd[(m - mbeg) * S + j];
                double acc = 0.0;
#pragma unroll
                for (int i = 0; i < S; ++i) acc = fma(Vk[k * LD + i], (i == j) ? yjj : Kk[i * LD + j], acc);
                if (m == 0) Z0[k * LD + j] = acc; else Z1[k * S + j] = acc;
            }
            __syncthreads();
            for (int e = tid; e < nm * S * S; e += nthr) {
                const int m = mbeg + e / (S * S), f = e % (S * S), k = f / S, j = f - k * S;
                V[m * NP * LD + k * LD + j] += (m == 0) ? Z0[k * LD + j] : Z1[k * S + j];
            }
            for (int e = tid; e < nm * S; e += nthr) K[(mbeg + e / S) * NP * LD + (e % S) * (LD + 1)] = cs[e];
            break;
        }
        prev_off = off;
        }
#ifdef MFS_ND_STAMPS
        if (blockIdx.x == 0 && nd_tid(role64) == 0) g_nd_stamps[8] += 1;
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
                const int mB = (tid < nm * HP * HP) ? mbeg + tid / (HP * HP) : mbeg;
                const int Q = ((tid < nm * HP * HP) ? tid % (HP * HP) : 0) % HP;
                const double* Kk = K + mB * NP * LD;
                const double* Va = V + mB * NP * LD + Q * LD;
                const double* Vb = V + mB * NP * LD + (Q + HP) * LD;
                a11 = Kk[o11]; a12 = Kk[o12]; a21 = Kk[o21]; a22 = Kk[o22];
                x0 = Va[p1]; x1 = Va[p2]; y0 = Vb[p1]; y1 = Vb[p2];
            }
            if (tid < nm * HP) {
                const int m = mbeg + tid / HP, P = tid % HP;
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
            if constexpr (L::kTables) {
                // item (m, P, Q): the 2x2 block (pair P, pair Q) of K_m and the eigenvector entries of rows Q and Q + HP in
                // the two columns of pair P -- both need the rotation of pair P, so an item reads two rotations, not four
                const bool hasB = tid < nm * HP * HP;
                const int mB = hasB ? mbeg + tid / (HP * HP) : mbeg, blk = hasB ? tid % (HP * HP) : 0;
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
                for (int e = tid; e < nm * HP * HP; e += nthr) {
                    const int m = mbeg + e / (HP * HP), blk = e % (HP * HP), P = blk / HP, Q = blk - P * HP;
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
                for (int e = tid; e < nm * S * HP; e += nthr) {
                    const int m = mbeg + e / (S * HP), f = e % (S * HP), row = f / HP, P = f - row * HP;
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
}

// Eigenvalues and tensor-product weights of the s^2-node rule (quadratures.py:165-170); Normal-closure predictions only.
template <int N, int TK>
__device__ void weights_nd(double* __restrict__ Sm, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, LD = L::LD;
    const int tid = nd_tid(role64), nthr = blockDim.x;
    const double* K = Sm + L::oK;
    const double* V = Sm + L::oV;
    ND_STAMP_BEGIN;
    // -- eigenvalues and tensor-product weights (quadratures.py:165-170)
    double* lam = Sm + L::oLam;
    double* W = Sm + L::oW;
    for (int e = tid; e < 2 * S; e += nthr) lam[(e / S) * L::LS + (e % S)] = K[(e / S) * NP * LD + (e % S) * (LD + 1)];
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
}

// ---------------------------------------------------------------------------------------------------------------------
// bilinear-form path (identity (*) at the top of this file)
// ---------------------------------------------------------------------------------------------------------------------
// Row-times-vector with the vector distributed one entry per lane (lane j holds u_j, S <= 32: DPP rows 0 and 1) and row i
// of the matrix in the registers of lane i.  The vector reaches the other lanes as the DPP operand of the multiply-add
// itself: v_permlane16_swap on (u, copy of u) leaves row 0's entries in both rows of one register and row 1's in both rows
// of another, then entry j is `row_newbcast:j mod 16` of the one or the other -- S fused instructions and one lane swap.
// (Measured per product at S = 21, cycles: v_readlane pairs as the compiler schedules them, through one scalar pair, 660;
// the vector re-read from LDS by every lane 600; readlanes batched into distinct scalar registers 445; this form 250.)
template <int J0, int J1, int J2, bool NOP>
__device__ __forceinline__ void fma_bcast3(double& a0, double& a1, double& a2, const double s0, const double s1,
                                           const double s2, const double k0, const double k1, const double k2) {
    // (the blocks of one product are chained through the three accumulators, so only the first needs the hazard nop
    //  after the lane swap that wrote the DPP sources)
    if constexpr (NOP)
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %3, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %5, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(s0), "v"(s1), "v"(s2), "v"(k0), "v"(k1), "v"(k2), "n"(J0), "n"(J1), "n"(J2));
    else
        asm("v_fmac_f64_dpp %0, %3, %6 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %1, %4, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
            "v_fmac_f64_dpp %2, %5, %8 row_newbcast:%11 row_mask:0xf bank_mask:0xf"
            : "+v"(a0), "+v"(a1), "+v"(a2)
            : "v"(s0), "v"(s1), "v"(s2), "v"(k0), "v"(k1), "v"(k2), "n"(J0), "n"(J1), "n"(J2));
}
template <int J0>
__device__ __forceinline__ void fma_bcast1(double& a0, const double s0, const double k0) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(a0) : "v"(s0), "v"(k0), "n"(J0));
}
template <int S>
__device__ __forceinline__ double dpp_matvec(const double (&kr)[S], const double u) {
    static_assert(S <= 32, "the vector lives in DPP rows 0 and 1");
    double d0, d1;
    row_dup(u, d0, d1);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    static_for<0, S / 3>([&](auto Bc) {
        constexpr int j = 3 * Bc;
        fma_bcast3<j % 16, (j + 1) % 16, (j + 2) % 16, j == 0>(a0, a1, a2, (j < 16) ? d0 : d1, (j + 1 < 16) ? d0 : d1,
                                                       (j + 2 < 16) ? d0 : d1, kr[j], kr[j + 1], kr[j + 2]);
    });
    if constexpr (S % 3 >= 1) fma_bcast1<(S - S % 3) % 16>(a0, (S - S % 3 < 16) ? d0 : d1, kr[S - S % 3]);
    if constexpr (S % 3 == 2) fma_bcast1<(S - 1) % 16>(a1, (S - 1 < 16) ? d0 : d1, kr[S - 1]);
    return (a0 + a1) + a2;
}

// Krylov vectors PK[w][p] = (K_w - d_w I)^p PK[w][0], p < npow, for the matrices in wmask at once (d_w: the centre the
// moments are wanted about, in units of lambda -- powers of the shifted matrix give the moments about the NEW mean directly,
// where shifting the moment array afterwards took two more LDS-bound passes): wave w owns K_w, lane i row i of
// it in registers; a step is one lane swap and S fused DPP multiply-adds (dpp_matvec).  No block barrier inside (one
// wave per chain); the caller synchronises before and after.
template <int N, int TK>
__device__ void krylov_nd(double* __restrict__ Sm, const int npow, const int wmask, const double d0, const double d1, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, LD = L::LD, NPW = L::NPW;
    const int tid = nd_tid(role64);
    if (tid < 128 && ((wmask >> (tid >> 6)) & 1)) {
        const int w = tid >> 6, lane = tid & 63, li = (lane < S) ? lane : S - 1;
        const double* Kw = Sm + L::oK + w * NP * LD + li * LD;
        double* pk = Sm + L::oPK + w * NPW * NP;
        const double d = w ? d1 : d0;
        double kr[S];
        static_for<0, S>([&](auto Jc) { kr[Jc] = Kw[Jc]; });
        wave_sync();
        double u = pk[li];                        // the start vector, one entry per lane
        for (int p = 1; p < npow; ++p) {
            u = fma(-d, u, dpp_matvec<S>(kr, u));
            if (lane < S) pk[p * NP + lane] = u;
        }
    }
}

// M[p][q] = fac sc0^p sc1^q PK[0][p] . PK[1][q] for p, q < npow with p + q <= maxdeg (the only entries any later stage
// reads; the rest are zeroed): the moments sum_ij W_ij xi0_i^p xi1_j^q of the rule about its own centre
// (xi = x - mean = scale * lambda).
// sum_{al < EA, be < EB} q[al EB + be] Mw[al MLD + be], at most a dozen terms' reads in flight at a time
template <int EA, int EB, int MLD>
__device__ __forceinline__ double window_sum_nd(const double* __restrict__ q, const double* __restrict__ Mw) {
    constexpr int RG = (12 / EB > 0) ? 12 / EB : 1;      // block rows per batch
    double s0 = 0.0, s1 = 0.0;
    static_for<0, (EA + RG - 1) / RG>([&](auto Gc) {
        constexpr int a0 = Gc * RG, a1 = (a0 + RG < EA) ? a0 + RG : EA, CNT = (a1 - a0) * EB;
        double qv[CNT], mv[CNT];
        static_for<0, CNT>([&](auto Tc) {
            constexpr int al = a0 + Tc / EB, be = Tc % EB;
            qv[Tc] = q[al * EB + be]; mv[Tc] = Mw[al * MLD + be];
        });
        static_for<0, CNT>([&](auto Tc) {
            if constexpr (Tc % 2 == 0) s0 = fma(qv[Tc], mv[Tc], s0); else s1 = fma(qv[Tc], mv[Tc], s1);
        });
    });
    return s0 + s1;
}
__device__ __forceinline__ double ipow32(double x, int n) {    // x^n, 0 <= n < 32
    double r = 1.0;
#pragma unroll
    for (int bit = 0; bit < 5; ++bit) { if (n & 1) r *= x; x *= x; n >>= 1; }
    return r;
}
// (This Gram product of the two Krylov families is the one GEMM-shaped piece of a step -- [npow x S] . [S x npow] -- and goes
//  through the fp64 matrix core: v_mfma_f64_16x16x4_f64, one 16 x 16 tile of M per wave, ceil(S / 4) instructions, operands
//  A[i = lane & 15][k = lane >> 4] = PK[0][16 ti + i][4 s + k], B[k][j] = PK[1][16 tj + j][4 s + k], results
//  D[(lane >> 4) + 4 r][lane & 15] in register r.  As one thread per entry with 2 S LDS reads each it took 2.5 k cycles
//  per call, two calls per step.)
template <int N, int TK>
__device__ void bilinear_moments_nd(double* __restrict__ Sm, const int npow, const int maxdeg, const double sc0,
                                    const double sc1, const double fac, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, NPW = L::NPW, MLD = L::MLD, KS = (S + 3) / 4;
    static_assert(NPW <= 32, "at most 2 x 2 tiles of 16 x 16");
    typedef double d4 __attribute__((ext_vector_type(4)));
    double* M = Sm + L::oM;
    const int tid = nd_tid(role64);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    const int side = (npow + 15) >> 4;
    for (int tile = w; tile < side * side; tile += 4) {
        const int ti = tile / side, tj = tile - ti * side;
        const double* ua = Sm + L::oPK + min(16 * ti + r16, NPW - 1) * NP;
        const double* ub = Sm + L::oPK + NPW * NP + min(16 * tj + r16, NPW - 1) * NP;
        double av[KS], bv[KS];
        static_for<0, KS>([&](auto Sc) {
            constexpr int j0 = 4 * Sc;
            if constexpr (j0 + 3 < S) { av[Sc] = ua[j0 + kk]; bv[Sc] = ub[j0 + kk]; }
            else {      // the last slice runs past the vectors: a zero on one side, a valid (finite) entry on the other
                const int jc = min(j0 + kk, S - 1);
                const double x = ua[jc];
                av[Sc] = (j0 + kk < S) ? x : 0.0; bv[Sc] = ub[jc];
            }
        });
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        static_for<0, KS>([&](auto Sc) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[Sc], bv[Sc], acc, 0, 0, 0); });
        const int q = 16 * tj + r16, pb = 16 * ti + kk;
        double fq = fac, f0 = 1.0, f4 = 1.0;
        if (sc0 != 1.0 || sc1 != 1.0) { fq *= ipow32(sc1, q); f0 = ipow32(sc0, pb); f4 = (sc0 * sc0) * (sc0 * sc0); }
        fq *= f0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pp = pb + 4 * r;
            if (pp < npow && q < npow) M[pp * MLD + q] = (pp + q <= maxdeg) ? acc[r] * fq : 0.0;
            fq *= f4;
        }
    }
}

// M[a][b] <- fac * sum_{j0 <= a, j1 <= b} C(a, j0) (-d0)^(a-j0) C(b, j1) (-d1)^(b-j1) M[j0][j1] for a + b <= maxdeg: the
// same sums about the shifted centre (xi - d).  Two separable passes through M2; the powers of -d come from a small LDS
// table so that the inner loops are plain multiply-adds.
template <int N, int TK, int NJ>
__device__ void shift_moments_nd(double* __restrict__ Sm, const int nout, const int maxdeg, const double d0,
                                 const double d1, const double fac, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int NPW = L::NPW, MLD = L::MLD;
    double* M = Sm + L::oM;
    double* M2 = Sm + L::oM2;
    const double* bin = Sm + L::oBin;
    double* pw = Sm + L::oPw;      // [2][NPW]
    if (nd_tid(role64) < 2) {
        const double d = nd_tid(role64) ? -d1 : -d0;
        double v = 1.0;
        for (int k = 0; k < NPW; ++k) { pw[nd_tid(role64) * NPW + k] = v; v *= d; }
    }
    __syncthreads();
    // Fixed trip count NJ >= nout with every LDS read of an entry issued before any arithmetic (terms j > a are masked by
    // a select on the loaded value, not by a branch): a loop to the true bound a was one dependent LDS round trip per term.
    for (int e = nd_tid(role64); e < nout * nout; e += blockDim.x) {   // axis 0: M2[a][q], q <= maxdeg - a
        const int a = e / nout, q = e - a * nout;
        if (a + q > maxdeg) continue;
        double c[NJ], v[NJ];
        static_for<0, NJ>([&](auto Jc) {
            constexpr int j = Jc;
            c[j] = bin[a * NPW + j] * pw[max(a - j, 0)];
            v[j] = M[j * MLD + q];
        });
        double acc0 = 0.0, acc1 = 0.0;
        static_for<0, NJ>([&](auto Jc) {
            constexpr int j = Jc;
            const double vj = (j <= a) ? v[j] : 0.0;
            if constexpr (j % 2 == 0) acc0 = fma(c[j], vj, acc0); else acc1 = fma(c[j], vj, acc1);
        });
        M2[a * MLD + q] = acc0 + acc1;
    }
    __syncthreads();
    for (int e = nd_tid(role64); e < nout * nout; e += blockDim.x) {   // axis 1: M[a][b]
        const int a = e / nout, b = e - a * nout;
        if (a + b > maxdeg) continue;
        double c[NJ], v[NJ];
        static_for<0, NJ>([&](auto Jc) {
            constexpr int j = Jc;
            c[j] = bin[b * NPW + j] * pw[NPW + max(b - j, 0)];
            v[j] = M2[a * MLD + j];
        });
        double acc0 = 0.0, acc1 = 0.0;
        static_for<0, NJ>([&](auto Jc) {
            constexpr int j = Jc;
            const double vj = (j <= b) ? v[j] : 0.0;
            if constexpr (j % 2 == 0) acc0 = fma(c[j], vj, acc0); else acc1 = fma(c[j], vj, acc1);
        });
        M[a * MLD + b] = (acc0 + acc1) * fac;
    }
    __syncthreads();
}

// h_k = lik_k(X_k) e_0 WITHOUT an eigen-decomposition.  f(K) e_0 equals p(K) e_0 for any polynomial p that matches f on
// the spectrum; the likelihood factors are analytic, so the Chebyshev interpolant of f(lambda) = lik(y, scale lambda + mean)
// on a Gershgorin interval [lo, hi] of K converges geometrically and a degree <= 31 reaches rounding level for the
// models the reference ships (measured on config 5: the coefficients fall below 1e-15 of the largest by degree ~20).
// The coefficients' tail is CHECKED: if it has not reached 1e-14 of the largest, this returns false in the block-uniform
// flag and the caller diagonalises K_k instead (cyclic Jacobi) -- the result never silently depends on the truncation.
//   nodes      lambda_j = mid + half cos(pi (j + 1/2) / NCH)
//   c_i        = (2 - [i = 0]) / NCH sum_j f(lambda_j) cos(pi i (j + 1/2) / NCH)
//   h          = sum_i c_i T_i(Khat) e_0,   Khat = (K - mid) / half,  T_{i+1} = 2 Khat T_i - T_{i-1}
// followed in the same wave by K h.  Wave k works on matrix k (lane i = row i); for a component no factor reads h = e_0.
// Leaves PK[k][0] = h_k, PK[k][1] = K_k h_k.  No block barrier inside.
template <int N, int TK>
__device__ void cheb_h_nd(double* __restrict__ Sm, const FilterNdArgs& a, const int lik_mask,
                          const double (&yv)[MFS_ND_MAX_FACTORS], const double mean0, const double mean1,
                               const double scale0, const double scale1, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, LD = L::LD, NPW = L::NPW, NCH = L::NCH;
    const int tid = nd_tid(role64);
    if (tid >= 128) return;
    const int w = tid >> 6, lane = tid & 63, li = (lane < S) ? lane : S - 1;
    const double* Kw = Sm + L::oK + w * NP * LD + li * LD;
    double* pk = Sm + L::oPK + w * NPW * NP;
    double* wk = Sm + L::oChW + w * L::nChW;     // [0, NCH) coefficients, then two vectors of NP, then scalars
    double* cf = wk;
    double* vbuf = wk + NCH;
    double* sc = wk + NCH + L::nChV;
    double kr[S];
    static_for<0, S>([&](auto Jc) { kr[Jc] = Kw[Jc]; });
    ND_STAMP_BEGIN;
    if ((lik_mask >> w) & 1) {
        // ---- Gershgorin interval
        double dmin, dmax;
        {
            double r0 = 0.0, r1 = 0.0, r2 = 0.0;
            static_for<0, S>([&](auto Jc) {
                if constexpr (Jc % 3 == 0) r0 += fabs(kr[Jc]); else if constexpr (Jc % 3 == 1) r1 += fabs(kr[Jc]); else r2 += fabs(kr[Jc]);
            });
            const double dg = Kw[li];               // the diagonal entry of this lane's row (selecting it from the registers was 21 compare + select pairs)
            const double rad = ((r0 + r1) + r2) - fabs(dg);
            double lo = (lane < S) ? dg - rad : 1.79e308, hi = (lane < S) ? dg + rad : -1.79e308;
            dmin = -wave_max64(-lo); dmax = wave_max64(hi);
        }
        const double mid = 0.5 * (dmin + dmax);
        double half = 0.5 * (dmax - dmin);
        half = (half > 0.0) ? half : 1.0;
        const double ihalf = rcp_nr(half);        // (the interval only has to contain the spectrum; 1 ulp of its width is immaterial)
        ND_STAMP(16);
        // ---- f at the Chebyshev nodes (lanes < NCH), coefficients by the cosine table
        {
            const int j = (lane < NCH) ? lane : 0;
            const double lam = fma(half, Sm[L::oChX + j], mid);
            const double x = fma(lam, w ? scale1 : scale0, w ? mean1 : mean0);
            double f = 1.0;
            static_for<0, MFS_ND_MAX_FACTORS>([&](auto Qc) {      // (loaded by the caller)
                if (Qc < a.n_factors && a.fac_comp[Qc] == w) f *= likelihood_nd(a.fac_kind[Qc], Sm + L::oLik + 4 * Qc, yv[Qc], x);
            });
            if (lane < NCH) vbuf[lane] = f;      // (the vector buffers double as the sample buffer)
        }
        wave_sync();
        {
            const int i = (lane < NCH) ? lane : 0;
            const double* ct = Sm + L::oChC + i;         // column i of the [sample][coefficient] table: lanes read consecutive words
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int j = 0; j < NCH; j += 2) { c0 = fma(vbuf[j], ct[j * NCH], c0); c1 = fma(vbuf[j + 1], ct[(j + 1) * NCH], c1); }
            const double c = (c0 + c1) * ((i == 0) ? 1.0 / NCH : 2.0 / NCH);
            // tail and largest coefficient (wave reductions), effective degree
            double big = (lane < NCH) ? fabs(c) : 0.0, tail = (lane < NCH && lane >= NCH - 3) ? fabs(c) : 0.0;
            big = wave_max64(big); tail = wave_max64(tail);
            const bool ok = (tail <= 1e-14 * big) && finite(big);
            unsigned long long live = __ballot((lane < NCH) && (fabs(c) > 2e-15 * big));   // below: the rounding noise of the transform itself (~1e-15 of the largest)
            const int deg = ok ? (live ? 63 - __builtin_clzll(live) : 0) : -1;
            wave_sync();                          // everybody has read the samples
            if (lane < NCH) cf[lane] = c;
            if (lane == 0) { sc[0] = (double)deg; }
        }
        wave_sync();
        ND_STAMP(17);
        const int deg = (int)sc[0];
        if (deg < 0) {                            // not converged: tell the block, the caller falls back to Jacobi
            if (lane == 0) Sm[L::oMisc + 6] = 1.0;
            return;
        }
        // ---- h = sum_i c_i T_i(Khat) e_0; row li of Khat is (kr - mid delta) / half
        static_for<0, S>([&](auto Jc) { kr[Jc] = (kr[Jc] - ((Jc == li) ? mid : 0.0)) * ihalf; });
        double tprev = (li == 0) ? 1.0 : 0.0;     // T_0 e_0
        double tcur = kr[0];                      // T_1 e_0 = Khat e_0: column 0 = row 0 (symmetric), entry li
        double h = fma(cf[1], tcur, cf[0] * tprev);
        for (int i = 2; i <= deg; ++i) {
            const double tnext = fma(2.0, dpp_matvec<S>(kr, tcur), -tprev);
            h = fma(cf[i], tnext, h);
            tprev = tcur; tcur = tnext;
        }
        if (deg < 1) h = cf[0] * tprev;
        if (lane < S) pk[lane] = h;
        ND_STAMP(18);
#ifdef MFS_ND_STAMPS
        if (blockIdx.x == 0 && nd_tid(role64) == 0) g_nd_stamps[20] += deg;
#endif
        // ---- K h as well (K u = half Khat u + mid u): with h it gives p_y and the posterior mean, about which the caller
        //      then takes the powers
        {
            const double u = fma(half, dpp_matvec<S>(kr, h), mid * h);
            if (lane < S) pk[NP + lane] = u;
        }
        ND_STAMP(19);
    } else {
        if (lane < NP) pk[lane] = (lane == 0) ? 1.0 : 0.0;
        const double u = dpp_matvec<S>(kr, (lane == 0) ? 1.0 : 0.0);
        if (lane < S) pk[NP + lane] = u;
    }
}

// Chebyshev-grid form of the rule for a Normal-closure prediction.  The integrand E[(X' - c)^n | x] is a polynomial in x whose
// degree per variable is at most g (2N - 1), g = max(deg mu, ceil(deg Sigma / 2)) (each step of the Stein recursion
// multiplies by mu - c, every second one by an entry of Sigma).  A polynomial of degree < NCP per variable equals its
// interpolant on the NCP x NCP Chebyshev grid of a box that contains the spectra, so the rule's bilinear form (*) of it is
//     sum_pq Omega_pq F(x_p, x_q),     Omega_pq = (l_p(X_0) e_0)^T (l_q(X_1) e_0),   l_p the cardinal polynomials of the grid,
// EXACTLY the reference's sum over its s^2 eigen-nodes (quadratures.py:165-170) in exact arithmetic -- with no
// eigen-decomposition: l_p(X_k) e_0 = sum_a D[a][p] T_a(Khat_k) e_0 from NCP - 1 matrix-vector products per matrix.
// Leaves the grid coordinates (in units of lambda) in lam[k][p] and Omega in W[p][q]; waves 0 and 1 work, then all.
template <int N, int TK>
__device__ void cheb_grid_rule_nd(double* __restrict__ Sm, const int ncp, const int role64) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, NP = L::NP, LD = L::LD, LS = L::LS, NCM = L::kNcpMax;
    const int tid = nd_tid(role64);
    double* U = Sm + L::oU;              // [2][ncp][NP]
    const double* Dt = Sm + L::oChD;     // [a][p], row stride ncp
    if (tid < 128) {
        const int w = tid >> 6, lane = tid & 63, li = (lane < S) ? lane : S - 1;
        const double* Kw = Sm + L::oK + w * NP * LD + li * LD;
        double* Uw = U + w * ncp * NP;
        double kr[S];
        static_for<0, S>([&](auto Jc) { kr[Jc] = Kw[Jc]; });
        // Gershgorin interval of K_w
        double r0 = 0.0, r1 = 0.0, r2 = 0.0;
        static_for<0, S>([&](auto Jc) {
            if constexpr (Jc % 3 == 0) r0 += fabs(kr[Jc]); else if constexpr (Jc % 3 == 1) r1 += fabs(kr[Jc]); else r2 += fabs(kr[Jc]);
        });
        const double dg = Kw[li];
        const double rad = ((r0 + r1) + r2) - fabs(dg);
        const double lo = (lane < S) ? dg - rad : 1.79e308, hi = (lane < S) ? dg + rad : -1.79e308;
        const double dmin = -wave_max64(-lo), dmax = wave_max64(hi);
        const double mid = 0.5 * (dmin + dmax);
        double half = 0.5 * (dmax - dmin);
        half = (half > 0.0) ? half : 1.0;
        const double ihalf = rcp_nr(half);
        // T_a(Khat) e_0, Khat u = (K u - mid u) / half
        double tprev = (li == 0) ? 1.0 : 0.0;
        double tcur = (kr[0] - ((li == 0) ? mid : 0.0)) * ihalf;
        if (lane < S) { Uw[lane] = tprev; Uw[NP + lane] = tcur; }
        for (int a = 2; a < ncp; ++a) {
            const double ku = fma(-mid, tcur, dpp_matvec<S>(kr, tcur)) * ihalf;
            const double tnext = fma(2.0, ku, -tprev);
            if (lane < S) Uw[a * NP + lane] = tnext;
            tprev = tcur; tcur = tnext;
        }
        if (lane < ncp) Sm[L::oLam + w * LS + lane] = fma(half, Sm[L::oChG + lane], mid);
    }
    __syncthreads();
    // cardinal vectors u_p = sum_a D[a][p] t_a, into the tiles of K_0, K_1, V_0, V_1 (the matrices have done their work for
    // this half-step; the caller drops the warm start of a later Jacobi fallback), then the weights W[p][q] = u_p(K_0) . u_q(K_1).
    // Both are small matrix products -- [ncp x ncp]^T [ncp x S] per matrix, [ncp x S] [S x ncp] -- on the fp64 matrix core, one
    // 16 x 16 tile per wave and pass (as one thread per entry with a masked loop over kNcpMax terms: 6 k + 2.5 k cycles per
    // step).  Operands outside the true extents: addresses clamped, one side zeroed.
    typedef double d4 __attribute__((ext_vector_type(4)));
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r16 = lane & 15, kk = lane >> 4;
    double* Uo = Sm + L::oK;             // [2][ncp][NP]
    const int sp = (ncp + 15) >> 4;      // row tiles of p
    {
        constexpr int SC = (S + 15) / 16, KA = (NCM + 3) / 4;
        for (int tile = wv; tile < 2 * sp * SC; tile += 4) {
            const int w = tile / (sp * SC), rem = tile - w * sp * SC, ti = rem / SC, tj = rem - ti * SC;
            const int pp = 16 * ti + r16, r = 16 * tj + r16;
            const double* da = Dt + min(pp, ncp - 1);                       // A[i = p][k = a] = D[a][p]
            const double* tb = U + w * ncp * NP + min(r, NP - 1);           // B[k = a][j = r] = t_a[r]
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            double av[KA], bv[KA];
            static_for<0, KA>([&](auto Sc) {
                const int ai = 4 * Sc + kk, ac = min(ai, ncp - 1);
                const double x = da[ac * ncp];
                av[Sc] = (ai < ncp) ? x : 0.0; bv[Sc] = tb[ac * NP];
            });
            static_for<0, KA>([&](auto Sc) {
                if (4 * Sc < ncp) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[Sc], bv[Sc], acc, 0, 0, 0);
            });
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int po = 16 * ti + kk + 4 * q;
                if (po < ncp && r < S) Uo[(w * ncp + po) * NP + r] = acc[q];
            }
        }
    }
    __syncthreads();
    double* W = Sm + L::oW;
    {
        constexpr int KS = (S + 3) / 4;
        for (int tile = wv; tile < sp * sp; tile += 4) {
            const int ti = tile / sp, tj = tile - ti * sp;
            const double* ua = Uo + min(16 * ti + r16, ncp - 1) * NP;
            const double* ub = Uo + ncp * NP + min(16 * tj + r16, ncp - 1) * NP;
            double av[KS], bv[KS];
            static_for<0, KS>([&](auto Sc) {
                constexpr int j0 = 4 * Sc;
                if constexpr (j0 + 3 < S) { av[Sc] = ua[j0 + kk]; bv[Sc] = ub[j0 + kk]; }
                else {
                    const int jc = min(j0 + kk, S - 1);
                    const double x = ua[jc];
                    av[Sc] = (j0 + kk < S) ? x : 0.0; bv[Sc] = ub[jc];
                }
            });
            d4 acc = {0.0, 0.0, 0.0, 0.0};
            static_for<0, KS>([&](auto Sc) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[Sc], bv[Sc], acc, 0, 0, 0); });
            const int q = 16 * tj + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pp = 16 * ti + kk + 4 * r;
                if (pp < ncp && q < ncp) W[pp * ncp + q] = acc[r];
            }
        }
    }
    __syncthreads();
}

// TK = 0: operator-table transition (sde_cond_moments_tme); TK = 1: Normal closure (tme_normal / Euler--Maruyama)
// workgroups per CU the register budget is sized for.  Two: a 168-register build (three) spills 234 registers and measured
// slower at B = 512 (4.1 vs 3.0 ms per 100 steps) AND at B = 2048 (14.2 vs 11.7); MFS_ND_OCC overrides it for A/B runs.
#ifndef MFS_ND_OCC
#define MFS_ND_OCC 2
#endif
#define ND_TID nd_tid(role64)
// ... and three where the tile is under a third of the CU's 160 KB (TK = 0, N <= 6): without MachineLICM (Makefile) the
// 168-register build has ONE spilled register, where it had 234
#ifndef MFS_ND_OCC3
#define MFS_ND_OCC3 1
#endif
template <int N, int TK> constexpr int nd_occ() {
    return (MFS_ND_OCC3 && MFS_ND_OCC == 2 && TK == 0 && 3 * NdTile<N, TK>::kDoubles * 8 <= 160 * 1024) ? 3 : MFS_ND_OCC;
}
template <int N, int TK>
__global__ __launch_bounds__(256, (nd_occ<N, TK>())) void filternd_kernel(const FilterNdArgs a) {
    using L = NdTile<N, TK>;
    constexpr int S = L::S, Z = L::Z, P = L::P, NP = L::NP, LD = L::LD, R = L::R, RW = L::RW, ZB = L::ZB;
    constexpr int NPW = L::NPW, MLD = L::MLD;
    extern __shared__ __attribute__((aligned(16))) double Sm[];
    const int role64 = nd_assign_roles(Sm + L::oMisc);      // 64 x (the role of this wave): a scalar, handed to every helper
    const int tid = nd_tid(role64), b = blockIdx.x;
    const bool scaled = a.mode == MFS_MODE_SCALED;
    const bool raw = a.mode == MFS_MODE_RAW;
    double* mom = Sm + L::oMom;
    const double* coef = Sm + L::oCoef;
    double* M = Sm + L::oM;
    const int DD = a.D * a.D;

    {
        const double* src = a.coef + (a.coef_batched ? (size_t)b * L::kRows * DD : 0);
        for (int e = tid; e < L::kRows * DD; e += 256) Sm[L::oCoef + e] = src[e];
    }
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
    for (int e = tid; e < NPW * NPW; e += 256) {   // binomial coefficients C(i, j), exact in fp64 for i < 2N + 6 <= 20
        const int i = e / NPW, j = e - i * NPW;
        double c = (j <= i) ? 1.0 : 0.0;
        for (int k = 1; k <= j && j <= i; ++k) c = c * (double)(i - j + k) / (double)k;
        Sm[L::oBin + e] = c;
    }
    for (int e = tid; e < L::NCH * (L::NCH + 1); e += 256) {   // Chebyshev nodes and the cosine table of the coefficient transform
        constexpr int NCH = L::NCH;
        if (e < NCH) Sm[L::oChX + e] = cospi(((double)e + 0.5) / NCH);
        else { const int j = (e - NCH) / NCH, i = (e - NCH) % NCH; Sm[L::oChC + j * NCH + i] = cospi((double)i * ((double)j + 0.5) / NCH); }   // [sample j][coefficient i]
    }
    if constexpr (L::kOperator) {   // term list, u16 words: al | be << 3 | ea << 6 | eb << 9 | row << 12 (the re-centring pass decodes them;
                                    // the contraction walks the rows, below); word 0 = the count
        using TermWord = typename L::TermWord;
        TermWord* tw = reinterpret_cast<TermWord*>(Sm + L::oTerms);
        if (tid == 0) {
            unsigned n = 0;
            for (int k = 0; k < a.n_terms_used && k < L::kTerms; ++k) {
                const int ea = a.ext[k] & 0xff, eb = a.ext[k] >> 8;
                for (int al = 0; al < ea; ++al)
                    for (int be = 0; be < eb; ++be)
                        tw[1 + n++] = (TermWord)((unsigned)al | ((unsigned)be << 3) | ((unsigned)ea << 6) | ((unsigned)eb << 9) | ((unsigned)k << 12));
            }
            tw[0] = (TermWord)n;
            // row words: k0 | k1 << 3 | ea << 6 | eb << 9 | (index of the row's first term) << 12 | (wave pair that takes it) << 22
            unsigned* rw = reinterpret_cast<unsigned*>(Sm + L::oRowTab);
            unsigned nr = 0, first = 0;
            for (int k = 0; k < a.n_terms_used && k < L::kTerms; ++k) {
                const unsigned ea = a.ext[k] & 0xff, eb = a.ext[k] >> 8;
                if (ea * eb == 0) continue;
                rw[nr++] = (unsigned)kKap0[k] | ((unsigned)kKap1[k] << 3) | (ea << 6) | (eb << 9) | (first << 12) | ((2 * first >= n ? 1u : 0u) << 22);
                first += ea * eb;
            }
            for (unsigned q = nr; q < 31; ++q) rw[q] = 0;
            rw[31] = nr;
        }
        for (int e = tid; e < P * L::FFS; e += 256) Sm[L::oFf + e] = ffact(e / L::FFS, e % L::FFS) * ((e % L::FFS <= e / L::FFS) ? 1.0 : 0.0);
    }
    if (tid < 8) {
        const double* src = a.lik + (a.lik_batched ? (size_t)b * a.n_factors * 4 : 0);
        Sm[L::oLik + tid] = (tid < a.n_factors * 4) ? src[tid] : 0.0;
    }
    // a later chunk of a run resumes from the carry: moments, scalars and the eigenvector tiles a warm start reads
    const bool resume = (a.t_begin > 0) && (a.carry != nullptr);
    double* cw = a.carry ? a.carry + (size_t)b * L::kCarry : nullptr;
    {
        const double* src = resume ? cw : a.m0 + (a.m0_batched ? (size_t)b * Z : 0);
        for (int e = tid; e < Z; e += 256) mom[e] = src[e];
        if (resume) for (int e = tid; e < 2 * NP * LD; e += 256) Sm[L::oV + e] = cw[Z + 8 + e];
    }
    // The block-uniform state (mean, scale, nell) lives in LDS: st[0..4].  Every thread computes the new values identically from
    // LDS data (no broadcast is needed), thread 0 writes them back at the end of a half-step, and each half-step reads them
    // AFTER its front end -- as per-thread variables live across the whole time loop they were spilled to scratch around the
    // register-hungry phases and came back through `s_waitcnt vmcnt(0)` reloads in two dozen places per step.
    double* st = Sm + L::oState;
    if (tid == 0) {
        double m0v = 0.0, m1v = 0.0, s0v = 1.0, s1v = 1.0, nl = 0.0;
        if (resume) {
            m0v = cw[Z]; m1v = cw[Z + 1]; s0v = cw[Z + 2]; s1v = cw[Z + 3]; nl = cw[Z + 4];
        } else {
            if (!raw) { const double* m = a.mean0 + (a.m0_batched ? 2 * b : 0); m0v = m[0]; m1v = m[1]; }
            if (scaled) { const double* m = a.scale0 + (a.m0_batched ? 2 * b : 0); s0v = m[0]; s1v = m[1]; }
        }
        st[0] = m0v; st[1] = m1v; st[2] = s0v; st[3] = s1v; st[4] = nl;
    }
    double* red = Sm + L::oRed;
    if (tid == 0) { red[16 * ZB] = 0.0; Sm[L::oMisc + 5] = resume ? cw[Z + 5] : __hiloint2double(0, -1); }   // flag slot 1: step of the first non-finite result (an int in the low word)
    // which matrices an update diagonalises: the components a likelihood factor reads
    // (component 2 = a factor of BOTH components, e.g. a bearing measurement: no bilinear form of matrix functions exists for
    //  it; Normal-closure kernels integrate it over their node set -- Chebyshev grid or eigen-nodes -- like the prediction)
    const bool joint = L::kJoint && a.n_factors == 1 && a.fac_comp[0] == 2;
    int lik_mask = 0;
    for (int f = 0; f < a.n_factors; ++f) lik_mask |= (a.fac_comp[f] < 2) ? (1 << a.fac_comp[f]) : 0;
    const int ubeg = (lik_mask & 1) ? 0 : 1, uend = (lik_mask & 2) ? 2 : 1;
    // Normal closure: the size of the Chebyshev grid that integrates every moment's integrand exactly (cheb_grid_rule_nd), from
    // the per-variable degrees of mu (rows 0, 1) and Sigma (rows 2 .. 4); 0 = too large for the tables, eigen-nodes instead
    int ncp = 0;
    if constexpr (TK == 1) {
        int dmu = 0, dsg = 0;
        for (int r = 0; r < 5; ++r) {
            const int ex = a.ext[r], dg = max(ex & 0xff, ex >> 8) - 1;
            if (r < 2) dmu = max(dmu, dg); else dsg = max(dsg, dg);
        }
        const int g = max(1, max(dmu, (dsg + 1) / 2));
        ncp = g * (P - 1) + 1;
        if (joint && a.joint_grid) ncp = L::kNcpMax;     // (A/B: a joint likelihood on the same grid -- take all the tables hold)
        if (ncp > L::kNcpMax || a.force_eigen) ncp = 0;   // (table size)
        if (ncp > 0) {
            for (int e = tid; e < ncp * ncp; e += 256) {
                const int aa = e / ncp, pp = e - aa * ncp;
                Sm[L::oChD + e] = ((aa == 0) ? 1.0 : 2.0) / (double)ncp * cospi((double)aa * ((double)pp + 0.5) / (double)ncp);
            }
            if (tid < ncp) Sm[L::oChG + tid] = cospi(((double)tid + 0.5) / (double)ncp);
        }
    }
    __syncthreads();
    bool dead = resume ? (cw[Z + 6] != 0.0) : false;
    int warm_mask = resume ? (int)cw[Z + 7] : 0;
    const double qnan = __builtin_nan("");
    const double* yrow = a.ys + (size_t)b * a.T * a.ny;
    double* bx = Sm + L::oBx;
    double* qs = Sm + L::oQs;

    for (int t = a.t_begin; t < a.t_end; ++t) {
        if (!dead) {
            bool bad = false;

            // The two half-steps share ONE call site of the front end (inlined there: as a called function it saved and
            // restored two dozen callee-saved registers through scratch on every call, and twice inlined it doubles the
            // largest piece of straight-line code in the kernel).
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
            // (ND_TID: a fresh thread id at every use -- even one id per half-step was spilled and reloaded from scratch in four dozen places)
            const bool poisoned = a.stable ? front_nd<N, TK, true>(Sm, a.inds, a.stable, role64) : front_nd<N, TK, false>(Sm, a.inds, 0, role64);
            bad = bad || poisoned;
            const double mean0 = st[0], mean1 = st[1], scale0 = st[2], scale1 = st[3];
            // =========================================================================================================
            // prediction (filtering.py:262-266 / :330-331 / :183-190)
            // =========================================================================================================
            if (half == 0) {
                double c0 = 0.0, c1 = 0.0, ns0 = 1.0, ns1 = 1.0;
                if constexpr (L::kOperator) {
                    ND_STAMP_BEGIN;
                    // ---- the low corner of the rule's moment array about its own centre: all that the new means (and
                    //      scales) need -- a few Krylov steps from e_0
                    const int npow = P + a.D - 1;
                    int nlow = 2;     // extents of the blocks that are summed over the rule: mean rows, variance rows
                    for (int r4 = 0; r4 < (scaled ? 4 : 2); ++r4) {
                        const int row = (r4 < 2) ? r4 : L::kTerms - 2 + r4;
                        nlow = max(nlow, max(a.ext[row] & 0xff, a.ext[row] >> 8));
                    }
                    if (ND_TID < 2 * NP) Sm[L::oPK + (ND_TID / NP) * NPW * NP + (ND_TID % NP)] = (ND_TID % NP == 0) ? 1.0 : 0.0;
                    const int maxdeg = P - 1 + 2 * (a.D - 1);     // highest total degree a re-centred block reaches from a moment
                    if (2 * (nlow - 1) <= P - 1) {
                        // ... which the rule reproduces: its moments of total degree <= 2N - 1 about its own centre ARE the
                        // moments it was built from (quadratures.py:120-178; tests/test_multi_dim_quadrature.py:90-98 hold that
                        // to 1e-12), so the corner is read off the moment vector -- no Krylov steps, no dot products
                        for (int e = ND_TID; e < nlow * nlow; e += 256) {
                            const int pp = e / nlow, q = e - pp * nlow, dg = pp + q;
                            double v = 0.0;
                            if (dg <= 2 * (nlow - 1)) {
                                v = mom[dg * (dg + 1) / 2 + pp];
                                for (int k2 = 0; k2 < pp; ++k2) v *= scale0;
                                for (int k2 = 0; k2 < q; ++k2) v *= scale1;
                            }
                            M[pp * MLD + q] = v;
                        }
                        __syncthreads();
                        ND_STAMP(10);
                    } else {     // (orders too low for that: a few Krylov steps from e_0)
                        __syncthreads();
                        krylov_nd<N, TK>(Sm, nlow, 3, 0.0, 0.0, role64);
                        __syncthreads();
                        ND_STAMP(10);
                        bilinear_moments_nd<N, TK>(Sm, nlow, 2 * (nlow - 1), scale0, scale1, 1.0, role64);
                        __syncthreads();
                    }
                    ND_STAMP(11);
                    // ---- sums of the mean rows (and, scaled mode, the variance rows) of the table over the rule: each is a
                    //      polynomial in x = xi + mean, so sum W Q(x) = sum_{al,be} Q'[al][be] M[al][be] with Q' the block
                    //      re-centred at the rule's centre.  One wave per row, one (i, j, al, be) product per lane and pass, all
                    //      LDS reads independent, then a wave sum (the two nested per-thread loops this replaces were
                    //      5 k cycles of dependent LDS round trips per step).
                    {
                        const int r4 = ND_TID >> 6, lane = ND_TID & 63;
                        const int row = (r4 == 0) ? 1 : (r4 == 1) ? 0 : L::kTerms - 2 + r4;   // kappa (1,0) is row 1, (0,1) row 0; the last two rows are the variances
                        double acc = 0.0;
                        if (r4 < 2 || scaled) {
                            const int ea = a.ext[row] & 0xff, eb = a.ext[row] >> 8;
                            const int ta = ea * (ea + 1) / 2, tb = eb * (eb + 1) / 2;
                            const double* bin = Sm + L::oBin;
                            for (int e = lane; e < ta * tb; e += 64) {
                                const int pa = e / tb, pb = e - pa * tb;
                                int i = 0, j = 0;
                                while ((i + 1) * (i + 2) / 2 <= pa) ++i;
                                while ((j + 1) * (j + 2) / 2 <= pb) ++j;
                                const int al = pa - i * (i + 1) / 2, be = pb - j * (j + 1) / 2;
                                const double cv = coef[row * DD + i * a.D + j], b0 = bin[i * NPW + al], b1 = bin[j * NPW + be];
                                const double mv = M[al * MLD + be];
                                double pw = 1.0;
                                for (int q = 0; q < i - al; ++q) pw *= mean0;
                                for (int q = 0; q < j - be; ++q) pw *= mean1;
                                acc = fma(cv * b0 * b1 * pw, mv, acc);
                            }
                        }
                        acc = wave_sum64(acc);
                        if (lane == 0) bx[r4] = acc;
                    }
                    __syncthreads();
                    // E[X'_k | x] = x_k + Q_{e_k}(x): the mean of the prediction is the rule's own first moment plus the sum
                    if (!raw) { c0 = mean0 + M[1 * MLD + 0] + bx[0]; c1 = mean1 + M[0 * MLD + 1] + bx[1]; }
                    if (scaled) { ns0 = sqrt(bx[2]); ns1 = sqrt(bx[3]); }
                    ND_STAMP(12);
                    // ---- waves 0, 1: Krylov vectors of the matrices shifted to the NEW mean; meanwhile waves 2, 3: the coefficient
                    //      blocks re-centred there, Q_kappa(c + eta) in powers of eta, one packed term (kappa, alpha, beta) per
                    //      thread and pass
                    __syncthreads();     // (everybody has read bx and the low corner of M)
                    if (ND_TID < 128) {
                        krylov_nd<N, TK>(Sm, npow, 3, (c0 - mean0) / scale0, (c1 - mean1) / scale1, role64);
                    } else {
                        const typename L::TermWord* tw = reinterpret_cast<const typename L::TermWord*>(Sm + L::oTerms);
                        const int nt = (int)tw[0];
                        const double* bin = Sm + L::oBin;
                        for (int e = ND_TID - 128; e < nt; e += 128) {
                            const unsigned w = tw[1 + e];
                            const int al = w & 7, be = (w >> 3) & 7, ea = (w >> 6) & 7, eb = (w >> 9) & 7, row = w >> 12;
                            const double* blk = coef + row * DD;
                            double acc = 0.0, pi = 1.0;
                            for (int i = al; i < ea; ++i) {
                                double pj = 1.0, rsum = 0.0;
                                for (int j = be; j < eb; ++j) { rsum = fma(bin[j * NPW + be] * pj, blk[i * a.D + j], rsum); pj *= c1; }
                                acc = fma(bin[i * NPW + al] * pi, rsum, acc);
                                pi *= c0;
                            }
                            qs[e] = acc;        // (term order: the contraction walks the list front to back)
                        }
                    }
                    __syncthreads();
                    ND_STAMP(13);
                    // ---- the moment array about the new mean
                    bilinear_moments_nd<N, TK>(Sm, npow, maxdeg, scale0, scale1, 1.0, role64);
                    __syncthreads();
                    ND_STAMP(14);
                    // ---- contraction: E_n = M[n] + sum_kappa n!/(n-kappa)! sum_{al,be} Q'_kappa[al][be] M[n - kappa + (al, be)].
                    //      Lane = moment n (two waves cover them), the term list split between the two wave pairs at a row
                    //      boundary: the walk over (kappa, al, be) is then the same for every lane of a wave -- the row's
                    //      extents, the term index and the Q' address are scalar, the falling factorials are read once per
                    //      row and the window of M is one base address per row plus immediate offsets.  Per term that leaves
                    //      two LDS reads and one multiply-add (as one (kappa, al, be) term at a time per thread, each decoded from
                    //      its packed word -- about 25 instructions per term -- this was 7.2 k cycles per step).
                    {
                        double* part = Sm + L::oM2;      // [2][Z]
                        static_assert(2 * Z <= NPW * MLD && Z <= 128, "partial sums must fit in the M2 tile");
                        const double* ff = Sm + L::oFf;
                        int zn0, zn1; nd_exponents(min(ND_TID & 127, Z - 1), zn0, zn1);    // the moment this thread owns (ND_TID & 127 < Z)
                        const int pair = __builtin_amdgcn_readfirstlane(ND_TID >> 7);
                        static_assert(L::kTerms < 31, "row words live in the lanes of one register");
                        const unsigned myrow = reinterpret_cast<const unsigned*>(Sm + L::oRowTab)[ND_TID & 31];
                        const int nrows = __builtin_amdgcn_readlane(myrow, 31);
                        const double* f0 = ff + zn0 * L::FFS;
                        const double* f1 = ff + zn1 * L::FFS;
                        double v = 0.0;
                        for (int k = 0; k < nrows; ++k) {
                            const unsigned rwd = __builtin_amdgcn_readlane(myrow, k);
                            if ((int)(rwd >> 22) != pair) continue;
                            const int k0 = rwd & 7, k1 = (rwd >> 3) & 7, ea = (rwd >> 6) & 7, eb = (rwd >> 9) & 7, first = (rwd >> 12) & 1023;
                            const double fa = f0[k0], fb = f1[k1];        // (zero where kappa exceeds n: the window's base is clamped, not branched on)
                            const double* Mw = M + max(zn0 - k0, 0) * MLD + max(zn1 - k1, 0);
                            const double* q = qs + first;
                            // the block's extents select a fully unrolled window sum (compile-time offsets, every LDS read of
                            // the window in flight before the first multiply-add: with run-time trip counts each term was
                            // its own read - wait - multiply-add round trip)
                            double in = 0.0;
                            static_for<1, L::kMaxD + 1>([&](auto Ea) {
                                if (ea == Ea) static_for<1, L::kMaxD + 1>([&](auto Eb) {
                                    if (eb == Eb) in = window_sum_nd<Ea, Eb, MLD>(q, Mw);
                                });
                            });
                            v = fma(fa * fb, in, v);
                        }
                        if ((ND_TID & 127) < Z) part[pair * Z + (ND_TID & 127)] = v;
                        __syncthreads();
                        if (ND_TID < Z) {
                            double v = (part[ND_TID] + part[Z + ND_TID]) + M[zn0 * MLD + zn1];
                            if (scaled) v *= ipow32(1.0 / ns0, zn0) * ipow32(1.0 / ns1, zn1);
                            mom[ND_TID] = v;
                            if (!finite(v)) red[16 * ZB] = 1.0;
                        }
                    }
                    ND_STAMP(6);
                } else {
                    // ---- Normal closure: the s^2 nodes explicitly.  Both K_k diagonalised, weights, then per node the
                    //      Stein recursion (equal to raw_moments_mvn_kan(mu(x) - c, S(x), (a, b)), moments.py:110-154)
                    //      -- or, when the degree of the integrands allows, the Chebyshev grid with the weights of the same
                    //      bilinear form (cheb_grid_rule_nd: no eigen-decomposition)
                    int Rn = R, Sn = S;
                    if (ncp > 0) {
                        ND_STAMP_BEGIN;
                        cheb_grid_rule_nd<N, TK>(Sm, ncp, role64);
                        Rn = ncp * ncp; Sn = ncp;
                        warm_mask = 0;        // (the eigenvector tiles were scratch)
                        ND_STAMP(3);
                    } else {
                        jacobi_nd<N, TK>(Sm, 0, 2, poisoned ? 0 : warm_mask, role64);
                        weights_nd<N, TK>(Sm, role64);
                        warm_mask = poisoned ? 0 : 3;
                    }
                    ND_STAMP_BEGIN;
                    constexpr int LS = L::LS;
                    const double* lam = Sm + L::oLam;
                    const double* W = Sm + L::oW;
                    const double qm0 = mean0, qm1 = mean1, qs0 = scale0, qs1 = scale1;
                    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                    dispatch_extent<L::kMaxD>(a.D, [&](auto Dc) {
                        for (int e = ND_TID; e < Rn; e += 256) {
                            const int i0 = e / Sn, i1 = e - i0 * Sn;
                            const double w = W[e];
                            const double x0 = fma(lam[i0], qs0, qm0), x1 = fma(lam[LS + i1], qs1, qm1);
                            s0 = fma(w, poly2d_full<Dc>(coef + 0 * DD, x0, x1), s0);       // mu_0(x)
                            s1 = fma(w, poly2d_full<Dc>(coef + 1 * DD, x0, x1), s1);       // mu_1(x)
                            if (scaled) {  // scale <- sqrt(sum w var_k(x)), filtering.py:186
                                s2 = fma(w, poly2d_full<Dc>(coef + 2 * DD, x0, x1), s2);
                                s3 = fma(w, poly2d_full<Dc>(coef + 4 * DD, x0, x1), s3);
                            }
                        }
                    });
                    s0 = wave_sum64(s0); s1 = wave_sum64(s1);
                    if (scaled) { s2 = wave_sum64(s2); s3 = wave_sum64(s3); }
                    if ((ND_TID & 63) == 0) {
                        double* r3 = red + 4 * RW * (ND_TID >> 6) + 16 * ZB;
                        r3[1] = s0; r3[2] = s1; r3[4] = s2; r3[5] = s3;
                    }
                    __syncthreads();
                    {
                        const double* q = red + 16 * ZB;
                        s0 = q[1] + q[4 * RW + 1] + q[8 * RW + 1] + q[12 * RW + 1];
                        s1 = q[2] + q[4 * RW + 2] + q[8 * RW + 2] + q[12 * RW + 2];
                        s2 = q[4] + q[4 * RW + 4] + q[8 * RW + 4] + q[12 * RW + 4];
                        s3 = q[5] + q[4 * RW + 5] + q[8 * RW + 5] + q[12 * RW + 5];
                    }
                    if (!raw) { c0 = s0; c1 = s1; }
                    if (scaled) { ns0 = sqrt(s2); ns1 = sqrt(s3); }
                    ND_STAMP(6);
                    const int lane16 = ND_TID & 15;
                    const int cls = ((lane16 & 1) << 3) | ((lane16 & 2) << 1) | ((lane16 & 4) >> 1) | ((lane16 & 8) >> 3);
                    double* myred = red + (ND_TID >> 4) * RW + cls;
                    //   M(0,b) = m_1 M(0,b-1) + (b-1) S_11 M(0,b-2)
                    //   M(a,b) = m_0 M(a-1,b) + (a-1) S_00 M(a-2,b) + b S_01 M(a-1,b-1)
                    // three rows of the table live at a time; entries are emitted row by row: slot e(a, b) = a P - a(a-1)/2 + b
                    dispatch_extent<L::kMaxD>(a.D, [&](auto Dc) {
                    for (int base = 0; base < Rn; base += 256) {
                        // (a wave whose 64 nodes of this pass all lie beyond the rule adds zeros: it skips the pass -- at 23 x 23
                        //  nodes that is three of the twelve wave-passes, issue slots the co-resident workgroup can use)
                        if (base > 0 && base + (__builtin_amdgcn_readfirstlane(ND_TID) & ~63) >= Rn) continue;
                        double wA, mA0, mA1, sA00, sA01, sA11;
                        {
                            const int eA = base + ND_TID;
                            const bool okA = eA < Rn;
                            const int iA0 = okA ? eA / Sn : 0, iA1 = okA ? eA - iA0 * Sn : 0;
                            wA = okA ? W[eA] : 0.0;
                            const double xA0 = fma(lam[iA0], qs0, qm0), xA1 = fma(lam[LS + iA1], qs1, qm1);
                            mA0 = poly2d_full<Dc>(coef + 0 * DD, xA0, xA1) - c0;
                            mA1 = poly2d_full<Dc>(coef + 1 * DD, xA0, xA1) - c1;
                            sA00 = poly2d_full<Dc>(coef + 2 * DD, xA0, xA1);
                            sA01 = poly2d_full<Dc>(coef + 3 * DD, xA0, xA1);
                            sA11 = poly2d_full<Dc>(coef + 4 * DD, xA0, xA1);
                        }
                        double MA[3][P], bt[16];
                        static_for<0, P>([&](auto N0c) {
                            constexpr int n0 = N0c, r = n0 % 3, r1 = (n0 + 2) % 3, r2 = (n0 + 1) % 3;   // rows n0, n0-1, n0-2
                            static_for<0, P - n0>([&](auto N1c) {
                                constexpr int n1 = N1c;
                                double vA;
                                if constexpr (n0 == 0) {
                                    if constexpr (n1 == 0) vA = wA;       // (the recursion is linear: started from the node's weight it emits the weighted moments, one multiply per entry less)
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
                                bt[e % 16] = vA;
                                if constexpr (e % 16 == 15 || e == Z - 1) {
                                    if constexpr (e % 16 != 15) static_for<e % 16 + 1, 16>([&](auto Jc) { bt[Jc] = 0.0; });
                                    const double v = row_reduce16(bt, lane16);
                                    double* slot = myred + 16 * (e / 16);
                                    *slot = (base == 0) ? v : *slot + v;
                                }
                            });
                        });
                    }
                    });
                    __syncthreads();
                    if (ND_TID < Z) {
                        int zn0, zn1; nd_exponents(min(ND_TID & 127, Z - 1), zn0, zn1);    // the moment this thread owns (ND_TID & 127 < Z)
                        const int zi = ND_TID, n0 = zn0, n1 = zn1;
                        const int e = n0 * P - n0 * (n0 - 1) / 2 + n1;    // predictions of a Normal closure emit row by row
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) v += red[q * RW + e];
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
                    ND_STAMP(7);
                }
                if (ND_TID == 0) {
                    if (!raw) { st[0] = c0; st[1] = c1; }
                    if (scaled) { st[2] = ns0; st[3] = ns1; }
                }
                __syncthreads();
                bad = bad || (red[16 * ZB] != 0.0);
                __syncthreads();
                if (ND_TID == 0) red[16 * ZB] = 0.0;
            }
            // =========================================================================================================
            // update (filtering.py:268-275 / :333-339 / :192-202): bilinear form with h_k = lik_k(X_k) e_0
            // =========================================================================================================
            else {
                // h_k = lik_k(X_k) e_0 and K_k h_k: Chebyshev evaluation (no eigen-decomposition), checked; they give p_y and the
                // posterior mean, and the powers of the matrices shifted THERE give the posterior central moments directly.
                // Cyclic Jacobi + spectral evaluation (and a binomial shift of the moment array) only if the coefficients of
                // some factor have not converged.
                ND_STAMP_BEGIN;
                if (ND_TID == 0) Sm[L::oMisc + 6] = 0.0;
                // this step's measurements (loaded at the top of the step -- 1.5 k cycles earlier -- they cost four registers
                // across the prediction and the front end, and the pass got 3 % slower)
                double ypre[MFS_ND_MAX_FACTORS];
                static_for<0, MFS_ND_MAX_FACTORS>([&](auto Qc) { ypre[Qc] = (Qc < a.n_factors) ? yrow[(size_t)t * a.ny + a.fac_ycol[Qc]] : 0.0; });
                __syncthreads();
                double c0 = 0.0, c1 = 0.0;
                bool done_joint = false;
                if constexpr (L::kJoint) {
                  if (joint) {
                    // ---- a likelihood of both components (filtering.py:263-275 evaluates measurement_cond_pdf at the tensor
                    //      nodes for ANY callable): sum_ij W_ij l(y, x_ij) xi_0^a xi_1^b over the node set of the Normal-closure
                    //      prediction.  Default: the reference's own s^2 eigen-nodes (both K_k diagonalised by cyclic Jacobi, weights
                    //      from the eigenvectors, quadratures.py:163-170) -- no assumption on l at all; on the bearing-only example
                    //      (N = 4) NLL 4e-13, moments 4e-10 from the oracle.  MFS_ND_UPDATE=grid: the NCP x NCP Chebyshev grid with
                    //      the weights Omega of the rule's bilinear form, exact for polynomial integrands only -- l is interpolated
                    //      on a Gershgorin box, which on the same example costs the odd high moments five digits (7e-5)
                    done_joint = true;
                    int Rn = R, Sn = S;
                    bool on_grid = false;
                    if constexpr (TK == 1) {
                        if (a.joint_grid && ncp > 0) {
                            cheb_grid_rule_nd<N, TK>(Sm, ncp, role64);
                            Rn = ncp * ncp; Sn = ncp;
                            warm_mask = 0;
                            on_grid = true;
                        }
                    }
                    if (!on_grid) {
                        jacobi_nd<N, TK>(Sm, 0, 2, poisoned ? 0 : warm_mask, role64);
                        weights_nd<N, TK>(Sm, role64);
                        warm_mask = poisoned ? 0 : 3;
                    }
                    constexpr int LS = L::LS;
                    const double* lam = Sm + L::oLam;
                    const double* W = Sm + L::oW;
                    const double yv = ypre[0];
                    const int lane16 = ND_TID & 15;
                    const int cls = ((lane16 & 1) << 3) | ((lane16 & 2) << 1) | ((lane16 & 4) >> 1) | ((lane16 & 8) >> 3);
                    double* myred = red + (ND_TID >> 4) * RW + cls;
                    for (int base = 0; base < Rn; base += 256) {
                        if (base > 0 && base + (__builtin_amdgcn_readfirstlane(ND_TID) & ~63) >= Rn) continue;     // (as in the prediction's node pass)
                        const int eA = base + ND_TID;
                        const bool okA = eA < Rn;
                        const int iA0 = okA ? eA / Sn : 0, iA1 = okA ? eA - iA0 * Sn : 0;
                        const double xi0 = lam[iA0] * scale0, xi1 = lam[LS + iA1] * scale1;      // node - mean
                        const double wl = okA ? W[eA] * likelihood_joint_nd(a.fac_kind[0], Sm + L::oLik, yv, xi0 + mean0, xi1 + mean1) : 0.0;
                        double MA[2][P], bt[16];
                        static_for<0, P>([&](auto N0c) {
                            constexpr int n0 = N0c, r = n0 % 2, r1 = (n0 + 1) % 2;
                            static_for<0, P - n0>([&](auto N1c) {
                                constexpr int n1 = N1c;
                                double vA;
                                if constexpr (n0 == 0) { if constexpr (n1 == 0) vA = wl; else vA = xi1 * MA[0][n1 - 1]; }
                                else vA = xi0 * MA[r1][n1];
                                MA[r][n1] = vA;
                                constexpr int e = n0 * P - n0 * (n0 - 1) / 2 + n1;
                                bt[e % 16] = vA;
                                if constexpr (e % 16 == 15 || e == Z - 1) {
                                    if constexpr (e % 16 != 15) static_for<e % 16 + 1, 16>([&](auto Jc) { bt[Jc] = 0.0; });
                                    const double v = row_reduce16(bt, lane16);
                                    double* slot = myred + 16 * (e / 16);
                                    *slot = (base == 0) ? v : *slot + v;
                                }
                            });
                        });
                    }
                    __syncthreads();
                    if (ND_TID < Z) {          // the sums about the OLD mean into the moment array
                        int zn0, zn1; nd_exponents(min(ND_TID & 127, Z - 1), zn0, zn1);    // the moment this thread owns (ND_TID & 127 < Z)
                        const int n0 = zn0, n1 = zn1;
                        const int e = n0 * P - n0 * (n0 - 1) / 2 + n1;
                        double v = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) v += red[q * RW + e];
                        M[n0 * MLD + n1] = v;
                    }
                    __syncthreads();
                    const double py = M[0];
                    const double ipy = 1.0 / py;
                    if (!raw) { c0 = fma(M[1 * MLD], ipy, mean0); c1 = fma(M[1], ipy, mean1); }
                    if (ND_TID == 0) st[4] -= fast_log<true>(py);
                    __syncthreads();   // (everybody has read M[0], M[1], M[MLD] before the shift overwrites M)
                    shift_moments_nd<N, TK, P>(Sm, P, P - 1, c0 - mean0, c1 - mean1, ipy, role64);
                  }
                }
                if (!done_joint) {
                cheb_h_nd<N, TK>(Sm, a, (MFS_ND_FORCE_JACOBI || a.force_eigen) ? 0 : lik_mask, ypre, mean0, mean1, scale0, scale1, role64);
                __syncthreads();
                if (MFS_ND_FORCE_JACOBI || a.force_eigen || Sm[L::oMisc + 6] != 0.0) {
                    jacobi_nd<N, TK>(Sm, ubeg, uend, poisoned ? 0 : warm_mask, role64);
                    warm_mask = poisoned ? 0 : (warm_mask | lik_mask);
                    __syncthreads();
                    // ---- g_k[i] = lik_k(x_k,i) V_k[0][i], the spectral coefficients of h_k (or h_k = e_0 where no factor reads
                    //      component k); parked in the rotation records, which are free until the next Jacobi
                    double* g = Sm + L::oCs;
                    if (ND_TID < 2 * S) {
                        const int k = ND_TID / S, i = ND_TID - k * S;
                        if ((lik_mask >> k) & 1) {
                            const double x = fma(Sm[L::oK + k * NP * LD + i * (LD + 1)], k ? scale1 : scale0, k ? mean1 : mean0);
                            double l = 1.0;
                            static_for<0, MFS_ND_MAX_FACTORS>([&](auto Fc) {
                                if (Fc < a.n_factors && a.fac_comp[Fc] == k) l *= likelihood_nd(a.fac_kind[Fc], Sm + L::oLik + 4 * Fc, ypre[Fc], x);
                            });
                            g[k * NP + i] = l * Sm[L::oV + k * NP * LD + i];
                        }
                    }
                    __syncthreads();
                    // ---- PK[k][p] = X-powers applied to h_k.  Where K_k was diagonalised this is spectral,
                    //      PK[k][p][r] = sum_i V_k[r][i] lambda_i^p g_k[i]  (the tile now holds the eigenvalues, not K_k);
                    //      for the other component it is the Krylov recurrence from e_0 on the intact K_k.
                    double* GL = Sm + L::oM;     // [2][P][NP]: lambda_i^p g_k[i]; M / M2 are free until the moments are formed
                    static_assert(2 * P * NP <= 2 * NPW * MLD, "spectral coefficient table must fit in the M tiles");
                    for (int e = ND_TID; e < 2 * P * S; e += 256) {
                        const int k = e / (P * S), f = e - k * P * S, p = f / S, i = f - p * S;
                        if (!((lik_mask >> k) & 1)) continue;
                        const double lam_i = Sm[L::oK + k * NP * LD + i * (LD + 1)];
                        double v = g[k * NP + i];
                        for (int q = 0; q < p; ++q) v *= lam_i;
                        GL[(k * P + p) * NP + i] = v;
                    }
                    if (ND_TID < 2 * NP && !((lik_mask >> (ND_TID / NP)) & 1))
                        Sm[L::oPK + (ND_TID / NP) * NPW * NP + (ND_TID % NP)] = (ND_TID % NP == 0) ? 1.0 : 0.0;
                    __syncthreads();
                    for (int e = ND_TID; e < 2 * P * S; e += 256) {
                        const int k = e / (P * S), f = e - k * P * S, p = f / S, r = f - p * S;
                        if (!((lik_mask >> k) & 1)) continue;
                        const double* Vr = Sm + L::oV + k * NP * LD + r * LD;
                        const double* gl = GL + (k * P + p) * NP;
                        double a0 = 0.0, a1 = 0.0;
    #pragma unroll
                        for (int i = 0; i + 1 < S; i += 2) { a0 = fma(Vr[i], gl[i], a0); a1 = fma(Vr[i + 1], gl[i + 1], a1); }
                        if constexpr (S & 1) a0 = fma(Vr[S - 1], gl[S - 1], a0);
                        Sm[L::oPK + (k * NPW + p) * NP + r] = a0 + a1;
                    }
                    krylov_nd<N, TK>(Sm, P, 3 & ~lik_mask, 0.0, 0.0, role64);
                    __syncthreads();
                    ND_STAMP(15);
                    bilinear_moments_nd<N, TK>(Sm, P, P - 1, scale0, scale1, 1.0, role64);
                    __syncthreads();
                    const double py = M[0];
                    const double ipy = 1.0 / py;
                    if (!raw) { c0 = fma(M[1 * MLD], ipy, mean0); c1 = fma(M[1], ipy, mean1); }
                    if (ND_TID == 0) st[4] -= fast_log<true>(py);
                    __syncthreads();   // (everybody has read M[0], M[1], M[MLD] before the shift overwrites M)
                    shift_moments_nd<N, TK, P>(Sm, P, P - 1, c0 - mean0, c1 - mean1, ipy, role64);
                } else {
                    // p_y = h_0 . h_1 and the first moments (K_0 h_0) . h_1, h_0 . (K_1 h_1): every wave forms them itself (lane
                    // products and three wave sums), so no broadcast is needed
                    const int lane = ND_TID & 63;
                    const double* p0 = Sm + L::oPK;
                    const double* p1 = Sm + L::oPK + NPW * NP;
                    const bool in = lane < S;
                    const double h0 = in ? p0[lane] : 0.0, u0 = in ? p0[NP + lane] : 0.0;
                    const double h1 = in ? p1[lane] : 0.0, u1 = in ? p1[NP + lane] : 0.0;
                    const double py = wave_sum64(h0 * h1), a10 = wave_sum64(u0 * h1), a01 = wave_sum64(h0 * u1);
                    const double ipy = 1.0 / py;
                    const double dl0 = raw ? 0.0 : a10 * ipy, dl1 = raw ? 0.0 : a01 * ipy;      // posterior mean - mean, in units of lambda
                    if (!raw) { c0 = fma(scale0, dl0, mean0); c1 = fma(scale1, dl1, mean1); }
                    if (ND_TID == 0) st[4] -= fast_log<true>(py);
                    __syncthreads();   // (everybody has read PK[.][1] before the powers overwrite it)
                    krylov_nd<N, TK>(Sm, P, 3, dl0, dl1, role64);
                    __syncthreads();
                    ND_STAMP(15);
                    bilinear_moments_nd<N, TK>(Sm, P, P - 1, scale0, scale1, ipy, role64);
                    __syncthreads();
                }
                }   // (separable likelihood)
                double ns0 = 1.0, ns1 = 1.0;
                if (scaled) { ns0 = sqrt(M[2 * MLD]); ns1 = sqrt(M[2]); }   // posterior standard deviations (:195-197)
                if (ND_TID < Z) {
                    const int zi = ND_TID;
                    int zn0, zn1; nd_exponents(min(ND_TID & 127, Z - 1), zn0, zn1);    // the moment this thread owns (ND_TID & 127 < Z)
                    double v = M[zn0 * MLD + zn1];
                    if (scaled) v *= ipow32(1.0 / ns0, zn0) * ipow32(1.0 / ns1, zn1);
                    mom[zi] = v;
                    if (!finite(v)) red[16 * ZB] = 1.0;
                }
                if (ND_TID == 0) {
                    if (!raw) { st[0] = c0; st[1] = c1; }
                    if (scaled) { st[2] = ns0; st[3] = ns1; }
                }
                __syncthreads();
                bad = bad || (red[16 * ZB] != 0.0);
                __syncthreads();
                if (ND_TID == 0) red[16 * ZB] = 0.0;
                ND_STAMP(7);
#ifdef MFS_ND_STAMPS
                if (blockIdx.x == 0 && nd_tid(role64) == 0) g_nd_stamps[9] += 1;
#endif
            }
            }   // half
            bad = bad || !finite(st[4]) || !finite(st[0]) || !finite(st[1]) || !finite(st[2]) || !finite(st[3]);
            if (bad) { dead = true; if (tid == 0) Sm[L::oMisc + 5] = __hiloint2double(0, t); }
        } else {
            for (int zi = tid; zi < Z; zi += 256) mom[zi] = qnan;
            if (tid == 0) {
                st[0] = st[1] = st[4] = qnan;
                if (scaled) st[2] = st[3] = qnan;
            }
        }
        __syncthreads();
        if (a.out_mom) {
            double* dst = a.out_mom + ((size_t)b * a.T + t) * Z;
            for (int zi = tid; zi < Z; zi += 256) dst[zi] = mom[zi];
        }
        if (tid == 0 && a.out_mean) {
            a.out_mean[((size_t)b * a.T + t) * 2] = st[0];
            a.out_mean[((size_t)b * a.T + t) * 2 + 1] = st[1];
        }
        if (tid == 0 && a.out_scale) {
            a.out_scale[((size_t)b * a.T + t) * 2] = st[2];
            a.out_scale[((size_t)b * a.T + t) * 2 + 1] = st[3];
        }
    }
    if (a.t_end < a.T && cw) {      // not the last chunk: park the state
        for (int e = tid; e < Z; e += 256) cw[e] = mom[e];
        for (int e = tid; e < 2 * NP * LD; e += 256) cw[Z + 8 + e] = Sm[L::oV + e];
        if (tid == 0) {
            cw[Z] = st[0]; cw[Z + 1] = st[1]; cw[Z + 2] = st[2]; cw[Z + 3] = st[3]; cw[Z + 4] = st[4];
            cw[Z + 5] = Sm[L::oMisc + 5]; cw[Z + 6] = dead ? 1.0 : 0.0; cw[Z + 7] = (double)warm_mask;
        }
    } else if (tid == 0) {
        a.out_nell[b] = st[4];
        if (a.out_first_nan) a.out_first_nan[b] = __double2loint(Sm[L::oMisc + 5]);
    }
}

}  // namespace mfs

#undef ND_TID
