// filter1d_fast.hpp -- register-resident 1-D moment-filter step for gfx950 (the default path; the LDS-tile kernel in
// filter1d_kernel.hpp remains as the "dense" path: cross-checks, and the graded rules of stable=True).  stable=True and odd
// moment counts run on the extended variant of this kernel (EXT template parameter: the completed rule in closed form).
//
// A filter is owned by G lanes (G = 8: half a DPP row; 16: one row; 32: two rows; 64), lane l holding row l of the
// problem in VGPRs.  Cross-lane traffic is the DPP operand of the arithmetic instruction wherever the ISA allows it
// (v_fmac_f64_dpp row_newbcast), bank-masked moves for G = 8, gfx950 lane swaps (v_permlane16/32_swap) for G >= 32.
//
// Quadrature (reference mfs/one_dim/quadtures.py:122-133), restructured around two identities of Hankel moment
// matrices -- same mathematics, O(N^3/6 + N^2) work instead of O(12 N^3):
//
//   * H is G shifted by one moment: H = [G[:, 1:], g_N] with g_N = (m_N .. m_{2N-1}).  Since R^-1 G = R^T, the first
//     triangular solve R^-1 H costs nothing for N-1 of its columns (they are columns of R^T) and the remaining column
//     R^-1 g_N is exactly row N of the Cholesky factor of the (N+1) x N extended Hankel matrix: lane N carries it
//     through the same elimination.  (quadtures.py:128, inner triangular_solve)
//   * K = R^-1 H R^-T is the Jacobi (tridiagonal) matrix of the orthonormal polynomials (Golub & Welsch 1969): with
//     pivots piv_j = R_jj^2 and sub_j = R_{j+1,j} R_jj,
//         K_jj = sub_j / piv_j - sub_{j-1} / piv_{j-1},      K_{j+1,j}^2 = piv_{j+1} / piv_j.
//     The second triangular solve collapses to these ratios.  (quadtures.py:129, outer triangular_solve)
//   * Golub-Welsch eigensolve (quadtures.py:131-133) on the tridiagonal: lane k isolates eigenvalue k with Sturm
//     counts and reaches it with Laguerre's iteration on the characteristic polynomial (all lanes run the same
//     three-term recurrence on their own abscissa: no cross-lane traffic), started from the rule the filter already
//     knows (see the kernel); the squared first eigenvector components are w_k = 1 / sum_j c_j p_j(lambda_k)^2 with
//     c_j = piv_0 / piv_j.
//   * The Cholesky factor itself is never formed: only its pivots and sub-diagonal enter, so the elimination runs in
//     square-root-free (LDL^T) form.
//
// In exact arithmetic this equals the reference's dense route; in fp64 the two differ by the rounding-level
// off-tridiagonal noise of the dense K, which is the same size as the difference between any two dense
// implementations (DESIGN.md "parity"; tests/test_gpu_parity_1d.py compares both paths with the oracle).
#pragma once
#include <type_traits>

#include "filter1d_kernel.hpp"

namespace mfs {

#ifdef MFS_1D_STAMPS
// diagnostic build only (scratch/): cycles per phase accumulated by lane 0 of block 0
__device__ unsigned long long g_1d_stamps[16];
#define F1_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = clock64(); g_1d_stamps[slot] += now_ - t_last_; t_last_ = now_; } } while (0)
#define F1_STAMP_BEGIN unsigned long long t_last_ = clock64()
#else
#define F1_STAMP(slot) do {} while (0)
#define F1_STAMP_BEGIN do {} while (0)
#endif

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// value of `v` held by lane J of this lane's group
template <int G, int J>
__device__ __forceinline__ double bcast(double v) {
    if constexpr (G == 16) {
        // v_mov_b64_dpp row_newbcast:J.  bound_ctrl set (every lane has a source, so it changes nothing) lets the
        // compiler drop the pass-through operand; without it each broadcast drags a v_mov_b64 0 along
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);
    } else if constexpr (G == 8) {
        // two filters share a DPP row: lane J of the lower half goes to banks 0-1, lane J of the upper half to banks 2-3
        // (bank_mask leaves the other lanes of the destination alone, so the second move completes the first)
        // (written as asm: the builtin wants a defined pass-through value for the first move, a v_mov_b64 per broadcast,
        // although every lane is written by one of the two; s_nop 1 covers the VALU-write -> DPP-read hazard)
        double r;
        asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\t"
            "v_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc"
            : "=&v"(r) : "v"(v), "n"(J), "n"(8 + J));
        return r;
    } else if constexpr (G == 64) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), J);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(v), J);
        return __hiloint2double(hi, lo);
    } else {
        // G = 32: ds_swizzle in bit mode (and 0, or J): lane J of every 32-lane half -- the LDS crossbar like ds_bpermute,
        // but with the lane pattern in the instruction instead of an address register computed with VALU
        static_assert(G == 32, "lanes per filter");
        const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), J << 5);
        const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), J << 5);
        return __hiloint2double(hi, lo);
    }
}

// acc -= u * (v of lane J of the group).  For G = 16 the broadcast is the DPP modifier of the multiply-add itself
// (v_fmac_f64_dpp row_newbcast: one instruction instead of v_mov_b64_dpp + v_fma_f64; the compiler does not form it).
// s_nop 1 covers the two wait states a DPP read needs after a VALU write of its source.
template <int G, int J>
__device__ __forceinline__ void fnma_bcast(double& acc, const double u, const double v) {
    if constexpr (G == 16) {
        asm("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
            : "+v"(acc) : "v"(v), "v"(u), "n"(J));
    } else {   // (a bank-masked v_fmac_f64_dpp pair for G = 8 does not keep the masked lanes' accumulator: tools/diag/dpp_bank_test.hip)
        acc = fma(-u, bcast<G, J>(v), acc);
    }
}

// two columns' updates from the same factor in one block: one hazard nop serves both
template <int J0, int J1>
__device__ __forceinline__ void fnma_bcast2(double& a0, double& a1, const double u, const double v) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %2, -%3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, -%3 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1) : "v"(v), "v"(u), "n"(J0), "n"(J1));
}

template <int J0, int J1, int J2, int J3>
__device__ __forceinline__ void fnma_bcast4(double& a0, double& a1, double& a2, double& a3, const double u, const double v) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %4, -%5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, -%5 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, -%5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, -%5 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(v), "v"(u), "n"(J0), "n"(J1), "n"(J2), "n"(J3));
}

template <int J0>
__device__ __forceinline__ void fnma_bcast8(double* a, const double u, const double v) {
    asm("s_nop 1\n\tv_fmac_f64_dpp %0, %8, -%9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %8, -%9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %8, -%9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %8, -%9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %8, -%9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %8, -%9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %8, -%9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %8, -%9 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "v"(v), "v"(u), "n"(J0), "n"(J0 + 1), "n"(J0 + 2), "n"(J0 + 3), "n"(J0 + 4), "n"(J0 + 5), "n"(J0 + 6), "n"(J0 + 7));
}

// acc[j] -= u * (v of lane j - OFF of this lane's DPP row) for the columns j = J0 .. J1 - 1: blocks of eight, then singles
template <int J0, int J1, int OFF>
__device__ __forceinline__ void fnma_bcast_range(double* acc, const double u, const double v) {
    if constexpr (J0 < J1) {
        constexpr int n8 = (J1 - J0) / 8;
        static_for<0, n8>([&](auto Pc) { fnma_bcast8<J0 - OFF + 8 * Pc>(acc + J0 + 8 * Pc, u, v); });
        constexpr int j4 = J0 + 8 * n8, rem = J1 - j4;            // the remainder in blocks of 4 / 2 / 1: one hazard nop per block
        if constexpr (rem >= 4) fnma_bcast4<j4 - OFF, j4 - OFF + 1, j4 - OFF + 2, j4 - OFF + 3>(acc[j4], acc[j4 + 1], acc[j4 + 2], acc[j4 + 3], u, v);
        constexpr int j2 = j4 + ((rem >= 4) ? 4 : 0), rem2 = J1 - j2;
        if constexpr (rem2 >= 2) fnma_bcast2<j2 - OFF, j2 - OFF + 1>(acc[j2], acc[j2 + 1], u, v);
        constexpr int j1 = j2 + ((rem2 >= 2) ? 2 : 0);
        if constexpr (j1 < J1) fnma_bcast<16, j1 - OFF>(acc[j1], u, v);
    }
}

// acc = -u * (v of lane J of the group): the first term of a partial sum.  Where the broadcast is a separate move this
// is a plain multiplication; v_mul_f64 has no DPP form, so 16-lane groups keep the multiply-add into a zero.
template <int G, int J>
__device__ __forceinline__ void nmul_bcast(double& acc, const double u, const double v) {
    if constexpr (G == 16) {
        acc = 0.0;
        fnma_bcast<G, J>(acc, u, v);
    } else {
        acc = -u * bcast<G, J>(v);
    }
}

// v_min_f64 / v_max_f64 as single instructions: fmin() / fmax() compile to a canonicalising v_max_f64 x, x in front of
// each (signalling-NaN quieting), which doubles the cost of the Gershgorin bounds; the operands here are results of
// arithmetic, and a NaN among them has already poisoned the rule.
__device__ __forceinline__ double vmin_f64(const double a, const double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax_f64(const double a, const double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// sum over the G lanes of the group, result in every lane.  G = 16 is one DPP row: xor-1, xor-2 inside quads,
// then half-row mirror and row mirror -- 2 v_mov_b32_dpp + 1 v_add_f64 per stage at VALU latency instead of
// ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    // (bound_ctrl set: these patterns give every lane a source, and with it the compiler needs no pass-through value)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// gfx950 lane swaps: v_permlane16_swap on (v, copy of v) leaves the even DPP row's data in both rows of a pair in one
// result and the odd row's in the other; v_permlane32_swap does the same with the two 32-lane halves of the wave
__device__ __forceinline__ void row_dup(const double v, double& even_rows, double& odd_rows) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    even_rows = __hiloint2double(rh[0], rl[0]);
    odd_rows = __hiloint2double(rh[1], rl[1]);
}
__device__ __forceinline__ void half_dup(const double v, double& lower, double& upper) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    lower = __hiloint2double(rh[0], rl[0]);
    upper = __hiloint2double(rh[1], rl[1]);
}

template <int G>
__device__ __forceinline__ double gsum(double v) {
    if constexpr (G == 32 || G == 64) {
        // row sums by DPP, then the rows of a pair (and, for a whole-wave group, the two halves) through lane swaps:
        // every lane adds the same numbers in the same order
        double a, b;
        row_dup(gsum<16>(v), a, b);
        double t = a + b;
        if constexpr (G == 64) { half_dup(t, a, b); t = a + b; }
        return t;
    } else if constexpr (G == 16) {
        v += dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
        v += dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
        v += dpp_move<0x141>(v);  // row_half_mirror
        v += dpp_move<0x140>(v);  // row_mirror
        return __builtin_amdgcn_update_dpp(0.0, v, 0x150, 0xf, 0xf, true);  // lane 0's sum to every lane
    } else if constexpr (G == 8) {
        v += dpp_move<0xB1>(v);
        v += dpp_move<0x4E>(v);
        v += dpp_move<0x141>(v);  // row_half_mirror: the other quad of this half-row
        return bcast<8, 0>(v);
    } else {
        return group_sum<G>(v);
    }
}

// true when `flag` holds in every lane of this lane's group (all lanes of a group are active together)
template <int G>
__device__ __forceinline__ bool gall(bool flag, int grp) {
    if constexpr (G == 64) {
        return __builtin_amdgcn_ballot_w64(!flag) == 0ull;
    } else {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
        const unsigned long long want = ((G == 32) ? 0xffffffffull : (G == 16) ? 0xffffull : 0xffull) << (grp * G);
        return (m & want) == want;
    }
}

template <int G>
__device__ __forceinline__ bool gany(bool flag, int grp) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    if constexpr (G == 64) return m != 0ull;
    const unsigned long long want = ((G == 32) ? 0xffffffffull : (G == 16) ? 0xffffull : 0xffull) << (grp * G);
    return (m & want) != 0ull;
}

// acc[r] = row r of the coefficient table evaluated at u, for R rows at once.  The LDS copy of the table is
// degree-major and zero-padded to kCoefRows rows ([ceil4(degree + 1)][kCoefRows]): the coefficients of one degree are
// contiguous, read unconditionally in wide LDS reads that are all in flight together, and the Horner recurrences of the
// rows advance side by side -- no per-coefficient branch, no per-coefficient LDS latency.
constexpr int kCoefRows = (MFS_MAX_TERMS + 1 + 1) & ~1;   // 9 rows (operator terms + variance), padded to 10
template <int R>
__device__ __forceinline__ void horner_rows(const double* __restrict__ table, const int degree, const double u,
                                            double (&acc)[R]) {
    static_assert(R <= kCoefRows, "table rows");
    // the LDS table is also zero-padded at the top to a multiple of four degrees: the coefficients of four degrees are
    // read together (one LDS latency per four Horner steps instead of one per step; leading zeros leave acc at zero)
    const int top = (degree + 4) & ~3;
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    for (int jb = top - 4; jb >= 0; jb -= 4) {
        const double* t = table + jb * kCoefRows;
        double cj[4][R];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) cj[q][r] = t[q * kCoefRows + r];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 3; q >= 0; --q)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(acc[r], u, cj[q][r]);
    }
}

__device__ __forceinline__ double rcp_nr(const double v) {
    double y = __builtin_amdgcn_rcp(v);
    y = fma(fma(-v, y, 1.0), y, y);
    y = fma(fma(-v, y, 1.0), y, y);
    return y;
}

// the same, but 1 / (+-inf) = 0 like a true divide (the Newton steps alone turn it into NaN)
__device__ __forceinline__ double rcp_sat(const double v) {
    const double y0 = __builtin_amdgcn_rcp(v);
    double y = fma(fma(-v, y0, 1.0), y0, y0);
    y = fma(fma(-v, y, 1.0), y, y);
    return (y0 == 0.0) ? y0 : y;
}

__device__ __forceinline__ double rsq_nr(const double v) {
    double y = __builtin_amdgcn_rsq(v);
    y = fma(y, fma(-0.5 * v * y, y, 0.5), y);
    y = fma(y, fma(-0.5 * v * y, y, 0.5), y);
    return y;
}

// ---- elementary functions of the step, written for instruction count (the kernel is VALU-issue bound on one wave per
// SIMD): ~22 / 33 / 34 VALU instructions for exp / tanh / log against 34 / 155 / 90 for the ocml routines.  Absolute
// accuracy ~1-2 ulp of the result's magnitude scale; tanh keeps ABSOLUTE (not relative) accuracy near zero, which is
// what a polynomial in tanh x needs.
// (SC: the polynomial's constants as SCALAR operands of `v_fma_f64`.  The plain form compiles to `v_fmac_f64`, whose
//  addend is the destination, so every constant is first copied into a vector register pair -- loop-invariant copies that
//  get hoisted out of the time loop.  The 1-D kernels have the registers for that; the N-D kernel does not, spilled them and
//  reloaded each one from scratch between two dependent multiply-adds: ten serialised `s_waitcnt vmcnt(0)` per exponential.)
template <bool SC>
__device__ __forceinline__ double fma_const(const double p, const double r, const double c) {
    if constexpr (SC) {
        double o;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
        return o;
    } else {
        return fma(p, r, c);
    }
}
// (Default of the 1-D kernels: off.  MFS_EXP_SC=1 frees ~20 vector registers there -- N = 7: 229 -> 207 -- but they have no
//  spills to cure and the pass times do not move: headline 6.76 against 6.69 ms, config-4 shard 54.3 against 54.6.)
#ifndef MFS_EXP_SC
#define MFS_EXP_SC 0
#endif
template <bool SC = (MFS_EXP_SC != 0)>
__device__ __forceinline__ double fast_exp(const double y) {
    // e^y = 2^k e^r, k = rint(y / ln 2), |r| <= ln 2 / 2, Taylor to degree 13 (r^14 / 14! < 5e-18)
    const double yc = vmin_f64(vmax_f64(y, -745.5), 710.0);     // e^710 = +inf as in libm (the clamp would swallow a NaN: restored below)
    const double k = __builtin_rint(yc * 1.4426950408889634);
    double r = fma(k, -0.6931471803691238, yc);                  // ln 2 split: hi part has 21 trailing zero bits
    r = fma(k, -1.9082149292705877e-10, r);
    double p = 1.6059043836821613e-10;                           // 1 / 13!
    p = fma_const<SC>(p, r, 2.08767569878681e-09);
    p = fma_const<SC>(p, r, 2.505210838544172e-08);
    p = fma_const<SC>(p, r, 2.755731922398589e-07);
    p = fma_const<SC>(p, r, 2.7557319223985893e-06);
    p = fma_const<SC>(p, r, 2.48015873015873e-05);
    p = fma_const<SC>(p, r, 0.0001984126984126984);
    p = fma_const<SC>(p, r, 0.001388888888888889);
    p = fma_const<SC>(p, r, 0.008333333333333333);
    p = fma_const<SC>(p, r, 0.041666666666666664);
    p = fma_const<SC>(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double e = ldexp(p, (int)k);
    return (y != y) ? y : e;
}

__device__ __forceinline__ double fast_tanh(const double x) {
    const double a2 = vmin_f64(fabs(x) + fabs(x), 40.0);        // tanh(20) rounds to 1
    const double e = fast_exp(a2);
    const double t = fma(-2.0, rcp_nr(e + 1.0), 1.0);            // 1 - 2 / (e^{2|x|} + 1)
    return (x != x) ? x : copysign(t, x);
}

// natural logarithm of a positive finite number (anything else gives a non-finite result, which is all the caller needs:
// a non-finite negative log-likelihood poisons the replicate)
template <bool SC = (MFS_EXP_SC != 0)>
__device__ __forceinline__ double fast_log(const double v) {
    // v = m 2^e, m in [0.5, 1): ln m from the fp32 hardware log as a seed y0 and one exact correction
    // ln m = y0 + log1p(m e^{-y0} - 1), |m e^{-y0} - 1| ~ 1e-7
    const double m = __builtin_amdgcn_frexp_mant(v);
    const int ex = __builtin_amdgcn_frexp_exp(v);
    const double y0 = (double)(__builtin_amdgcn_logf((float)m) * 0.6931471805599453f);
    const double d = fma(m, fast_exp<SC>(-y0), -1.0);
    const double lnm = y0 + fma(-0.5 * d, d, d);
    const double r = fma((double)ex, 0.6931471805599453, lnm);
    const double inf = __builtin_inf();
    return (v == 0.0) ? -inf : (v == inf) ? inf : r;      // log 0 = -inf, log inf = inf as in libm
}

// weighted TME moments of one node for all orders n < M2 (see the call site); KT = number of operator terms.
//   f_n = sum_{k <= KT} q_k C(n, k) dx^(n-k),   q_0 = 1, q_k = k! Q_k.
// With G^(j)_n = sum_k C(n, k) q_{k+j} dx^(n-k) (so f_n = G^(0)_n, G^(j)_0 = q_j, G^(KT)_n = q_KT dx^n), Pascal's rule gives
// G^(j)_{n+1} = dx G^(j)_n + G^(j+1)_n: KT multiply-adds and one multiplication per order, the weight folded into the
// start values and, in scaled mode, the scale folded into dx and q_k (f_n / s^n = the same sums with dx / s, q_k / s^k).
template <int KT, int M2>
__device__ __forceinline__ void operator_moments(const double (&rows)[MFS_MAX_TERMS + 1], const double dx,
                                                 const double w, const double inv_sc, const bool node,
                                                 double* __restrict__ row) {
    double Gt[KT + 1];
    Gt[0] = w;
    {
        double fs = w;   // w k! / s^k
#pragma unroll
        for (int k = 1; k <= KT; ++k) { fs *= (double)k * inv_sc; Gt[k] = rows[k - 1] * fs; }
    }
    const double dxs = dx * inv_sc;
#pragma unroll
    for (int n = 0; n < M2; ++n) {
        row[n] = Gt[0];
#pragma unroll
        for (int jj = 0; jj < KT; ++jj) Gt[jj] = fma(dxs, Gt[jj], Gt[jj + 1]);
        Gt[KT] *= dxs;
    }
}

template <int N, int G>
struct FastTile {
    static constexpr int M2 = 2 * N;
    static constexpr int oMom = 0;             // [2N] moment vector (Hankel source)
    static constexpr int oTab = M2;            // [G][2N + 1] per-lane contributions (row padded: conflict-free); rows
    static constexpr int TLD = M2 + 1;         // N..G-1 belong to the spare lanes, written unconditionally, never read
    static constexpr int oLik = oTab + G * TLD;
    static constexpr int oLfac = (oLik + MFS_MAX_LIK + 1) & ~1;  // log(y!) for y = 0..kLfacMax (Poisson likelihood)
    static constexpr int oCoef = oLfac + 34;                    // model table, (degree + 1) * kCoefRows doubles
    static constexpr int fixedDoubles = oCoef;
    // Extended variant (stable = 1): a rule whose LDL^T has a pivot that is not > 0 takes the dense route (`quadrature` of
    // filter1d_kernel.hpp) in place -- its tiles [G -> R][K][rotations][v0][x][w] start at oTab like the contribution
    // table, which is scratch while a rule is formed -- so the tables behind (lik, lfac, coef) move up by kExtShift.
    static constexpr int kDenseEnd = (Tile<N>::oCoef + 1) & ~1;
    static constexpr int kExtShift = (kDenseEnd > oLik) ? kDenseEnd - oLik : 0;
};

constexpr int kMaxEigIters = 64;
constexpr int kLfacMax = 32;
#ifndef MFS_LAG_STOP
#define MFS_LAG_STOP 1e-6
#endif
constexpr double kLagStop = MFS_LAG_STOP;

// Poisson pmf with log(y!) looked up for the counts that actually occur (y <= 32) instead of a sum of logs per step
__device__ __forceinline__ double likelihood_fast(const int kind, const double* __restrict__ lp,
                                                  const double* __restrict__ lfac, const double y, const double x) {
    if (kind == MFS_LIK_POISSON_SOFTPLUS) {
        const double rate = fast_log(1.0 + fast_exp(lp[0] * x));
        const double lf = (y >= 0.0 && y <= (double)kLfacMax && y == floor(y)) ? lfac[(int)y] : log_factorial(y);
        return fast_exp(y * fast_log(rate) - rate - lf);
    }
    if (kind == MFS_LIK_BERNOULLI_LOGISTIC) {
        const double z = lp[0] + x * (lp[1] + x * (lp[2] + x * lp[3]));
        const double p = rcp_sat(1.0 + fast_exp(-z));   // e^{-z} may be +inf: p = 0, as 1 / (1 + inf) upstream
        return (y > 0.5) ? p : 1.0 - p;
    }
    if (kind == MFS_LIK_GAUSSIAN) {
        const double r = y - fma(lp[0], x, lp[1]);
        return fast_exp(-0.5 * r * r * rcp_nr(lp[2])) * rsq_nr(6.283185307179586476925 * lp[2]);
    }
    return likelihood(kind, lp, y, x);
}

// ---------------------------------------------------------------------------------------------------------------
// stable = 1 (mfs/utils.py:525-538), the rule of a COMPLETED factor.  R = L diag(f_k), f_k = d_k < 0 ? eps : sqrt(d_k),
// eps = 1e-8 ||G||_F, no longer satisfies R R^T = G -- but K = R^-1 H R^-T (quadtures.py:128-129) is STILL tridiagonal.  The
// rows of L^-1 are the monic polynomials pi_j orthogonal for the (now indefinite) moment functional -- all that needs is
// non-zero leading minors -- so M = L^-1 H L^-T has M_jj = <x pi_j, pi_j> = alpha_j d_j, M_{j+1,j} = <x pi_j, pi_{j+1}> = d_{j+1}
// and zeros elsewhere, whatever the signs of the d_j, and K = F^-1 M F^-1, F = diag(f):
//     K_jj = alpha_j s_j,    K_{j+1,j}^2 = (d_{j+1} / d_j) s_j s_{j+1},    s_j = d_j / f_j^2  (1 where d_j > 0, d_j / eps^2 < 0 else)
// with alpha_j and d_{j+1} / d_j the very ratios every other rule is built from.  (Checked in 80-digit arithmetic on the
// completed rules of config 2: the dense K's off-tridiagonal entries are < 1e-63 and these formulae hold to 1e-59; in fp64 the
// reference's dense K carries rounding noise amplified by 1 / eps^2 there, which is why two fp64 implementations of the
// reference part ways after a completion.)  The completed rule therefore costs one LDS pass for ||G||_F and N multiplications
// on top of a plain rule.  Its first-row eigenvector components come from the twisted factorisation below.

// Squared first component of the normalised eigenvector of the symmetric tridiagonal (a, b2) at the eigenvalue lam, from the
// twisted factorisation N_r D_r N_r^T of T - lam I (Parlett & Dhillon; LAPACK dlar1v): forward pivots Dp, backward pivots Dm,
// the twist where |gamma_r| = |Dp_r + Dm_r - (a_r - lam)| is smallest, z_r = 1 and
//   z_j^2 = b2_j / Dp_j^2 z_{j+1}^2 (j < r),    z_j^2 = b2_{j-1} / Dm_j^2 z_{j-1}^2 (j > r).
// The three-term recurrence the plain rules use for the same number is unstable when T is nearly reducible, which a
// completed factor makes it (off by up to 0.5 on config-2 cases; this form agrees with LAPACK's eigenvectors to 1e-13).
template <int N>
__device__ __forceinline__ double twisted_weight(const double (&a)[N], const double (&b2)[N], const double lam) {
    double ip[N], im[N], gam[N];    // 1 / Dp, 1 / Dm
    double d = a[0] - lam;
    static_for<0, N>([&](auto Jc) {
        constexpr int j = Jc;
        gam[j] = d;
        ip[j] = rcp_nr((d == 0.0) ? 1e-290 : d);
        if constexpr (j + 1 < N) d = fma(-b2[j], ip[j], a[j + 1] - lam);
    });
    d = a[N - 1] - lam;
    static_for<0, N>([&](auto Jc) {
        constexpr int j = N - 1 - Jc;
        gam[j] += d - (a[j] - lam);
        im[j] = rcp_nr((d == 0.0) ? 1e-290 : d);
        if constexpr (j > 0) d = fma(-b2[j - 1], im[j], a[j - 1] - lam);
    });
    int r = 0;
    double gmin = fabs(gam[0]);
    static_for<1, N>([&](auto Jc) { const double gj = fabs(gam[Jc]); const bool less = gj < gmin; r = less ? (int)Jc : r; gmin = less ? gj : gmin; });
    double zz = 1.0, sm = 1.0;
    static_for<0, N - 1>([&](auto Jc) {
        constexpr int j = N - 2 - Jc;
        const bool take = j < r;
        const double zn = zz * (b2[j] * ip[j] * ip[j]);
        zz = take ? zn : zz;
        sm += take ? zn : 0.0;
    });
    double yy = 1.0;
    static_for<1, N>([&](auto Jc) {
        constexpr int j = Jc;
        const bool take = j > r;
        const double yn = yy * (b2[j - 1] * im[j] * im[j]);
        yy = take ? yn : yy;
        sm += take ? yn : 0.0;
    });
    return zz * rcp_sat(sm);
}

// Gauss quadrature from the 2N moments in LDS.  Lane l < N returns node x and weight w; other lanes get w = 0.
// Returns the group-uniform poison flag (a Cholesky pivot was not > 0, as LAPACK potrf / XLA report).
template <int N, int G, bool EXT = false>
__device__ __forceinline__ bool quadrature_fast(const double* __restrict__ mom, const int l, const int grp,
                                                const double mean, const double scale, double& x_out, double& w_out,
                                                double& lam_io, const bool recentre = false, const bool atoms = false,
                                                const double w_atom = 0.0, double* __restrict__ S = nullptr,
                                                const int stable = 0, int* dense_rules = nullptr) {
    static_assert(N + 1 <= G, "needs one lane per row of the extended Hankel matrix");
    F1_STAMP_BEGIN;
    // -- row l of the extended Hankel matrix: g[j] = m[l + j], l = 0..N (quadtures.py:124-125)
    const int li = (l <= N) ? l : N;
    double g[N];
    static_for<0, N>([&](auto J) { g[J] = mom[li + J]; });
    // all N reads are issued here, before the elimination starts (left alone, the scheduler sinks them next to their
    // first uses and exposes one LDS latency per group of columns)
    __builtin_amdgcn_sched_barrier(0);

    F1_STAMP(0);
    // -- Cholesky, row per lane (quadtures.py:127; right-looking for 8- / 16- / 32-lane groups, left-looking by column for 64),
    //    kept in its square-root-free form: only the
    //    pivots d_j = L_jj^2 and the sub-diagonal enter the Jacobi matrix below, and L_ik L_jk = u_ik u_jk / d_k with the
    //    unnormalised columns u, so no square root is ever taken.  u_jk / d_k reaches the other lanes by DPP.
    double Ur[N], Vr[N];             // u_lk and u_lk / d_k of this lane's row
    // G = 32: a group spans two DPP rows, and row_newbcast only reaches inside a row.  gfx950's v_permlane16_swap on
    // (v, copy of v) leaves the even row's data in both rows of one register and the odd row's in both rows of the
    // other: with those two copies of u_lk / d_k the broadcast is again the DPP operand of the multiply-add
    // (lane j < 16 from the first copy, else from the second), instead of two ds_swizzle round trips per term.
    double piv[N], sub[N], ipiv[N];  // group-uniform
    bool poisoned = false;
    if constexpr (G == 16 || G == 8) {
        // right-looking order: as soon as column k is final every later column takes its term.  Each column is one
        // accumulator (its updates arrive a whole column apart, so no dependent-issue stall and no partial sums to add),
        // and for 16-lane groups the terms of one factor go out in asm blocks of 8 / 4 / 2 fused v_fmac_f64_dpp that
        // share a single DPP hazard nop (left-looking with two partial sums per column: 8.25 ms on config 2; this
        // order in pairs 7.79, fours 7.69, eights 7.62; the next pivot's column on its own in front: no better)
        static_for<0, N>([&](auto Kc) {
            constexpr int k = Kc;
            const double s = g[k];
            const double pj = bcast<G, k>(s);
            piv[k] = pj;
            sub[k] = bcast<G, k + 1>(s);
            poisoned |= !(pj > 0.0);
            // 1 / pivot: hardware v_rcp_f64 seed + two Newton-Raphson steps -> ~1e-16; the second step is folded into
            // the consumers (y1 (1 + delta) with delta = 1 - pivot y1) so that s * y1 and delta form side by side
            const double y0 = __builtin_amdgcn_rcp(pj);
            const double y1 = fma(fma(-pj, y0, 1.0), y0, y0);
            const double delta = fma(-pj, y1, 1.0);
            const double sy = s * y1;
            ipiv[k] = fma(y1, delta, y1);
            Ur[k] = s;
            Vr[k] = fma(sy, delta, sy);
            if constexpr (G == 16) {
                constexpr int n8 = (N - (k + 1)) / 8;
                static_for<0, n8>([&](auto Pc) { fnma_bcast8<k + 1 + 8 * Pc>(&g[k + 1 + 8 * Pc], Ur[k], Vr[k]); });
                constexpr int jq = k + 1 + 8 * n8;
                static_for<0, (N - jq + 3) / 4>([&](auto Pc) {
                    constexpr int j0 = jq + 4 * Pc;
                    if constexpr (j0 + 3 < N) fnma_bcast4<j0, j0 + 1, j0 + 2, j0 + 3>(g[j0], g[j0 + 1], g[j0 + 2], g[j0 + 3], Ur[k], Vr[k]);
                    else {
                        if constexpr (j0 + 1 < N) fnma_bcast2<j0, j0 + 1>(g[j0], g[j0 + 1], Ur[k], Vr[k]);
                        else if constexpr (j0 < N) fnma_bcast<G, j0>(g[j0], Ur[k], Vr[k]);
                        if constexpr (j0 + 2 < N) fnma_bcast<G, j0 + 2>(g[j0 + 2], Ur[k], Vr[k]);
                    }
                });
            } else {
                static_for<k + 1, N>([&](auto Jc) { fnma_bcast<G, Jc>(g[Jc], Ur[k], Vr[k]); });
            }
        });
    } else if constexpr (G == 32) {
        // the same order for two-row groups: the factor is duplicated over the rows once (row_dup) and used at once for
        // every later column -- lanes below 16 from the even-row copy, the others from the odd-row copy -- so no copies
        // of earlier factors are kept
        static_for<0, N>([&](auto Kc) {
            constexpr int k = Kc;
            const double s = g[k];
            const double pj = bcast<G, k>(s);
            piv[k] = pj;
            sub[k] = bcast<G, k + 1>(s);
            poisoned |= !(pj > 0.0);
            const double y0 = __builtin_amdgcn_rcp(pj);
            const double y1 = fma(fma(-pj, y0, 1.0), y0, y0);
            const double delta = fma(-pj, y1, 1.0);
            const double sy = s * y1;
            ipiv[k] = fma(y1, delta, y1);
            const double vk = fma(sy, delta, sy);
            if constexpr (k + 1 < N) {
                double va, vb;
                row_dup(vk, va, vb);
                constexpr int lo_end = (N < 16) ? N : 16;                   // columns k + 1 .. lo_end - 1 read the even-row copy
                if constexpr (k + 1 < lo_end) {
                    constexpr int n8 = (lo_end - (k + 1)) / 8;
                    static_for<0, n8>([&](auto Pc) { fnma_bcast8<k + 1 + 8 * Pc>(&g[k + 1 + 8 * Pc], s, va); });
                    static_for<k + 1 + 8 * n8, lo_end>([&](auto Jc) { fnma_bcast<16, Jc>(g[Jc], s, va); });
                }
                constexpr int hi0 = (k + 1 > 16) ? k + 1 : 16;              // columns >= 16: lane j - 16 of the odd-row copy
                if constexpr (hi0 < N) {
                    constexpr int n8 = (N - hi0) / 8;
                    static_for<0, n8>([&](auto Pc) { fnma_bcast8<hi0 - 16 + 8 * Pc>(&g[hi0 + 8 * Pc], s, vb); });
                    static_for<hi0 + 8 * n8, N>([&](auto Jc) { fnma_bcast<16, Jc - 16>(g[Jc], s, vb); });
                }
            }
        });
    } else   // whole-wave groups (N = 32, or on request): left-looking by column, the dot product in two partial sums
    static_for<0, N>([&](auto Jc) {
        constexpr int j = Jc;
        double s = g[j], s2 = 0.0;   // two partial sums: the dot product is not one dependent chain
        static_for<0, j>([&](auto Kc) {
            constexpr int k = Kc;
            constexpr int k0 = ((j - 1) % 2 == 1) ? 0 : 1;   // first term of s2
            if constexpr (k == k0) {
                nmul_bcast<G, j>(s2, Ur[k], Vr[k]);
            } else if constexpr ((j - 1 - k) % 2 == 0) fnma_bcast<G, j>(s, Ur[k], Vr[k]);   // k = j - 1 (the late one) lands here
            else fnma_bcast<G, j>(s2, Ur[k], Vr[k]);
        });
        if constexpr (j >= 2) s += s2;
        const double pj = bcast<G, j>(s);
        piv[j] = pj;
        sub[j] = bcast<G, j + 1>(s);
        poisoned |= !(pj > 0.0);
        // 1 / pivot: hardware v_rcp_f64 seed + two Newton-Raphson steps -> ~1e-16; the second step is folded into
        // the consumers (y1 (1 + delta) with delta = 1 - pivot y1) so that s * y1 and delta form side by side
        const double y0 = __builtin_amdgcn_rcp(pj);
        const double y1 = fma(fma(-pj, y0, 1.0), y0, y0);
        const double delta = fma(-pj, y1, 1.0);
        const double sy = s * y1;
        ipiv[j] = fma(y1, delta, y1);
        Ur[j] = s;
        Vr[j] = fma(sy, delta, sy);
    });

    F1_STAMP(1);
    double lam = 0.0, w = 0.0;
    double a[N], b2[N];              // the Jacobi matrix of the rule: diagonal and squared sub-diagonal (group-uniform)
    bool use_atoms = atoms, twisted = false;
    if constexpr (EXT) {
        // stable = 1 (mfs/utils.py:525-538): with every pivot > 0, R = L sqrt(D) IS the Cholesky factor and the rule is the
        // one formed below.  A pivot that is not > 0 (group-uniform; ~0.1 % of the rules of config 2) ends the Hankel
        // structure -- R R^T != G -- and only then the rule is that of the completed factor (completed_tridiagonal).
        if (stable && poisoned) {
            if (dense_rules) *dense_rules += 1;
            double fro = 0.0;
            static_for<0, N>([&](auto Jc) { const double v = mom[((l < N) ? l : N - 1) + Jc]; fro = fma(v, v, fro); });
            fro = gsum<G>((l < N) ? fro : 0.0);                 // ||G||_F^2
            const double ieps2 = 1e16 / fro;                    // 1 / eps^2
            const double qn = __builtin_nan("");
            double sj[N];
            static_for<0, N>([&](auto Jc) { sj[Jc] = (piv[Jc] > 0.0) ? 1.0 : ((piv[Jc] < 0.0) ? piv[Jc] * ieps2 : qn); });   // d_j = 0: 0 / 0 upstream
            static_for<0, N>([&](auto Jc) {
                constexpr int j = Jc;
                double aj = sub[j] * ipiv[j];
                if constexpr (j > 0) aj -= sub[j - 1] * ipiv[j - 1];
                a[j] = aj * sj[j];
                if constexpr (j < N - 1) b2[j] = (piv[j + 1] * ipiv[j]) * (sj[j] * sj[j + 1]); else b2[j] = 0.0;
            });
            use_atoms = false;
            twisted = true;
            poisoned = false;
        }
    }
    if (use_atoms) {
        // The moments are those of an N-atom measure whose atoms the caller holds (the posterior of an update: nodes
        // x_i, weights w_i l(y, x_i) / p_y), and the N-node Gauss rule of an N-atom measure IS that measure: the
        // Cholesky above has decided the poisoning exactly as the reference's does, the eigen-decomposition of
        // quadtures.py:128-133 would only return the atoms again, perturbed by the conditioning of the Hankel matrix
        // (1e-15 W at N = 7, 1e-9 ... 1e-7 W at N = 15; against 80-digit arithmetic the atoms are the closer of the two).
        lam = lam_io;
        w = w_atom;
    } else {
        // -- Jacobi matrix: a_j = K_jj, b2_j = K_{j+1,j}^2, and the weight normalisers c_j = piv_0 / piv_j
        if (!twisted) {
            static_for<0, N>([&](auto Jc) {
                constexpr int j = Jc;
                double aj = sub[j] * ipiv[j];
                if constexpr (j > 0) aj -= sub[j - 1] * ipiv[j - 1];
                a[j] = aj;
                if constexpr (j < N - 1) b2[j] = piv[j + 1] * ipiv[j]; else b2[j] = 0.0;
            });
        }
        double amin = 1.79e308, amax = -1.79e308, bmax2 = 0.0;
        static_for<0, N>([&](auto Jc) {
            constexpr int j = Jc;
            if constexpr (j < N - 1) bmax2 = vmax_f64(bmax2, b2[j]);
            amin = vmin_f64(amin, a[j]);
            amax = vmax_f64(amax, a[j]);
        });
        if constexpr (EXT) {
            // The eigenvalue iteration below resolves the spectrum to ~1e-15 of its WIDTH, and accepts on that scale.  A
            // completed rule can carry a node far outside the bulk with a negligible weight (an indefinite start: 1e13
            // standard deviations away); the rules that follow are graded the same way, and their bulk nodes would come
            // out with an error of 1e-15 x width.  Wider than 1e7 standard deviations of the rule's own measure
            // (b2_0 = its variance in rule units) -- or anything not finite -- takes the dense route: cyclic Jacobi on the
            // full K in the LDS tiles that alias the contribution table (`quadrature`, filter1d_kernel.hpp), which keeps the
            // relative accuracy of small eigenvalues.  Its eigenvalues come in no particular order (the filter's sums do
            // not care; as a start of the next iteration they only steer).
            const double wd0 = (amax - amin) + 4.0 * sqrt(bmax2);
            if (stable && !(wd0 * wd0 <= 1e14 * b2[0])) {
                if (dense_rules) *dense_rules += 1000;
#ifdef MFS_EXT_DEBUG
                const long long tj0 = clock64();
#endif
                quadrature<N, G>(S, l, mean, scale, 1);
#ifdef MFS_EXT_DEBUG
                if (dense_rules) dense_rules[2] += (int)((clock64() - tj0) >> 4);
#endif
                const int li2 = (l < N) ? l : N - 1;
                const double xd = S[Tile<N>::oX + li2], wdn = S[Tile<N>::oW + li2];
                wave_sync();                               // (every lane has its node before the tiles turn into the table again)
                lam_io = (xd - mean) / scale;
                x_out = (l < N) ? xd : mean;
                w_out = (l < N) ? wdn : 0.0;
                return !finite(wdn);
            }
        }

        F1_STAMP(2);
        if (!poisoned) {
            // -- eigenvalue k by Sturm counts + Laguerre iteration on the characteristic polynomial p_N (Li & Zeng 1994):
            //    for a polynomial with only real roots the Laguerre step from x towards the right (left) converges
            //    monotonically and cubically to the nearest root on that side, from ANY distance.  count(x) = number of
            //    eigenvalues below x tells the lane which root that is: count == k -> step right lands on lambda_k,
            //    count == k + 1 -> step left does; otherwise bisect the count bracket [lo, hi).
            // 2 max|b|, rounded up: only has to bound the spectrum (seed + one Newton step, then 1 + 1e-9)
            double rb = __builtin_amdgcn_rsq(bmax2);
            rb = fma(rb, fma(-0.5 * bmax2 * rb, rb, 0.5), rb);
            const double rad = (bmax2 > 0.0) ? 2.000000002 * bmax2 * rb : 0.0;
            const double width = (amax - amin) + 2.0 * rad;
            double lo = amin - rad - 1e-3 * width, hi = amax + rad + 1e-3 * width;
            const double wscale = fmax(fabs(lo), fabs(hi));
            const double tol = 1e-15 * wscale;  // ~4.5 eps ||K||: LAPACK-level absolute accuracy
            // the scale the early-acceptance rules below reason on: the spectrum's extent -- which for a rule with a stray
            // far node (extended variant: up to 1e7 standard deviations pass the test above) says nothing about the gaps
            // between the bulk eigenvalues, so there it is capped at 32 standard deviations of the rule's measure
            double wacc = wscale;
            if constexpr (EXT) wacc = fmin(wscale, 32.0 * sqrt(b2[0]));
            const int k = (l < N) ? l : N - 1;
            // start from this lane's eigenvalue of the previous rule when there is one (the Jacobi matrix moves little
            // between consecutive quadratures), else spread the lanes over the bracket
            double x = lo + (hi - lo) * ((double)k + 0.5) * (1.0 / (double)N);
            if (recentre) {
                // an approximate start is moved and stretched so that its first two spectral moments are right:
                // sum lambda = tr J = sum a_j,  sum lambda^2 = tr J^2 = sum a_j^2 + 2 sum b_j^2
                double tr = 0.0, tr2 = 0.0;
                static_for<0, N>([&](auto Jc) { tr += a[Jc]; tr2 = fma(a[Jc], a[Jc], tr2 + 2.0 * b2[Jc]); });
                const double gsel = (l < N) ? lam_io : 0.0;
                const double g1 = gsum<G>(gsel) * (1.0 / (double)N), g2 = gsum<G>(gsel * gsel) * (1.0 / (double)N);
                const double m1 = tr * (1.0 / (double)N), v_t = tr2 * (1.0 / (double)N) - m1 * m1, v_g = g2 - g1 * g1;
                // sqrt(v_t / v_g) to ~1e-7 (it only places a start): v_t rsq(v_t v_g)
                const double vv = v_t * v_g;
                double rr = __builtin_amdgcn_rsq(vv);
                rr = fma(rr, fma(-0.5 * vv * rr, rr, 0.5), rr);
                const double r = (v_t > 0.0 && v_g > 0.0) ? v_t * rr : 1.0;
                lam_io = fma(r, lam_io - g1, m1);
            }
            if (lam_io > lo && lam_io < hi) x = lam_io;
            bool conv = false;
            double prev_step = 0.0;
            for (int it = 0; it < kMaxEigIters; ++it) {
    #ifdef MFS_1D_STAMPS
                if (blockIdx.x == 0 && threadIdx.x == 0) g_1d_stamps[10] += 1;
    #endif
                // p, p' and h = p''/2 by the three-term recurrence (h_n = t h_{n-1} - b^2 h_{n-2} - p'_{n-1})
                double p0 = 1.0, p1 = a[0] - x, d0 = 0.0, d1 = -1.0, e0 = 0.0, e1 = 0.0;
                // Sturm count = number of sign changes along p_0..p_N.  The sign bits are shifted into one word as the
                // recurrence runs (one v_alignbit per step; floating-point compares would bounce through SGPR masks next
                // to the chain) and the changes are counted at the end.  p_0 = 1 > 0 is the zero bit above the first one.
                unsigned signs = (unsigned)__double2hiint(p1) >> 31;
                static_for<1, N>([&](auto Jc) {
                    constexpr int j = Jc;
                    const double t = a[j] - x;
                    const double pn = fma(t, p1, -b2[j - 1] * p0);
                    const double dn = fma(t, d1, fma(-b2[j - 1], d0, -p1));
                    const double en = fma(t, e1, fma(-b2[j - 1], e0, -d1));
                    signs = __builtin_amdgcn_alignbit(signs, (unsigned)__double2hiint(pn), 31);
                    p0 = p1; p1 = pn; d0 = d1; d1 = dn; e0 = e1; e1 = en;
                });
                const int cnt = __popc(signs ^ (signs >> 1));
                {   // straight-line and predicated (bitwise, not short-circuit, logic: no divergent branches in the loop);
                    // a converged lane is frozen by the selects at the end
                    const bool below = cnt <= k;
                    const double lo_n = below ? x : lo, hi_n = below ? hi : x;
                    const double mid = 0.5 * (lo_n + hi_n);
                    // S = sqrt((N-1) ((N-1) p'^2 - N p p'')) >= 0 for real-rooted p (clamped against rounding)
                    // (the step only has to be accurate enough to converge: seeds + one Newton step instead of
                    //  full-precision sqrt / divide; the accuracy of the root comes from the recurrence evaluation)
                    const double disc = fmax((double)(N - 1) * fma((double)(N - 1) * d1, d1, -(double)(2 * N) * p1 * e1), 0.0);
                    double rs = __builtin_amdgcn_rsq(disc);
                    rs = fma(rs, fma(-0.5 * disc * rs, rs, 0.5), rs);
                    const double S = copysign(disc > 0.0 ? disc * rs : 0.0, p1);
                    const bool right = (cnt == k), left = (cnt == k + 1);
                    const double den = right ? (d1 - S) : (d1 + S);
                    double rd = __builtin_amdgcn_rcp(den);
                    rd = fma(fma(-den, rd, 1.0), rd, rd);
                    double xn = x - (double)N * p1 * rd;
                    const bool ok = (right & (xn >= x) & (xn < hi_n)) | (left & (xn <= x) & (xn > lo_n));  // false for NaN
                    xn = ok ? xn : mid;
                    // Laguerre converges cubically near its root: e_next ~ e^3 / gap^2 with gap >~ W / N, so a step below
                    // 1e-6 W lands within ~N^2 1e-18 W of the root and the confirming evaluation can be skipped -- but only
                    // with evidence of that regime: a small step also occurs right after leaving the neighbourhood of a
                    // DIFFERENT root (steps then grow by ~N/(N-2) per iteration).  Hence: small AND at least 100x smaller
                    // than the previous Laguerre step of this lane ...
                    const double step = fabs(xn - x);
                    // ... or with the bound that needs no history: p'/p = sum_i 1/(x - lambda_i), so when the Newton step
                    // -p/p' points the way we travel, the roots ahead dominate that sum and the nearest of them lies within
                    // N |p/p'| of x.  If that is below 3e-7 W, the Laguerre step lands within ~(3e-7)^3 (N/W)^2 W of it.
                    // (This is what accepts, after ONE evaluation, the predict-half rule started from the reweighted
                    //  update-half rule -- see the kernel.)
                    const double nwt = -p1 * d1;   // sign of the Newton step -p/p'
                    const bool ahead = right ? (nwt > 0.0) : (nwt < 0.0);
                    const bool near = ok & ahead & ((double)N * fabs(p1) <= (3e-7 * wacc) * fabs(d1));
                    bool conv_n;
                    if constexpr (EXT) {
                        // A step below the tolerance is convergence only when the root it approaches lies AHEAD.  After a
                        // completed or dense rule the eigenvalues come back in another order, and a lane can start ON a
                        // root that is not its own (the atoms of a posterior survive a step to the last bit): travelling
                        // away from that root Laguerre's first steps are as small as the distance to it -- or zero -- and
                        // would be mistaken for convergence (two lanes on one eigenvalue, a rule of weight 1.5).  Such a
                        // lane is pushed off by a few tolerances instead; the count then says on which side its root lies.
                        const bool leaving = ok & !ahead & (step <= tol) & !((p1 == 0.0) & right);
                        const double push = right ? x + 4.0 * tol : x - 4.0 * tol;
                        xn = leaving ? (((push > lo_n) & (push < hi_n)) ? push : mid) : xn;
                        conv_n = (ok & !leaving & (step <= tol)) | (ok & (step <= kLagStop * wacc) & (step <= 1e-2 * prev_step)) |
                                 near | (hi_n - lo_n <= tol) | ((p1 == 0.0) & right);
                    } else {
                        conv_n = (ok & (step <= tol)) | (ok & (step <= kLagStop * wacc) & (step <= 1e-2 * prev_step)) |
                                 near | (hi_n - lo_n <= tol) | ((p1 == 0.0) & (right | left));
                    }
                    lo = conv ? lo : lo_n;
                    hi = conv ? hi : hi_n;
                    prev_step = conv ? prev_step : (ok ? step : 0.0);
                    x = conv ? x : xn;
                    conv = conv | conv_n;
                }
                if (gall<G>(conv, grp)) break;
            }
            lam = x;
            lam_io = x;
            F1_STAMP(3);
            // -- squared first eigenvector component: 1 / sum_j c_j p_j(lam)^2, c_j = piv_0 / piv_j  (quadtures.py:133, V[0, :]**2)
            double p0 = 1.0, p1 = a[0] - lam, acc = ipiv[0], acc2 = 0.0;
            static_for<1, N>([&](auto Jc) {
                constexpr int j = Jc;
                if constexpr (j % 2 == 1) acc = fma(ipiv[j] * p1, p1, acc); else acc2 = fma(ipiv[j] * p1, p1, acc2);
                const double pn = fma(a[j] - lam, p1, -b2[j - 1] * p0);
                p0 = p1; p1 = pn;
            });
            w = rcp_sat(piv[0] * (acc + acc2));   // the sum overflows for the outermost nodes of large rules: weight 0, not NaN
            if constexpr (EXT) { if (twisted) w = twisted_weight<N>(a, b2, lam); }
            F1_STAMP(4);
        }
    }
    const double qnan = __builtin_nan("");
    x_out = poisoned ? qnan : ((l < N) ? fma(scale, lam, mean) : mean);
    w_out = poisoned ? qnan : ((l < N) ? w : 0.0);
    return poisoned;
}

// ---------------------------------------------------------------------------------------------------------------
// the filter kernel (fast path)
// ---------------------------------------------------------------------------------------------------------------
// two waves per SIMD is the register budget that holds the unrolled state without scratch up to N = 16 (asking for
// three at N <= 8 spilled 24 dwords per lane and measured 12 % slower on config 4; four was 30 % slower); the 32- and
// 64-lane instantiations (N > 16) take the whole register file of a SIMD lane: their state does not fit in 256 VGPRs
// OCC = waves per SIMD the register budget is sized for.  Two (256 VGPRs) is the default for N <= 16; the orders whose
// time loop does not fit in 256 registers (N = 14..16: a few pointers spill to scratch, and every reload is a
// s_waitcnt vmcnt(0) in the step) also exist with OCC = 1 (512 registers), which the plan picks when the batch puts no
// more than one wave on a SIMD anyway.
// EXT = the extended variant: stable = 1 (LDL^T completion where a pivot is not > 0, see quadrature_fast) and odd moment
// counts (`extra`: one more moment per step, formed in the update and never read by a rule, filtering.py:65-66).  A separate
// instantiation, so that the plain kernel's registers and instruction stream are untouched.
template <int N, int G, int WPB, int OCC, bool EXT = false>
__global__ __launch_bounds__(WPB * 64, OCC) void filter1d_fast_kernel(const Filter1dArgs a, const int lds_doubles) {
    using L = FastTile<N, G>;
    constexpr int M2 = L::M2, TLD = L::TLD;
    constexpr int kShift = EXT ? L::kExtShift : 0;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.B) return;  // whole groups leave together; no block-level barrier is ever used
    double* S = smem + (size_t)slot * lds_doubles;
    double* mom = S + L::oMom;
    double* TAB = S + L::oTab;
    const double* coef = S + L::oCoef + kShift;
    const double* lp = S + L::oLik + kShift;
    double* lfac = S + L::oLfac + kShift;
    const int J1 = a.degree + 1;
    const int MS = EXT ? M2 + a.extra : M2;        // moments per row of m0 / out_moments

    {   // stage the model tables and the carry
        const double* src = a.coef + (a.coef_batched ? (size_t)b * a.n_rows * J1 : 0);
        for (int e = l; e < ((a.degree + 4) & ~3) * kCoefRows; e += G) {   // [row][degree] in HBM -> [degree][row] in LDS,
            const int j = e / kCoefRows, r = e - j * kCoefRows;             // zero-padded in both directions
            // operator tables: the variance row (source row n_terms) always lands in LDS row MFS_MAX_TERMS and the rows
            // between stay zero, so the time loop reads both from fixed registers whatever the number of terms
            int sr = r;
            if (a.trans_kind == MFS_TRANS_OPERATOR) sr = (r == MFS_MAX_TERMS) ? a.n_terms : (r < a.n_terms) ? r : a.n_rows;
            S[L::oCoef + kShift + e] = (sr < a.n_rows && j < J1) ? src[sr * J1 + j] : 0.0;
        }
        const double* ls = a.lik + (a.lik_batched ? (size_t)b * a.n_lik : 0);
        for (int e = l; e < MFS_MAX_LIK; e += G) S[L::oLik + kShift + e] = (e < a.n_lik) ? ls[e] : 0.0;
        if (a.lik_kind == MFS_LIK_POISSON_SOFTPLUS)
            for (int e = l; e <= kLfacMax; e += G) lfac[e] = log_factorial((double)e);
    }
    double mean = 0.0, scale = 1.0, nell = 0.0;
    int first_nan = -1;
    if (a.t_begin == 0) {
        const double* src = a.m0 + (a.m0_batched ? (size_t)b * MS : 0);
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
    const double* yrow = a.ys + (size_t)b * a.T;
    bool dead = (first_nan >= 0);
    const double qnan = __builtin_nan("");
    const bool node = (l < N);
    // The two rules of a step.  The posterior moments of the update half are the moments of the N-atom measure
    // {x_i, w_i l(y, x_i) / p_y}, and the N-node Gauss rule of an N-atom measure is that measure: in exact arithmetic the
    // predict-half rule of the next step IS the set of atoms this lane group already holds.  The reference reconstructs it
    // from the moments (Cholesky + eigensolve, quadtures.py:122-133); the kernel runs the Cholesky -- its pivots decide
    // the NaN poisoning exactly as upstream -- and takes nodes and weights from the atoms (MFS_PREDICT_RULE=recompute: the
    // reconstruction, started from the atoms and normally accepted after one Laguerre evaluation; the two differ by the
    // conditioning of the Hankel matrix, 1e-15 W at N = 7, 1e-9 ... 1e-7 W at N = 15, and against 80-digit arithmetic
    // they are equally close: tests/test_gpu_envelope.py).  The update-half rule (the predicted law is a mixture of N
    // continuous kernels, not N atoms) is computed in full; its eigenvalue iteration starts from the predict-half nodes,
    // moved and stretched to the trace and the squared Frobenius norm of the new Jacobi matrix (3.0 Laguerre evaluations
    // instead of 4 from the previous update rule's own eigenvalues).  A start only steers the iteration, never the
    // result; the atoms are part of the carry, so a chunked run is bit-identical.
    double gA = qnan, gW = 0.0;        // the posterior atoms of the last update, in rule units: nodes and weights
    bool have_atoms = false;
    if (a.t_begin != 0 && a.c_lam) {
        gA = a.c_lam[((size_t)b * 2 + 0) * G + l];
        gW = a.c_lam[((size_t)b * 2 + 1) * G + l];
        have_atoms = true;
    }
    double ywin = 0.0;                  // window of measurements (16 steps, or G when G < 16), one per lane
#ifdef MFS_EXT_DEBUG
    int dbg_dense[3] = {0, 0, 0};       // diagnostic build: rules of this replicate that took the slow routes; their cycles
#endif

    for (int t = a.t_begin; t < a.t_end; ++t) {
        // measurements: one coalesced 128-byte load per group every 16 steps, handed out by a lane shuffle
        constexpr int YW = (G < 16) ? G : 16;
        const int tw = (t - a.t_begin) & (YW - 1);
        if (tw == 0) {
            // the loaded window is moved into its own register HERE, so that the wait for the load (vmcnt is an in-order
            // counter: it also waits for every output store issued before) happens once per window, not at each step
            const double ld = (t + (l & (YW - 1)) < a.t_end) ? yrow[t + (l & (YW - 1))] : 0.0;
            asm volatile("v_mov_b64 %0, %1" : "=v"(ywin) : "v"(ld));
        }
        const double y = __shfl(ywin, tw, YW);
        double out0 = qnan, out1 = qnan;   // this lane's two moments of the step, kept in registers for the output store
        if (!dead) {
            int bad = 0;
            double gB = qnan;
#pragma nounroll
            for (int half = 0; half < 2; ++half) {
                double x, w;
                double lam_io = (half == 0) ? gA : gB;
#ifdef MFS_1D_STAMPS
                const unsigned long long it_before = g_1d_stamps[10];
#endif
                // predict half: the rule of the posterior moments is the posterior's own atoms (see quadrature_fast); the
                // reference's recomputation stays available as an A/B switch
                const bool atoms = (half == 0) & have_atoms & (a.recompute_rule == 0);
#ifdef MFS_EXT_DEBUG
                quadrature_fast<N, G, EXT>(mom, l, grp, mean, scale, x, w, lam_io, half == 1, atoms, gW, S, a.stable, dbg_dense);
#else
                quadrature_fast<N, G, EXT>(mom, l, grp, mean, scale, x, w, lam_io, half == 1, atoms, gW, S, a.stable);
#endif
                if (half == 0) gB = lam_io;
#ifdef MFS_1D_STAMPS
                if (blockIdx.x == 0 && threadIdx.x == 0) g_1d_stamps[11 + half] += g_1d_stamps[10] - it_before;
#endif
                F1_STAMP_BEGIN;
                const double u = (a.umap == MFS_U_TANH) ? fast_tanh(x) : x;
                double c = 0.0, inv_sc = 1.0, py = 1.0, ipy = 1.0;
                if (half == 0) {
                    // ---- prediction (filtering.py:76-79 / 144-148 / 221-225)
                    // every row of the model table at this lane's node, in one pass over the degrees
                    double rows[MFS_MAX_TERMS + 1];
                    horner_rows<MFS_MAX_TERMS + 1>(coef, a.degree, u, rows);
                    double mu, var;
                    if (a.trans_kind == MFS_TRANS_OPERATOR) {
                        mu = x + rows[0];
                        var = rows[MFS_MAX_TERMS];
                    } else {
                        mu = fma(a.mean_x_coef, x, rows[0]);
                        var = rows[1];
                    }
                    if (a.mode != MFS_MODE_RAW) {
                        mean = gsum<G>(w * mu);
                        c = mean;
                        if (a.mode == MFS_MODE_SCALED) {
                            scale = sqrt(gsum<G>(w * var));
                            inv_sc = 1.0 / scale;
                        }
                    }
                    double* row = TAB + l * TLD;   // spare lanes own spare rows: every store below is unconditional
                    if (a.trans_kind == MFS_TRANS_OPERATOR) {
                        // E[(X'-c)^n | x] = sum_k Q_k(u) n!/(n-k)! (x-c)^(n-k) = sum_k (k! Q_k) C(n,k) dx^(n-k);
                        // E_k(n) = C(n,k) dx^(n-k) advances in n by Pascal's rule E_k(n+1) = dx E_k(n) + E_{k-1}(n)
                        // (moments.py:141-179).  Unrolled for the number of operator terms the model really has.
                        switch (a.n_terms) {
                            case 2: operator_moments<2, M2>(rows, x - c, w, inv_sc, node, row); break;
                            case 4: operator_moments<4, M2>(rows, x - c, w, inv_sc, node, row); break;
                            case 6: operator_moments<6, M2>(rows, x - c, w, inv_sc, node, row); break;
                            // any other term count: the rows past the last term are zero in the LDS table
                            default: operator_moments<MFS_MAX_TERMS, M2>(rows, x - c, w, inv_sc, node, row); break;
                        }
                    } else {
                        // normal closure: E_0 = 1, E_1 = m, E_n = m E_{n-1} + (n-1) v E_{n-2}   (moments.py:70-74), run on
                        // F_n = w E_n / s^n directly: F_n = (m/s) F_{n-1} + (n-1) (v/s^2) F_{n-2}, F_0 = w
                        const double ms = (mu - c) * inv_sc, vs = var * inv_sc * inv_sc;
                        double f2 = w, f1 = w * ms, vn = vs;
                        row[0] = f2;
                        row[1] = f1;
#pragma unroll
                        for (int n = 2; n < M2; ++n) {
                            const double f = fma(ms, f1, vn * f2);
                            row[n] = f;
                            vn += vs;
                            f2 = f1;
                            f1 = f;
                        }
                    }
                } else {
                    // ---- update (filtering.py:82-85 / 151-157 / 228-236)
                    const double wl = node ? w * likelihood_fast(a.lik_kind, lp, lfac, y, x) : 0.0;
                    py = gsum<G>(wl);
                    ipy = rcp_nr(py);
                    if (a.mode != MFS_MODE_RAW) {
                        mean = gsum<G>(wl * x) * ipy;
                        c = mean;
                    }
                    if (a.mode == MFS_MODE_SCALED) {
                        scale = sqrt(gsum<G>(wl * (x - c) * (x - c)) * ipy);
                        inv_sc = 1.0 / scale;
                    }
                    const double dx = (x - c) * inv_sc;
                    {   // the spare lanes shadow the last eigenvalue (the DPP read must run with the node lanes active)
                        const double last = bcast<G, N - 1>(dx);
                        gA = node ? dx : last;
                        gW = wl * ipy;                 // (0 in the spare lanes)
                        have_atoms = true;
                    }
                    double* row = TAB + l * TLD;
                    {   // wl dx^n in four interleaved chains (a dependent multiply costs several issue slots)
                        const double dx2 = dx * dx, dx4 = dx2 * dx2;
                        double pw[4] = {wl, wl * dx, wl * dx2, wl * dx2 * dx};
#pragma unroll
                        for (int n = 0; n < M2; ++n) {
                            row[n] = pw[n & 3];
                            pw[n & 3] *= dx4;
                        }
                        if constexpr (EXT) row[M2] = pw[M2 & 3];    // wl dx^(2N): the row's pad slot (odd moment counts)
                    }
                    nell -= fast_log(py);
                }
                F1_STAMP(5 + half);
                wave_sync();
                {   // column sums of the contribution table: each lane owns moments l and l + G (2N <= 2G - 2), summed
                    // together in three partial sums each -- a dependent add costs several issue slots on one wave
                    const int n0 = l, n1 = l + G;
                    const bool has0 = n0 < M2, has1 = n1 < M2;
                    const bool tails = EXT && a.extra && half == 1;               // odd count: the pad column is summed too,
                    const bool tail0 = tails && n0 == M2, tail1 = tails && n1 == M2;  // by the lane whose slot is order 2N
                    const double* t0 = TAB + ((has0 || tail0) ? n0 : 0);
                    const double* t1 = TAB + ((has1 || tail1) ? n1 : 0);
                    double c0[N], c1[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) { c0[i] = t0[i * TLD]; c1[i] = t1[i * TLD]; }
                    __builtin_amdgcn_sched_barrier(0);   // every read in flight before the first add (one LDS latency, not N)
                    double s0[3] = {0.0, 0.0, 0.0}, s1[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int i = 0; i < N; ++i) { s0[i % 3] += c0[i]; s1[i % 3] += c1[i]; }
                    double acc0 = (s0[0] + s0[1]) + s0[2], acc1 = (s1[0] + s1[1]) + s1[2];
                    if (half != 0) { acc0 *= ipy; acc1 *= ipy; }
                    if (has0) { mom[n0] = acc0; bad |= !finite(acc0); }
                    if (has1) { mom[n1] = acc1; bad |= !finite(acc1); }
                    out0 = acc0; out1 = acc1;     // (with an odd count out1 of lane 2N - G is the order-2N moment: written out, never carried)
                }
                wave_sync();
                F1_STAMP(7);
#ifdef MFS_1D_STAMPS
                if (blockIdx.x == 0 && threadIdx.x == 0) g_1d_stamps[9] += 1;
#endif
            }
            bad |= (int)(!finite(nell) || !finite(mean) || !finite(scale));
            if (gany<G>(bad != 0, grp)) { dead = true; first_nan = t; }
        } else {
            for (int n = l; n < M2; n += G) mom[n] = qnan;
            mean = qnan; scale = qnan; nell = qnan;
            wave_sync();
        }
        if (a.out_mom) {
            double* dst = a.out_mom + ((size_t)b * a.T + t) * MS;
            if (l < MS) dst[l] = out0;
            if (l + G < MS) dst[l + G] = out1;
        }
        if (l == 0) {
            if (a.out_mean) a.out_mean[(size_t)b * a.T + t] = mean;
            if (a.out_scale) a.out_scale[(size_t)b * a.T + t] = scale;
        }
    }
    if (a.t_end >= a.T) {
        if (l == 0) {
            a.out_nell[b] = nell;
            if (a.out_first_nan) a.out_first_nan[b] = first_nan;
#ifdef MFS_EXT_DEBUG
            if (EXT && a.out_mean && a.T > 3) { a.out_mean[(size_t)b * a.T + a.T - 1] = (double)dbg_dense[0]; a.out_mean[(size_t)b * a.T + a.T - 2] = (double)dbg_dense[1]; a.out_mean[(size_t)b * a.T + a.T - 3] = 16.0 * (double)dbg_dense[2]; }   // diagnostic build only
#endif
        }
    } else {
        for (int n = l; n < M2; n += G) a.c_mom[(size_t)b * M2 + n] = mom[n];
        if (a.c_lam) { a.c_lam[((size_t)b * 2 + 0) * G + l] = gA; a.c_lam[((size_t)b * 2 + 1) * G + l] = gW; }
        if (l == 0) {
            a.c_mean[b] = mean; a.c_scale[b] = scale; a.c_nell[b] = nell; a.c_first_nan[b] = first_nan;
        }
    }
}

template <int N, int G, int WPB>
__global__ __launch_bounds__(WPB * 64) void quadrature1d_fast_kernel(const Quad1dArgs a) {
    constexpr int M2 = 2 * N;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.B) return;
    double* S = smem + (size_t)slot * M2;
    for (int n = l; n < M2; n += G) S[n] = a.ms[(size_t)b * M2 + n];
    wave_sync();
    double x, w;
    double lam_dummy = __builtin_nan("");
    quadrature_fast<N, G>(S, l, grp, a.mean ? a.mean[b] : 0.0, a.scale ? a.scale[b] : 1.0, x, w, lam_dummy);
    if (l < N) {
        a.out_w[(size_t)b * N + l] = w;
        a.out_x[(size_t)b * N + l] = x;
    }
}

// the same with stable = 1: the completed rule needs the per-filter tile of the extended filter kernel
template <int N, int G, int WPB>
__global__ __launch_bounds__(WPB * 64) void quadrature1d_fast_ext_kernel(const Quad1dArgs a) {
    using L = FastTile<N, G>;
    constexpr int M2 = 2 * N, kStride = L::oLik + L::kExtShift;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.B) return;
    double* S = smem + (size_t)slot * kStride;
    for (int n = l; n < M2; n += G) S[n] = a.ms[(size_t)b * M2 + n];
    wave_sync();
    double x, w;
    double lam_dummy = __builtin_nan("");
    quadrature_fast<N, G, true>(S, l, grp, a.mean ? a.mean[b] : 0.0, a.scale ? a.scale[b] : 1.0, x, w, lam_dummy, false, false,
                                0.0, S, a.stable);
    if (l < N) {
        a.out_w[(size_t)b * N + l] = w;
        a.out_x[(size_t)b * N + l] = x;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// characteristic function from moments: E[exp(i z X)] ~= sum_n w_n exp(i z x_n)
// (mfs/one_dim/moments.py:309-337; callers dardel/benes_bernoulli/post_processing_mf.py:37-60).  One group per
// moment vector: the rule comes from quadrature_fast, every lane then holds all N (x, w) pairs and walks the z grid
// lane-parallel, so the complex outputs of a group are 16 consecutive 16-byte elements (coalesced stores).
// ---------------------------------------------------------------------------------------------------------------
struct Cf1dArgs {
    int count, nz;
    const double* ms;     // [count][2N]
    const double* mean;   // [count] or null
    const double* scale;  // [count] or null
    const double* zs;     // [nz]
    double* out;          // [count][nz][2]  (re, im)
};

template <int N, int G, int WPB>
__global__ __launch_bounds__(WPB * 64) void cf1d_fast_kernel(const Cf1dArgs a) {
    constexpr int M2 = 2 * N;
    constexpr int FPW = 64 / G;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / G, l = lane - grp * G;
    const int slot = wave * FPW + grp;
    const int b = blockIdx.x * (WPB * FPW) + slot;
    if (b >= a.count) return;
    double* S = smem + (size_t)slot * M2;
    for (int n = l; n < M2; n += G) S[n] = a.ms[(size_t)b * M2 + n];
    wave_sync();
    double x, w, lam = __builtin_nan("");
    quadrature_fast<N, G>(S, l, grp, a.mean ? a.mean[b] : 0.0, a.scale ? a.scale[b] : 1.0, x, w, lam);
    double xs[N], ws[N];
    static_for<0, N>([&](auto Jc) { xs[Jc] = bcast<G, Jc>(x); ws[Jc] = bcast<G, Jc>(w); });
    double* dst = a.out + (size_t)b * a.nz * 2;
    for (int k = l; k < a.nz; k += G) {
        const double z = a.zs[k];
        double re = 0.0, im = 0.0;
        static_for<0, N>([&](auto Jc) {
            double sn, cn;
            sincos(z * xs[Jc], &sn, &cn);
            re = fma(ws[Jc], cn, re);
            im = fma(ws[Jc], sn, im);
        });
        reinterpret_cast<double2*>(dst)[k] = make_double2(re, im);
    }
}

}  // namespace mfs
