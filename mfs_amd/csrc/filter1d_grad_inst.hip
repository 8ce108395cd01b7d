// filter1d_grad_inst.hip -- instantiates the forward-mode gradient kernels (N = 2..16 quadrature nodes, P = 1..4
// parameters; 8 lanes per filter up to N = 8, 16 above) and registers their launchers.
#include <cstdio>
#include "filter1d_grad.hpp"
#include "launch_util.hpp"

namespace mfs {

constexpr int kGradMaxN = 16, kGradMaxP = 4;
Filter1dGradLaunch g_grad_table[kGradMaxN + 1][kGradMaxP + 1];

// lanes per filter: lane l < N owns root l and the moment sums stride over the group, so N <= 8 runs on 8 lanes -- eight filters
// per wave instead of four, on the same instruction stream (MFS_GRAD_G_SMALL=16 for the A/B)
#ifndef MFS_GRAD_G_SMALL
#define MFS_GRAD_G_SMALL 8
#endif
template <int N> constexpr int grad_lanes() { return (N <= 8) ? MFS_GRAD_G_SMALL : 16; }

template <int N, int P>
int grad_lds_bytes() {
    constexpr int G = grad_lanes<N>(), DW = (1 + P) * 2 * N;
    return (64 / G) * (DW + G * DW + kCoefDoubles * (1 + P) + MFS_MAX_LIK * (1 + P) + ((N > 8) ? 2 * DW + 2 * (1 + P) * N : 0)) * 8;
}

template <int N, int P>
hipError_t launch_grad(const Filter1dGradArgs& a, int n_filters, hipStream_t s) {
    constexpr int G = grad_lanes<N>(), FPW = 64 / G;
    if (hipError_t e = ensure_dynamic_lds<&filter1d_grad_kernel<N, G, P>>(); e != hipSuccess) return e;
    const int lds = grad_lds_bytes<N, P>();
    hipLaunchKernelGGL((filter1d_grad_kernel<N, G, P>), dim3((n_filters + FPW - 1) / FPW), dim3(64), lds, s, a);
#ifdef MFS_GRAD_WARM_DEBUG
    {
        (void)hipStreamSynchronize(s);
        unsigned long long c[4];
        (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(g_grad_warm_dbg), sizeof(c));
        printf("[warm roots] rules x lanes %llu, warm-started %llu (%.3f); wave-level: %llu of %llu rules skipped the halvings\n", c[0], c[1],
               (double)c[1] / (double)c[0], c[2], c[3]);
    }
#endif
    return hipGetLastError();
}

template <int N, int P>
void reg_grad() {
    g_grad_table[N][P] = &launch_grad<N, P>;
    if constexpr (P < kGradMaxP) reg_grad<N, P + 1>();
    else if constexpr (N < kGradMaxN) reg_grad<N + 1, 1>();
}

struct GradRegistrar { GradRegistrar() { reg_grad<2, 1>(); } };
static GradRegistrar grad_registrar_instance;

}  // namespace mfs
