// filter1d_inst.hip -- instantiates the 1-D kernels for quadrature orders MFS_NLO..MFS_NHI (one translation unit per
// range so that `make -j` compiles them in parallel) and registers their launchers.
#include "filter1d_kernel.hpp"

#ifndef MFS_NLO
#error "compile with -DMFS_NLO=.. -DMFS_NHI=.."
#endif

namespace mfs {

extern KernelEntry g_table[MFS_MAX_N + 1][3];  // [N][0: G=16, 1: G=32, 2: G=64], defined in capi.hip

constexpr int kBlockLdsBudget = 64 * 1024;

template <int N, int G>
constexpr int waves_per_block() {
    constexpr int per_wave = (64 / G) * Tile<N>::kDoubles * 8;
    return (4 * per_wave <= kBlockLdsBudget) ? 4 : (2 * per_wave <= kBlockLdsBudget) ? 2 : 1;
}

template <int N, int G>
hipError_t launch_filter(const Filter1dArgs& a, int grid, int lds, hipStream_t s) {
    constexpr int WPB = waves_per_block<N, G>();
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&filter1d_kernel<N, G, WPB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((filter1d_kernel<N, G, WPB>), dim3(grid), dim3(WPB * 64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
hipError_t launch_quad(const Quad1dArgs& a, int grid, int lds, hipStream_t s) {
    constexpr int WPB = waves_per_block<N, G>();
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&quadrature1d_kernel<N, G, WPB>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    hipLaunchKernelGGL((quadrature1d_kernel<N, G, WPB>), dim3(grid), dim3(WPB * 64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
void reg(int gi) {
    KernelEntry& e = g_table[N][gi];
    e.filter = &launch_filter<N, G>;
    e.quad = &launch_quad<N, G>;
    e.lds_doubles_per_filter = Tile<N>::kDoubles;
    e.waves_per_block = waves_per_block<N, G>();
}

template <int N>
void reg_all() {
    if constexpr (N <= 16) reg<N, 16>(0);
    if constexpr (N <= 32) reg<N, 32>(1);
    reg<N, 64>(2);
}

template <int LO, int HI>
struct RegRange {
    static void run() {
        reg_all<LO>();
        if constexpr (LO < HI) RegRange<LO + 1, HI>::run();
    }
};

struct Registrar {
    Registrar() { RegRange<MFS_NLO, MFS_NHI>::run(); }
};
static Registrar registrar_instance;

}  // namespace mfs
