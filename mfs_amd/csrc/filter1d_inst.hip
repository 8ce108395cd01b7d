// filter1d_inst.hip -- instantiates the 1-D kernels for quadrature orders MFS_NLO..MFS_NHI (one translation unit per
// range so that `make -j` compiles them in parallel) and registers their launchers.
#include "filter1d_fast.hpp"
#include "launch_util.hpp"

#ifndef MFS_NLO
#error "compile with -DMFS_NLO=.. -DMFS_NHI=.."
#endif

namespace mfs {

extern KernelEntry g_table[MFS_MAX_N + 1][kSlots];  // defined in capi.hip
extern Filter1dFastLaunch g_fast_filter[MFS_MAX_N + 1][4];
extern Filter1dFastLaunch g_fast_filter_wide[MFS_MAX_N + 1][4];  // nullptr where the default budget does not spill
extern Filter1dFastLaunch g_fast_filter_ext[MFS_MAX_N + 1][4];       // extended variant (stable = 1, odd moment counts); default lane count only
extern Filter1dFastLaunch g_fast_filter_ext_wide[MFS_MAX_N + 1][4];
extern Quad1dLaunch g_quad_ext[MFS_MAX_N + 1][4];                     // quadrature entry with stable = 1
extern int g_quad_ext_lds[MFS_MAX_N + 1][4];                         // its LDS doubles per filter
extern int g_fast_ext_shift[MFS_MAX_N + 1][4];                      // extra LDS doubles per filter of the extended variant
using Cf1dLaunch = hipError_t (*)(const Cf1dArgs&, int grid, int lds, hipStream_t);
extern Cf1dLaunch g_cf[MFS_MAX_N + 1][4];

constexpr int kBlockLdsBudget = 64 * 1024;

template <int N, int G>
constexpr int waves_per_block() {
    constexpr int per_wave = (64 / G) * Tile<N>::kDoubles * 8;
    return (4 * per_wave <= kBlockLdsBudget) ? 4 : (2 * per_wave <= kBlockLdsBudget) ? 2 : 1;
}

template <int N, int G>
hipError_t launch_filter(const Filter1dArgs& a, int grid, int lds, hipStream_t s) {
    constexpr int WPB = waves_per_block<N, G>();
    if (hipError_t e = ensure_dynamic_lds<&filter1d_kernel<N, G, WPB>>(); e != hipSuccess) return e;
    hipLaunchKernelGGL((filter1d_kernel<N, G, WPB>), dim3(grid), dim3(WPB * 64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
hipError_t launch_quad(const Quad1dArgs& a, int grid, int lds, hipStream_t s) {
    constexpr int WPB = waves_per_block<N, G>();
    if (hipError_t e = ensure_dynamic_lds<&quadrature1d_kernel<N, G, WPB>>(); e != hipSuccess) return e;
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
    e.lanes_per_filter = G;
}

// ---- fast (register-resident) path: single-wave workgroups, LDS = filters per wave x (fixed + model table)
template <int N, int G, int OCC, bool EXT = false>
hipError_t launch_filter_fast(const Filter1dArgs& a, int grid, int lds_doubles, hipStream_t s) {
    if (hipError_t e = ensure_dynamic_lds<&filter1d_fast_kernel<N, G, 1, OCC, EXT>>(); e != hipSuccess) return e;
    hipLaunchKernelGGL((filter1d_fast_kernel<N, G, 1, OCC, EXT>), dim3(grid), dim3(64), (64 / G) * lds_doubles * 8, s, a,
                       lds_doubles);
    return hipGetLastError();
}

template <int N, int G>
hipError_t launch_quad_fast(const Quad1dArgs& a, int grid, int lds, hipStream_t s) {
    hipLaunchKernelGGL((quadrature1d_fast_kernel<N, G, 1>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
hipError_t launch_quad_fast_ext(const Quad1dArgs& a, int grid, int lds, hipStream_t s) {
    if (hipError_t e = ensure_dynamic_lds<&quadrature1d_fast_ext_kernel<N, G, 1>>(); e != hipSuccess) return e;
    hipLaunchKernelGGL((quadrature1d_fast_ext_kernel<N, G, 1>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
hipError_t launch_cf_fast(const Cf1dArgs& a, int grid, int lds, hipStream_t s) {
    hipLaunchKernelGGL((cf1d_fast_kernel<N, G, 1>), dim3(grid), dim3(64), lds, s, a);
    return hipGetLastError();
}

template <int N, int G>
void reg_fast(int gi) {
    g_cf[N][gi] = &launch_cf_fast<N, G>;
    KernelEntry& e = g_table[N][3 + gi];  // gi: 0..2 = G 16 / 32 / 64, 3 = G 8
    // (A/B switch.  Three waves per SIMD for N <= 8 -- the config-4 shard's kernel -- is a 168-register build that spills 66
    //  registers: 59-62 ms against 54.6 at two waves, round 3.)
#ifndef MFS_FAST_OCC_SMALL
#define MFS_FAST_OCC_SMALL 2
#endif
    constexpr int occ = (N <= 8) ? MFS_FAST_OCC_SMALL : (N <= 16) ? 2 : 1;
    g_fast_filter[N][gi] = &launch_filter_fast<N, G, occ>;
    // stable = 1 / odd moment counts: the extended variant, for the default lane count of the order (others: dense path)
    if constexpr (G == ((N + 1 <= 8) ? 8 : (N + 1 <= 16) ? 16 : (N + 1 <= 32) ? 32 : 64)) {
        g_fast_filter_ext[N][gi] = &launch_filter_fast<N, G, occ, true>;
        g_fast_ext_shift[N][gi] = FastTile<N, G>::kExtShift;
        g_quad_ext[N][gi] = &launch_quad_fast_ext<N, G>;
        g_quad_ext_lds[N][gi] = FastTile<N, G>::oLik + FastTile<N, G>::kExtShift;
        if constexpr (N >= 14 && N <= 16) g_fast_filter_ext_wide[N][gi] = &launch_filter_fast<N, G, 1, true>;
    }
    // one-wave-per-SIMD register budget for the orders that spill at two (their default lane count only)
    if constexpr (N >= 14 && N <= 16 && G == ((N + 1 <= 16) ? 16 : 32)) g_fast_filter_wide[N][gi] = &launch_filter_fast<N, G, 1>;
    e.filter = nullptr;
    e.quad = &launch_quad_fast<N, G>;
    e.lds_doubles_per_filter = FastTile<N, G>::fixedDoubles;
    e.waves_per_block = 1;
    e.lanes_per_filter = G;
}

template <int N>
void reg_all() {
    if constexpr (N <= 16) reg<N, 16>(0);
    if constexpr (N <= 32) reg<N, 32>(1);
    reg<N, 64>(2);
    if constexpr (N + 1 <= 8) reg_fast<N, 8>(3);
    if constexpr (N + 1 <= 16) reg_fast<N, 16>(0);
    if constexpr (N + 1 <= 32) reg_fast<N, 32>(1);
    reg_fast<N, 64>(2);
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
