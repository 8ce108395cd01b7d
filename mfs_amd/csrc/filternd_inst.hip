// filternd_inst.hip -- instantiates the d = 2 N-D kernels for quadrature orders 2..7 and the transition families
// (TK = 0 operator table with |kappa| <= 4, TK = 3 the same with the node tables of a joint likelihood, TK = 2 with |kappa| <= 6, TK = 1 Normal closure), and registers their launchers.
#include "filternd_kernel.hpp"
#include "launch_util.hpp"

namespace mfs {

using FilterNdLaunch = hipError_t (*)(const FilterNdArgs&, int grid, hipStream_t);
struct NdEntry { FilterNdLaunch launch, launch_gauss, launch_hi, launch_joint; int S, Z, lds_bytes, carry_doubles; };
NdEntry g_nd_table[8];

template <int N, int TK>
hipError_t launch_nd(const FilterNdArgs& a, int grid, hipStream_t s) {
    constexpr int lds = NdTile<N, TK>::kDoubles * 8;
    if (hipError_t e = ensure_dynamic_lds<&filternd_kernel<N, TK>>(); e != hipSuccess) return e;
    hipLaunchKernelGGL((filternd_kernel<N, TK>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <int N>
void reg_nd() {
    constexpr int d03 = (N <= 6 && NdTile<N, 3>::kDoubles > NdTile<N, 0>::kDoubles) ? NdTile<N, 3>::kDoubles : NdTile<N, 0>::kDoubles;
    constexpr int d0 = d03, d1 = NdTile<N, 1>::kDoubles, d2 = NdTile<N, 2>::kDoubles;
    FilterNdLaunch joint = nullptr;     // operator tables + the node tables of a joint likelihood (TK = 3), N <= 6
    if constexpr (N <= 6) joint = &launch_nd<N, 3>;
    g_nd_table[N] = NdEntry{&launch_nd<N, 0>, &launch_nd<N, 1>, &launch_nd<N, 2>, joint, NdTile<N, 0>::S, NdTile<N, 0>::Z,
                            (d0 > d1 ? (d0 > d2 ? d0 : d2) : (d1 > d2 ? d1 : d2)) * 8, NdTile<N, 0>::kCarry};
    static_assert(NdTile<N, 0>::kCarry == NdTile<N, 1>::kCarry && NdTile<N, 0>::kCarry == NdTile<N, 2>::kCarry, "one carry layout");
    if constexpr (N < 7) reg_nd<N + 1>();
}

// diagnostic: the kernels' own elementary functions on an array (tests bound their error against libm)
__global__ void elementary_kernel(const int which, const int n, const double* __restrict__ x, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    out[i] = (which == 0) ? fast_exp(v) : (which == 1) ? fast_tanh(v) : fast_log(v);
}
hipError_t launch_elementary(int which, int n, const double* d_x, double* d_out, hipStream_t s) {
    hipLaunchKernelGGL(elementary_kernel, dim3((n + 255) / 256), dim3(256), 0, s, which, n, d_x, d_out);
    return hipGetLastError();
}

struct NdRegistrar { NdRegistrar() { reg_nd<2>(); } };
static NdRegistrar nd_registrar_instance;

}  // namespace mfs
