// diagnostic: N = 7 (8-lane groups) build of the 1-D kernel with phase stamps (never part of libmfs_hip.so)
#define MFS_1D_STAMPS
#define MFS_NLO 7
#define MFS_NHI 7
#include "../../mfs_amd/csrc/filter1d_inst.hip"
extern "C" int mfs_debug_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mfs::g_1d_stamps), 16 * sizeof(unsigned long long));
}
