// launch_util.hpp -- host-side helpers shared by the launcher translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>

namespace mfs {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per (kernel, device).  One atomic device mask per kernel
// instantiation (the template parameter makes the static unique): a bit is published only after the call succeeded on
// that device, two threads racing on a fresh (kernel, device) both make the (idempotent) call, nobody launches before
// the attribute is set.  Devices >= 64 simply set it on every launch.
template <auto Kernel>
inline hipError_t ensure_dynamic_lds(int bytes = 160 * 1024) {
    static std::atomic<uint64_t> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = (dev >= 0 && dev < 64) ? (uint64_t{1} << dev) : 0;
    if (bit && (done.load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess && bit) done.fetch_or(bit, std::memory_order_release);
    return e;
}

}  // namespace mfs
