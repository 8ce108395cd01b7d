// diagnostic: where do the four waves of 256-thread workgroups land (SIMD, wave slot, CU) when two workgroups share a CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256, 2) void probe(unsigned* out, int spin) {
    extern __shared__ double lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    double v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = fma(v, 1.0000001, 1e-9);   // keep the workgroups resident together
    lds[threadIdx.x] = v;
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main() {
    const int B = 512;
    unsigned* d; hipMalloc(&d, B * 4 * 2 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(B), dim3(256), 70 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(B * 8); hipMemcpy(h.data(), d, B * 32, hipMemcpyDeviceToHost);
    for (int b : {0, 1, 2, 8, 255, 256, 257, 264, 511}) {
        printf("wg %3d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned hw = h[(b * 4 + w) * 2], x = h[(b * 4 + w) * 2 + 1];
            printf("  [slot %u simd %u cu %u sh %u se %u xcc %u]", hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7, x & 15);
        }
        printf("\n");
    }
    // statistics: per workgroup, are the four waves on four distinct SIMDs in order, and do they share one slot number?
    int inorder = 0, sameslot = 0, distinct = 0; int slotpar[2] = {0, 0};
    for (int b = 0; b < B; ++b) {
        unsigned s[4], sl[4]; for (int w = 0; w < 4; ++w) { s[w] = (h[(b * 4 + w) * 2] >> 4) & 3; sl[w] = h[(b * 4 + w) * 2] & 15; }
        distinct += ((1u << s[0]) | (1u << s[1]) | (1u << s[2]) | (1u << s[3])) == 15;
        inorder += (s[1] == (s[0] + 1) % 4) && (s[2] == (s[0] + 2) % 4) && (s[3] == (s[0] + 3) % 4);
        sameslot += (sl[0] == sl[1]) && (sl[1] == sl[2]) && (sl[2] == sl[3]);
        slotpar[sl[0] & 1]++;
    }
    printf("distinct SIMDs %d / %d, consecutive SIMDs %d, same slot on all four %d, slot parity of wave 0: even %d odd %d\n", distinct, B, inorder, sameslot, slotpar[0], slotpar[1]);
    return 0;
}
