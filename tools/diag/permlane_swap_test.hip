#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int l = threadIdx.x;
    int x = 100 + l, y = 100 + l;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    out[l] = r[0]; out[64 + l] = r[1];
    auto q = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[128 + l] = q[0]; out[192 + l] = q[1];
}
int main() {
    int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int s = 0; s < 4; ++s) { printf("%d:", s); for (int i = 0; i < 64; i += 4) printf(" %d", h[64 * s + i]); printf("\n"); }
    return 0;
}
