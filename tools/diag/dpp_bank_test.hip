// diagnostic: do bank-masked row_newbcast DPP operations behave as documented for 64-bit operands on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
    const int l = threadIdx.x;
    double v = 100.0 + l, u = 1.0, acc = 0.0;
    double r = __builtin_amdgcn_update_dpp(0.0, v, 0x150 + 3, 0xf, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, v, 0x150 + 8 + 3, 0xf, 0xc, false);
    out[l] = r;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:3 row_mask:0xf bank_mask:0x3\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(v), "v"(u));
    out[64 + l] = acc;
    double acc2 = 0.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, -%2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc2) : "v"(v), "v"(u));
    out[128 + l] = acc2;
}
int main() {
    double* d; hipMalloc(&d, 192 * 8); k<<<1, 64>>>(d); double h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mov  :"); for (int i = 0; i < 32; ++i) printf(" %g", h[i]); printf("\n");
    printf("fmac :"); for (int i = 0; i < 32; ++i) printf(" %g", h[64 + i]); printf("\n");
    printf("fmacF:"); for (int i = 0; i < 32; ++i) printf(" %g", h[128 + i]); printf("\n");
    return 0;
}
