// diagnostic: dependent-issue latency of fp64 VALU ops on one wave per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH>
__global__ void fma_chain(double* out, double a, double b, unsigned long long* cyc) {
    double x[CH];
    for (int i = 0; i < CH; ++i) x[i] = threadIdx.x * 1e-3 + i;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) x[i] = fma(x[i], a, b);
    }
    unsigned long long t1 = clock64();
    double s = 0; for (int i = 0; i < CH; ++i) s += x[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void rcp_chain(double* out, double a, unsigned long long* cyc) {
    double x = threadIdx.x * 1e-3 + 1.5;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x = __builtin_amdgcn_rcp(x) + a;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void dpp_chain(double* out, double a, unsigned long long* cyc) {
    double x = threadIdx.x * 1e-3 + 1.5;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x = __builtin_amdgcn_update_dpp(0.0, x, 0x150 + 3, 0xf, 0xf, false) * a;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void cmp_chain(double* out, double a, unsigned long long* cyc) {
    double x = threadIdx.x * 1e-3 + 1.5;
    unsigned long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) x = (x > a) ? x * 0.999 : x * 1.001;
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
#define RUN(name, call, nops) call; hipDeviceSynchronize(); call; hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.2f cycles per op\n", name, (double)h / (nops));
    RUN("fma f64, 1 chain", (fma_chain<1><<<1, 64>>>(out, 0.999, 1e-3, cyc)), 2048.0)
    RUN("fma f64, 2 chains (per op)", (fma_chain<2><<<1, 64>>>(out, 0.999, 1e-3, cyc)), 4096.0)
    RUN("fma f64, 4 chains (per op)", (fma_chain<4><<<1, 64>>>(out, 0.999, 1e-3, cyc)), 8192.0)
    RUN("fma f64, 8 chains (per op)", (fma_chain<8><<<1, 64>>>(out, 0.999, 1e-3, cyc)), 16384.0)
    RUN("rcp f64 + add, chain (pair)", (rcp_chain<<<1, 64>>>(out, 0.5, cyc)), 2048.0)
    RUN("dpp bcast + mul, chain (pair)", (dpp_chain<<<1, 64>>>(out, 0.999, cyc)), 2048.0)
    RUN("cmp + cndmask + mul (triple)", (cmp_chain<<<1, 64>>>(out, 1.0, cyc)), 2048.0)
    return 0;
}
