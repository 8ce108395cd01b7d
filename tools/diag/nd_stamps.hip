// diagnostic: run the N-D kernel with phase stamps on a synthetic well-conditioned problem read from a .bin dump
#ifndef NO_STAMPS
#define MFS_ND_STAMPS
#endif
#include "../../mfs_amd/csrc/filternd_kernel.hpp"
#include <cstdio>
#include <vector>
#include <cstdlib>
using namespace mfs;
template <int TK>
int run(int argc, char** argv) {
    constexpr int N = 6;
    FILE* f = fopen(argv[1], "rb");
    int hdr[6]; fread(hdr, 4, 6, f);  // T, B, D, n_terms, z, s
    int T = hdr[0], B = hdr[1], D = hdr[2], nt = hdr[3], z = hdr[4], s = hdr[5];
    std::vector<double> coef(nd_rows<TK>() * D * D), lik(4), m0(z), mean0(2), ys((size_t)B * T);
    std::vector<int> inds(3 * s * s);
    fread(coef.data(), 8, coef.size(), f); fread(lik.data(), 8, 4, f); fread(inds.data(), 4, inds.size(), f);
    fread(m0.data(), 8, z, f); fread(mean0.data(), 8, 2, f); fread(ys.data(), 8, ys.size(), f); fclose(f);
    if (argc > 3) {   // timing run: tile the measurement rows up to the requested batch
        const int B2 = atoi(argv[3]);
        std::vector<double> y2((size_t)B2 * T);
        for (int b = 0; b < B2; ++b) for (int t = 0; t < T; ++t) y2[(size_t)b * T + t] = ys[(size_t)(b % B) * T + t];
        ys.swap(y2); B = B2;
    }
    double *dc, *dl, *dm, *dmean, *dys, *dnell, *dmeans; int* di;
    hipMalloc(&dc, coef.size() * 8); hipMalloc(&dl, 32); hipMalloc(&dm, z * 8); hipMalloc(&dmean, 16);
    hipMalloc(&dys, ys.size() * 8); hipMalloc(&dnell, B * 8); hipMalloc(&di, inds.size() * 4); hipMalloc(&dmeans, (size_t)B * T * 16);
    hipMemcpy(dc, coef.data(), coef.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dl, lik.data(), 32, hipMemcpyHostToDevice);
    hipMemcpy(dm, m0.data(), z * 8, hipMemcpyHostToDevice); hipMemcpy(dmean, mean0.data(), 16, hipMemcpyHostToDevice);
    hipMemcpy(dys, ys.data(), ys.size() * 8, hipMemcpyHostToDevice); hipMemcpy(di, inds.data(), inds.size() * 4, hipMemcpyHostToDevice);
    FilterNdArgs a{}; a.mode = 1; a.T = T; a.t_begin = 0; a.t_end = T; a.B = B; a.stable = 0; a.n_terms_used = nt; a.D = D; a.n_factors = 1; a.ny = 1;
    a.fac_kind[0] = 0; a.fac_comp[0] = 0; a.fac_ycol[0] = 0; for (int k = 0; k < nd_rows<TK>(); ++k) { int ea = 0, eb = 0; for (int i = 0; i < D; ++i) for (int j = 0; j < D; ++j) if (coef[(size_t)k * D * D + i * D + j] != 0.0) { if (i + 1 > ea) ea = i + 1; if (j + 1 > eb) eb = j + 1; } a.ext[k] = ea == 0 ? 0 : (ea | (eb << 8)); } a.coef = dc; a.lik = dl; a.inds = di; a.m0 = dm; a.m0_batched = 0; a.mean0 = dmean; a.ys = dys;
    a.out_mom = nullptr; a.out_mean = dmeans; a.out_nell = dnell; a.out_first_nan = nullptr;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&filternd_kernel<N, TK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    constexpr int lds = NdTile<N, TK>::kDoubles * 8;
    hipLaunchKernelGGL((filternd_kernel<N, TK>), dim3(B), dim3(256), lds, 0, a);
    hipDeviceSynchronize();
    if (argc > 3 && argc <= 4) {       // (a fifth argument: phase stamps of block 0 at this batch size, i.e. with a co-resident workgroup)
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((filternd_kernel<N, TK>), dim3(B), dim3(256), lds, 0, a);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("B = %d, T = %d: %.3f ms per pass (diagnostic build), lds %d B\n", B, T, ms / 3, lds);
#ifdef MFS_ND_ROLE_DEBUG
        unsigned fb[2]; hipMemcpyFromSymbol(fb, HIP_SYMBOL(g_nd_role_fallbacks), sizeof(fb));
        printf("role fallbacks %u, workgroups whose wave 0 is not role 0: %u (of %d launched)\n", fb[0], fb[1], 4 * B);
#endif
        return 0;
    }
#ifndef MFS_ND_STAMPS
    printf("(built without stamps: timing runs only)\n"); return 0;
#else
    unsigned long long st[24]; hipMemcpyFromSymbol(st, HIP_SYMBOL(g_nd_stamps), sizeof(st));
    const char* names[] = {"gather (x2)", "cholesky (x2)", "trsm+sym (x2)", "warm-start matmuls / Chebyshev-grid rule", "jacobi sweeps", "weights", "bilinear predict", "bilinear update"};
    double tot = 0; for (int i = 0; i < 8; ++i) tot += st[i];
    printf("steps %llu, Jacobi sweeps %llu (%.2f per update rule; one matrix)\n", st[9], st[8], (double)st[8] / st[9]);
    for (int i = 0; i < 8; ++i) printf("%-24s %10.0f cycles per step  %5.1f %%\n", names[i], (double)st[i] / st[9], 100.0 * st[i] / tot);
    const char* sub[] = {"  predict: Krylov, low corner", "  predict: moment array, low corner", "  predict: means", "  predict: Krylov at new mean || re-centring", "  predict: moment array", "  update: h, K h, means, shifted powers"};
    for (int i = 0; i < 6; ++i) printf("%-36s %10.0f cycles per step\n", sub[i], (double)st[10 + i] / st[9]);
    const char* sub2[] = {"    cheb: Gershgorin", "    cheb: samples + coefficients", "    cheb: recurrence", "    cheb: K h"};
    for (int i = 0; i < 4; ++i) printf("%-36s %10.0f cycles per step\n", sub2[i], (double)st[16 + i] / st[9]);
    printf("    mean Chebyshev degree %.1f\n", (double)st[20] / st[9]);
    printf("  front end (x2): load + elimination %.0f, block solves %.0f cycles per step\n", (double)st[21] / st[9], (double)st[22] / st[9]);
    unsigned long long hist[8][40]; hipMemcpyFromSymbol(hist, HIP_SYMBOL(g_nd_hist), sizeof(hist));
    for (int t = 0; t < 8; ++t) {
        unsigned long long n = 0; for (int k = 0; k < 40; ++k) n += hist[t][k];
        if (!n) continue;
        printf("convergence test before sweep %d (%llu tests): -log10(off/dia) histogram:", t, n);
        for (int k = 0; k < 40; ++k) if (hist[t][k]) printf(" %d:%llu", k, hist[t][k]);
        printf("\n");
    }
    std::vector<double> nell(B); hipMemcpy(nell.data(), dnell, B * 8, hipMemcpyDeviceToHost); printf("nell[0] = %.10f\n", nell[0]);
    return 0;
#endif
}
int main(int argc, char** argv) { return (argc > 2 && argv[2][0] == '1') ? run<1>(argc, argv) : run<0>(argc, argv); }
