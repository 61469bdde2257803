// Microbenchmark of the weight-gradient GEMM loop (dw.hip): HBM-streamed vs cache-resident operands.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#define DW_MICRO 1
#include "../../dynhor_amd/csrc/dw.hip"
using namespace dh;

int main() {
    const int64_t nt = 4096;
    const int njobs = 10;
    float *A, *B, *slabs;
    const size_t arr = (size_t)nt * TILE_F;
    hipMalloc(&A, arr * 4 * njobs); hipMalloc(&B, arr * 4 * njobs);
    hipMemset(A, 0, arr * 4 * njobs); hipMemset(B, 0, arr * 4 * njobs);
    const int64_t gstride = (int64_t)njobs * 8 * 8 * 1024;
    hipMalloc(&slabs, (size_t)256 * gstride * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int variant = 0; variant < 3; ++variant) {
        DwJobs J{};
        for (int j = 0; j < njobs; ++j) {
            const size_t off = variant == 0 ? (size_t)j * arr : 0;
            J.j[j].A1 = A + off; J.j[j].B1 = B + off; J.j[j].A2 = nullptr; J.j[j].B2 = nullptr; J.j[j].nb = 8;
            J.j[j].off = (int64_t)j * 8 * 8 * 1024;
        }
        J.n = njobs;
        const int64_t mask = variant == 2 ? 15 : -1;
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(dw_kernel, dim3(256), dim3(512), 0, 0, J, nt, slabs, gstride, mask);
        hipEventRecord(a);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(dw_kernel, dim3(256), dim3(512), 0, 0, J, nt, slabs, gstride, mask);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
        const double flop = 2.0 * njobs * (double)nt * TM * 65536.0;
        const char* names[3] = {"distinct arrays per job (HBM stream)", "same arrays for all jobs (L3 reuse)", "16 tiles only (L2 resident)"};
        printf("dw %-40s %.3f ms  %.1f TFLOP/s (%.1f%%)\n", names[variant], ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    }
    return 0;
}
