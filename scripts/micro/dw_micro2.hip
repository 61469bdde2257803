// dW loop structure experiments: loads on/off, 8 waves x 1 WG/CU vs 4 waves x 2 WG/CU, 2x4 vs 1x8 wave tiles.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../dynhor_amd/csrc/tile.h"
using namespace dh;

template <int NA, int NBW> struct Ops { f32x4 a[NA]; f32x4 b[NBW]; };

template <int NA, int NBW>
__device__ __forceinline__ void ld(Ops<NA, NBW>& o, const f32x4* ap, const f32x4* bp, int kq) {
    DH_UNROLL for (int i = 0; i < NA; ++i) o.a[i] = ap[(kq * NA + i) * 64];
    DH_UNROLL for (int j = 0; j < NBW; ++j) o.b[j] = bp[(kq * NBW + j) * 64];
}
template <int NA, int NBW>
__device__ __forceinline__ void mm(f32x16 (&acc)[NA][NBW], const Ops<NA, NBW>& o) {
    DH_UNROLL for (int rr = 0; rr < 4; ++rr) DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i][rr], o.b[j][rr], acc[i][j], 0, 0, 0);
}

template <int NA, int NBW, int LOADS, int THREADS, int STREAM = 0>
__global__ __launch_bounds__(THREADS, 2) void k(const float* __restrict__ A, const float* __restrict__ B, float* out, int iters) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[NA][NBW];
    DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const f32x4* ap = reinterpret_cast<const f32x4*>(A) + (size_t)(blockIdx.x * 8 + wave) * 64 * 64 + lane;
    const f32x4* bp = reinterpret_cast<const f32x4*>(B) + (size_t)(blockIdx.x * 8 + wave) * 64 * 64 + lane;
    Ops<NA, NBW> s0, s1;
    ld(s0, ap, bp, 0); ld(s1, ap, bp, 1);
    for (int it = 0; it < iters; ++it) {
        if (STREAM == 1) { ap += 8 * 8 * 64 * 64 * (512 / THREADS); bp += 8 * 8 * 64 * 64 * (512 / THREADS); }   // private stream
        DH_UNROLL for (int kq = 0; kq < 8; kq += 2) {
            if (LOADS) ld(s1, ap, bp, (kq + 1) & 7);
            __builtin_amdgcn_sched_barrier(0);
            mm(acc, s0);
            __builtin_amdgcn_sched_barrier(0);
            if (LOADS) ld(s0, ap, bp, (kq + 2) & 7);
            __builtin_amdgcn_sched_barrier(0);
            mm(acc, s1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
}

template <int NA, int NBW, int LOADS, int THREADS, int STREAM = 0>
void run(const char* name, const float* A, const float* B, float* out) {
    const int grid = 256 * (512 / THREADS), iters = 256;
    (void)0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NA, NBW, LOADS, THREADS, STREAM>), dim3(grid), dim3(THREADS), 0, 0, A, B, out, iters);
    hipEventRecord(a);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<NA, NBW, LOADS, THREADS, STREAM>), dim3(grid), dim3(THREADS), 0, 0, A, B, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 3;
    const double flop = 2.0 * grid * (THREADS / 64) * (double)iters * 8 * 4 * NA * NBW * 2048.0;
    printf("%-44s %.3f ms  %.1f TFLOP/s (%.1f%%)\n", name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *A, *B, *out;
    const size_t n = (size_t)512 * 8 * 64 * 64 * 4 * 260 + 65536;
    hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&out, 1 << 22);
    hipMemset(A, 0, n * 4); hipMemset(B, 0, n * 4);
    run<2, 4, 0, 512>("2x4, no loads, 8 waves x 1 WG/CU", A, B, out);
    run<2, 4, 1, 512>("2x4, loads,    8 waves x 1 WG/CU", A, B, out);
    run<2, 4, 0, 256>("2x4, no loads, 4 waves x 2 WG/CU", A, B, out);
    run<2, 4, 1, 256>("2x4, loads,    4 waves x 2 WG/CU", A, B, out);
    run<2, 4, 1, 512, 1>("2x4, loads STREAMED, 8 waves x 1 WG/CU", A, B, out);
    run<2, 4, 1, 256, 1>("2x4, loads STREAMED, 4 waves x 2 WG/CU", A, B, out);
    run<1, 8, 1, 512>("1x8, loads,    8 waves x 1 WG/CU", A, B, out);
    run<2, 2, 1, 256>("2x2, loads,    4 waves x 2 WG/CU", A, B, out);
    run<2, 2, 0, 256>("2x2, no loads, 4 waves x 2 WG/CU", A, B, out);
    run<1, 2, 1, 256>("1x2, loads,    4 waves x 2 WG/CU", A, B, out);
    return 0;
}
