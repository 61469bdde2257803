// Microbenchmark of the tile GEMM core (tile.h gemm_rows) without epilogues: how close does the MFMA loop itself get
// to the fp32 MFMA peak, at TM=64 (2 WG/CU)?  (TM=128, 1 WG/CU, was measured in round 1 and its build switch removed.)  Build: make gemm_micro_64
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../dynhor_amd/csrc/tile.h"
using namespace dh;

__device__ __forceinline__ float softplus_fast(float z) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(z) * (100.f * 1.44269504f));
    const float l = __builtin_amdgcn_logf(1.f + e);
    return fmaf(l, 0.0069314718f, fmaxf(z, 0.f));
}

template <int MODE, int STAG>
__global__ __launch_bounds__(256, TM == 128 ? 1 : 2) void k(const f32x4* __restrict__ wp, float* out, int layers, int tiles_per_wg, const float* __restrict__ big, float* __restrict__ big2) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TM * LDX; i += 256) smain[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    if (STAG > 0 && blockIdx.x >= gridDim.x / 2) { for (int i = 0; i < STAG; ++i) __builtin_amdgcn_s_sleep(127); }
    if (STAG < 0 && (blockIdx.x & 1)) { for (int i = 0; i < -STAG; ++i) __builtin_amdgcn_s_sleep(127); }
    if (STAG == 100 && ((blockIdx.x >> 3) & 1)) { for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127); }
    f32x16 acc[MT][2];
    acc_zero(acc);
    for (int t = 0; t < tiles_per_wg; ++t)
        for (int l = 0; l < layers; ++l) {
            gemm_rows(acc, smain, LDX, 32, wp + (size_t)(l % 8) * 32 * 8 * 64, wave, lane);
            if (MODE == 4 || MODE == 7) {   // bias + relu
                DH_UNROLL for (int m = 0; m < MT; ++m) DH_UNROLL for (int t2 = 0; t2 < 2; ++t2) DH_UNROLL for (int r = 0; r < 16; ++r)
                    acc[m][t2][r] = fmaxf(acc[m][t2][r] * 1e-3f + 0.01f, 0.f);
            }
            if (MODE == 5 || MODE == 8 || MODE == 9) {   // softplus
                DH_UNROLL for (int m = 0; m < MT; ++m) DH_UNROLL for (int t2 = 0; t2 < 2; ++t2) DH_UNROLL for (int r = 0; r < 16; ++r)
                    acc[m][t2][r] = (MODE == 5) ? softplus100(acc[m][t2][r] * 1e-3f) : softplus_fast(acc[m][t2][r] * 1e-3f);
            }
            if (MODE == 6 || MODE == 8) {   // load a saved tile and multiply (reverse / tangent style)
                const f32x4* hp = reinterpret_cast<const f32x4*>(big + ((size_t)(blockIdx.x * tiles_per_wg + t) * 8 + (l % 8)) * TILE_F) + (size_t)wave * MT * 8 * 64 + lane;
                DH_UNROLL for (int m = 0; m < MT; ++m) {
                    DH_UNROLL for (int t2 = 0; t2 < 2; ++t2) DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = hp[((m * 2 + t2) * 4 + r4) * 64];
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) acc[m][t2][4 * r4 + rr] *= h[rr];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (MODE == 3 || MODE == 7 || MODE == 8) acc_store_native(acc, big2 + ((size_t)(blockIdx.x * tiles_per_wg + t) * 8 + (l % 8)) * TILE_F, wave, lane);
            if (MODE >= 1) { __syncthreads(); }
            if (MODE >= 2) { acc_to_lds(acc, smain, wave, lane); __syncthreads(); }
        }
    float s = 0.f;
    for (int m = 0; m < MT; ++m) for (int t2 = 0; t2 < 2; ++t2) for (int r = 0; r < 16; ++r) s += acc[m][t2][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int STAG = 0>
void run(const char* name, const f32x4* wp, float* out, int grid, const float* big, float* big2) {
    const int layers = 8, tiles = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, STAG>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles, big, big2);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, STAG>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles, big, big2);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double flop = 2.0 * grid * (double)tiles * layers * TM * 256.0 * 256.0;
    printf("TM=%d %-28s grid=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", TM, name, grid, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    f32x4* wp; float* out;
    hipMalloc(&wp, 8 * 256 * 256 * 4); hipMalloc(&out, 1024 * 256 * 4);
    std::vector<float> h(8 * 256 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2246822519u) >> 22) * 1e-4f - 0.05f;
    hipMemcpy(wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int grid = 256 * (TM == 128 ? 1 : 2);
    float *big, *big2;
    const size_t nb = (size_t)grid * 16 * 8 * TILE_F;
    (void)hipMalloc(&big, nb * 4); (void)hipMalloc(&big2, nb * 4);
    (void)hipMemset(big, 0, nb * 4);
    run<2>("gemm + lds handoff", wp, out, grid, big, big2);
    run<3>("handoff + store tile", wp, out, grid, big, big2);
    run<4>("handoff + relu", wp, out, grid, big, big2);
    run<5>("handoff + softplus", wp, out, grid, big, big2);
    run<6>("handoff + load tile", wp, out, grid, big, big2);
    run<7>("handoff + relu + store", wp, out, grid, big, big2);
    run<8>("fast softplus + load + store", wp, out, grid, big, big2);
    run<9>("handoff + fast softplus", wp, out, grid, big, big2);
    run<5, 4>("softplus, stagger 4", wp, out, grid, big, big2);
    run<8, 2>("fast sp + ld + st, stagger 2", wp, out, grid, big, big2);
    run<8, 4>("fast sp + ld + st, stagger 4", wp, out, grid, big, big2);
    run<8, 8>("fast sp + ld + st, stagger 8", wp, out, grid, big, big2);
    run<5, -4>("softplus, parity stagger 4", wp, out, grid, big, big2);
    run<8, -4>("fast sp+ld+st, parity stagger 4", wp, out, grid, big, big2);
    run<5, 100>("softplus, (b>>3)&1 stagger", wp, out, grid, big, big2);
    run<8, 100>("fast sp+ld+st, (b>>3)&1 stag", wp, out, grid, big, big2);
    return 0;
}
