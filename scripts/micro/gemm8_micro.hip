// What would an 8-wave / 128-row tile chain buy?  wave = one 32-col n-tile x 4 m-tiles (B loads per MFMA halved,
// LDS A reads doubled) vs the shipping 4-wave / 64-row x 2 workgroups-per-CU layout (gemm_micro.hip).
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TMX = 128, LDXX = 260;

__global__ __launch_bounds__(512, 2) void k8(const f32x4* __restrict__ wp, float* out, int layers, int tiles, int mode) {
    extern __shared__ __attribute__((aligned(16))) float smain[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TMX * LDXX; i += 512) smain[i] = (float)((i * 2654435761u) >> 20) * 1e-4f;
    __syncthreads();
    f32x16 acc[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float* xrow = smain + (lane & 31) * LDXX + 4 * (lane >> 5);
    for (int t = 0; t < tiles; ++t)
        for (int l = 0; l < layers; ++l) {
            const f32x4* wl = wp + (size_t)(l % 8) * 32 * 8 * 64 + wave * 64 + lane;
            f32x4 a0[4], a1[4], b0, b1;
            b0 = wl[0];
            for (int m = 0; m < 4; ++m) a0[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * LDXX);
#pragma unroll 1
            for (int kg = 0; kg < 32; kg += 2) {
                b1 = wl[(kg + 1) * 8 * 64];
                for (int m = 0; m < 4; ++m) a1[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * LDXX + (kg + 1) * 8);
                __builtin_amdgcn_sched_barrier(0);
                for (int s = 0; s < 4; ++s) for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m][s], b0[s], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int k2 = kg + 2 < 32 ? kg + 2 : kg + 1;
                b0 = wl[k2 * 8 * 64];
                for (int m = 0; m < 4; ++m) a0[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * LDXX + k2 * 8);
                __builtin_amdgcn_sched_barrier(0);
                for (int s = 0; s < 4; ++s) for (int m = 0; m < 4; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m][s], b1[s], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (mode >= 1) {
                __syncthreads();
                const int w = wave >> 1, tt = wave & 1;
                for (int m = 0; m < 4; ++m) {
                    float* base = smain + (m * 32 + 4 * (lane >> 5)) * LDXX + 64 * w + 32 * tt + (lane & 31);
                    for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * LDXX] = acc[m][r];
                }
                __syncthreads();
            }
        }
    float s = 0.f;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    f32x4* wp; float* out;
    hipMalloc(&wp, 8 * 256 * 256 * 4); hipMalloc(&out, 1024 * 512 * 4);
    std::vector<float> h(8 * 256 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2246822519u) >> 22) * 1e-4f - 0.05f;
    hipMemcpy(wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, TMX * LDXX * 4);
    const int grid = 256, layers = 8, tiles = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k8, dim3(grid), dim3(512), TMX * LDXX * 4, 0, wp, out, layers, tiles, mode);
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k8, dim3(grid), dim3(512), TMX * LDXX * 4, 0, wp, out, layers, tiles, mode);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        const double flop = 2.0 * grid * (double)tiles * layers * TMX * 256.0 * 256.0;
        printf("8 waves x TM=128, %s: %.3f ms  %.1f TFLOP/s (%.1f%%)\n", mode ? "gemm + lds handoff" : "gemm only", ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
    }
    return 0;
}
