// Development micro (round 5): how fast can ONE wave per SIMD issue v_mfma_f32_32x32x16_f16 when its accumulators live in arch VGPRs
// (what hipcc generates under __launch_bounds__(256, 2): 256 registers per wave) against AGPRs (inline asm, "a" constraint)?
// The tile-resident chains' phase stamps show a GEMM phase of 65 cycles per MFMA for a lone wave (peak: 32) whatever the prefetch depth.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_issue_micro mfma_issue_micro.hip ; ./mfma_issue_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void vgpr_form(const f16x8* in, float* out, long long* cyc, int iters) {
    f16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void agpr_form(const f16x8* in, float* out, long long* cyc, int iters) {
    f16x8 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// the tile-resident chains' k-chunk: A fragments (two m-tiles x hi / lo) from an LDS piece-plane image one chunk ahead, B fragments
// (two n-tiles x hi / lo) constant (LDSA) or streamed from global memory two chunks ahead (LDSA_GLB); 12 MFMAs product-major
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 264, PLANE = 64 * LDH;
#ifndef SYNC_EVERY
#define SYNC_EVERY 16
#endif
template <bool GLB, bool PRIO = false, bool L1HIT = false>
__global__ __launch_bounds__(256, 2) void chunk_like(const f16x8* in, float* out, long long* cyc, int iters, const u32x4* wts) {
    __shared__ __attribute__((aligned(16))) _Float16 img[2 * PLANE];
    for (int i = threadIdx.x; i < 2 * PLANE; i += 256) img[i] = (_Float16)((i * 7 % 13) * 0.01f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const u32x4* wp = wts + (2 * wave) * 2 * 64 + lane;
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    u32x4 a[2][2][2], b[4][2][2];
    auto loada = [&](u32x4 (&x)[2][2], int kc) {
        for (int m = 0; m < 2; ++m) for (int p = 0; p < 2; ++p) x[m][p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE + m * 32 * LDH + (kc & 15) * 16);
    };
    auto loadb = [&](u32x4 (&x)[2][2], int kc) {
        for (int t = 0; t < 2; ++t) for (int p = 0; p < 2; ++p) x[t][p] = GLB ? wp[(L1HIT ? 0 : (kc & 15) * 1024) + (t * 2 + p) * 64] : wp[(t * 2 + p) * 64];
    };
    loadb(b[0], 0); loadb(b[1], 1); loadb(b[2], 2); loada(a[0], 0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (GLB) loadb(b[(u + 3) & 3], it + u + 3);
            loada(a[(u + 1) & 1], it + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int pa[3] = {0, 1, 0}, pb[3] = {1, 0, 0};
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (PRIO) __builtin_amdgcn_s_setprio(1);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u & 1][m][pa[pr]]), __builtin_bit_cast(f16x8, b[u & 3][t][pb[pr]]), acc[m][t], 0, 0, 0);
                        if (PRIO) __builtin_amdgcn_s_setprio(0);
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[m][t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    (void)in;
}
// two tile groups in ONE 8-wave workgroup, in phase: waves w and w + 4 stream the SAME weight chunk at the same time (SHARE) -- does
// the CU's vector L1 merge the two requests, so that the weights cross the L2 -> CU path once per 128 points? -- or streams of their
// own (!SHARE: the two-workgroup case in one workgroup)
template <bool SHARE, int LAG = 0>
__global__ __launch_bounds__(512, 1) void chunk_paired(float* out, long long* cyc, int iters, const u32x4* wts) {
    __shared__ __attribute__((aligned(16))) _Float16 img[2 * PLANE];
    for (int i = threadIdx.x; i < 2 * PLANE; i += 512) img[i] = (_Float16)((i * 7 % 13) * 0.01f);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8;
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const u32x4* wp = wts + (2 * wave) * 2 * 64 + lane + (SHARE ? 0 : grp * 16 * 1024);
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    u32x4 a[2][2][2], b[4][2][2];
    auto loada = [&](u32x4 (&x)[2][2], int kc) {
        for (int m = 0; m < 2; ++m) for (int p = 0; p < 2; ++p) x[m][p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE + m * 32 * LDH + (kc & 15) * 16);
    };
    auto loadb = [&](u32x4 (&x)[2][2], int kc) {
        for (int t = 0; t < 2; ++t) for (int p = 0; p < 2; ++p) x[t][p] = wp[(kc & 15) * 1024 + (t * 2 + p) * 64];
    };
    // LAG: group 1 prefetches LAG chunks later than group 0 (3 - LAG chunks ahead), so that its loads find group 0's lines in L1
    const int ahead = 3 - (grp ? LAG : 0);
    loadb(b[0], 0); loadb(b[1], 1); loadb(b[2], 2); loada(a[0], 0);
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (LAG == 0 || grp == 0) loadb(b[(u + 3) & 3], it + u + 3);
            else loadb(b[(u + 3 - LAG) & 3], it + u + 3 - LAG);
            loada(a[(u + 1) & 1], it + u + 1);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int pa[3] = {0, 1, 0}, pb[3] = {1, 0, 0};
#pragma unroll
            for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[u & 1][m][pa[pr]]), __builtin_bit_cast(f16x8, b[u & 3][t][pb[pr]]), acc[m][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SYNC_EVERY && (it & (SYNC_EVERY - 1)) == 0) __syncthreads();     // a layer boundary every SYNC_EVERY chunks keeps the groups in phase
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[m][t][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <bool SHARE, int LAG = 0>
static void run_paired(const char* name, int grid) {
    float* out; long long* cyc; u32x4* wts;
    hipMalloc(&out, grid * 512 * 4); hipMalloc(&cyc, grid * 8); hipMalloc(&wts, 2 * 16 * 1024 * 16 + 65536);
    hipMemset(wts, 0, 2 * 16 * 1024 * 16 + 65536);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chunk_paired<SHARE, LAG>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, (const u32x4*)wts); hipDeviceSynchronize(); }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double m = 0; for (long long v : h) m += v; m /= grid;
    printf("%-56s grid %3d: %.1f cycles per MFMA per wave (%.0f per 12-MFMA chunk)\n", name, grid, m / (double)(iters * 12), m / iters);
    hipFree(out); hipFree(cyc); hipFree(wts);
}
template <bool GLB, bool PRIO = false, bool L1HIT = false>
static void run_chunk(const char* name, int grid) {
    f16x8* in; float* out; long long* cyc; u32x4* wts;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8); hipMalloc(&wts, 16 * 1024 * 16 + 65536);
    hipMemset(wts, 0, 16 * 1024 * 16 + 65536);
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((chunk_like<GLB, PRIO, L1HIT>), dim3(grid), dim3(256), 0, 0, (const f16x8*)in, out, cyc, iters, (const u32x4*)wts); hipDeviceSynchronize(); }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double m = 0; for (long long v : h) m += v; m /= grid;
    printf("%-56s grid %3d: %.1f cycles per MFMA per wave (%.0f per 12-MFMA chunk)\n", name, grid, m / (double)(iters * 12), m / iters);
    hipFree(in); hipFree(out); hipFree(cyc); hipFree(wts);
}
template <class K>
static void run(const char* name, K kern, int grid, int nacc) {
    f16x8* in; float* out; long long* cyc;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
    hipMemset(in, 0, 512 * 16);
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, (const f16x8*)in, out, cyc, iters);
        hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double m = 0; for (long long v : h) m += v; m /= grid;
    // counter ticks of the timed loop / event time of the launch: the rate of the counter __builtin_readcyclecounter reads (a lower
    // bound: the launch also holds the prologue and the epilogue)
    printf("%-56s grid %3d: %.1f cycles per MFMA per wave   [%.0f ticks in %.3f ms: counter >= %.2f GHz]\n", name, grid, m / (double)(iters * 3 * nacc), m, ms, m / (ms * 1e6));
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main() {
    run("VGPR-form builtin, 4 accumulators, 1 wave per SIMD", vgpr_form<4>, 256, 4);
    run("VGPR-form builtin, 4 accumulators, 2 waves per SIMD", vgpr_form<4>, 512, 4);
    run("AGPR-form asm,     4 accumulators, 1 wave per SIMD", agpr_form<4>, 256, 4);
    run("AGPR-form asm,     4 accumulators, 2 waves per SIMD", agpr_form<4>, 512, 4);
    run("VGPR-form builtin, 8 accumulators, 1 wave per SIMD", vgpr_form<8>, 256, 8);
    run("AGPR-form asm,     8 accumulators, 1 wave per SIMD", agpr_form<8>, 256, 8);
    run_chunk<false>("chain chunk: A from LDS, B constant, 1 wave per SIMD", 256);
    run_chunk<false>("chain chunk: A from LDS, B constant, 2 waves per SIMD", 512);
    run_chunk<true>("chain chunk: A from LDS, B from L2,   1 wave per SIMD", 256);
    run_chunk<true>("chain chunk: A from LDS, B from L2,   2 waves per SIMD", 512);
    run_chunk<true, true>("the same + s_setprio 1 / 0 around every MFMA, 1 wave", 256);
    run_chunk<true, true>("the same + s_setprio 1 / 0 around every MFMA, 2 waves", 512);
    run_chunk<true, false, true>("B re-loaded every chunk from ONE 16-KB block (L1 hits), 1 wave", 256);
    run_chunk<true, false, true>("B re-loaded every chunk from ONE 16-KB block (L1 hits), 2 waves", 512);
    run_paired<false>("paired groups in one workgroup, streams of their own", 256);
    run_paired<true>("paired groups in one workgroup, ONE shared stream", 256);
    run_paired<true, 1>("ONE shared stream, group 1 prefetches 1 chunk later", 256);
    run_paired<true, 2>("ONE shared stream, group 1 prefetches 2 chunks later", 256);
    run_paired<false, 2>("streams of their own, group 1 prefetches 2 chunks later", 256);
    return 0;
}
