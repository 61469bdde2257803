// Variant of bf16x3_micro.hip: the LDS activation image stays fp32 (66.5 KB -> two workgroups per CU, as in tile.h) and
// every wave splits its A fragments into bf16 pieces on the fly (4x redundant VALU across the workgroup's waves), hoping
// the split hides in the MFMA issue shadow (24 of every 32 cycles are free for the VALU) of the two waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 bf16x3_fp32lds_micro.hip -o bf16x3_fp32lds_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TM = 64, RT = 2, LDX = 260, NP = 3;

struct Bf3 { bf16x8 p[3]; };
__device__ __forceinline__ Bf3 split3(const f32x4& lo, const f32x4& hi) {
    Bf3 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = e < 4 ? lo[e] : hi[e - 4];
        const __bf16 h1 = (__bf16)v;
        const float r1 = v - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        r.p[0][e] = h1; r.p[1][e] = h2; r.p[2][e] = (__bf16)(r1 - (float)h2);
    }
    return r;
}
__device__ __forceinline__ float softplus_fast(float z) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(z) * (100.f * 1.44269504f));
    const float l = __builtin_amdgcn_logf(1.f + e);
    return fmaf(l, 0.0069314718f, fmaxf(z, 0.f));
}

// MODE 0: GEMM loop (with on-the-fly split) only; 1: + softplus + fp32 LDS write-back.  PIN: 1 = sched_barrier between the
// fetch/split block and the MFMA block, 0 = leave the interleaving to the compiler
template <int MODE, int PIN>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ wp, float* out, int layers, int tiles) {
    __shared__ __attribute__((aligned(16))) float X[TM * LDX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    for (int i = tid; i < TM * LDX; i += 256) X[i] = (float)((i * 2654435761u) >> 22) * 1e-3f;
    __syncthreads();
    f32x16 acc[RT][2];
    for (int r = 0; r < RT; ++r) for (int t = 0; t < 2; ++t) for (int q = 0; q < 16; ++q) acc[r][t][q] = 0.f;
    for (int tl = 0; tl < tiles; ++tl)
        for (int l = 0; l < layers; ++l) {
            const bf16x8* wl = wp + (size_t)(l % 8) * NP * 16 * 8 * 64 + lane;
            const float* xrow = X + j * LDX + 8 * kg;
            Bf3 a[RT], an[RT], b[2], bn[2];
            auto fetch = [&](Bf3 (&aa)[RT], Bf3 (&bb)[2], int kc) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bb[t].p[p] = wl[((p * 16 + kc) * 8 + (2 * wave + t)) * 64];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + r * 32 * LDX + kc * 16);
                    const f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + r * 32 * LDX + kc * 16 + 4);
                    aa[r] = split3(lo, hi);
                }
            };
            auto mul = [&](const Bf3 (&aa)[RT], const Bf3 (&bb)[2]) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[2], bb[t].p[0], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[1], bb[t].p[1], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[0], bb[t].p[2], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[1], bb[t].p[0], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[0], bb[t].p[1], acc[r][t], 0, 0, 0);
                        acc[r][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[r].p[0], bb[t].p[0], acc[r][t], 0, 0, 0);
                    }
            };
            fetch(a, b, 0);
#pragma unroll 1
            for (int kc = 0; kc < 16; kc += 2) {
                fetch(an, bn, kc + 1);
                if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
                mul(a, b);
                if (PIN == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);      // DS reads of the next A fragments
                    __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);      // weight pieces
#pragma unroll
                    for (int q = 0; q < 24; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);  // four VALU of the split
                    }
                }
                if (PIN) __builtin_amdgcn_sched_barrier(0);
                fetch(a, b, kc + 2 < 16 ? kc + 2 : 15);
                if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
                mul(an, bn);
                if (PIN == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
                    for (int q = 0; q < 24; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                }
                if (PIN) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 1) {
                __syncthreads();
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int row = r * 32 + 8 * (q >> 2) + 4 * kg + (q & 3), col = 64 * wave + 32 * t + j;
                            X[row * LDX + col] = softplus_fast(acc[r][t][q] * 1e-3f);
                            acc[r][t][q] = 0.f;
                        }
                __syncthreads();
            }
        }
    float s = 0.f;
    for (int r = 0; r < RT; ++r) for (int t = 0; t < 2; ++t) for (int q = 0; q < 16; ++q) s += acc[r][t][q];
    out[blockIdx.x * 256 + tid] = s + X[tid];
}

template <int MODE, int PIN>
void run(const char* name, const bf16x8* wp, float* out, int grid) {
    const int layers = 8, tiles = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PIN>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, PIN>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = 2.0 * TM * 256 * 256 * layers * tiles * grid;
    printf("fp32-LDS split-on-fetch %-40s %8.3f ms  %7.1f TFLOP/s fp32-equivalent\n", name, ms, flop / ms / 1e9);
}

int main() {
    bf16x8* wp; float* out;
    const size_t nw = (size_t)8 * NP * 16 * 8 * 64;
    hipMalloc(&wp, nw * sizeof(bf16x8));
    std::vector<unsigned short> h(nw * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 7919u) % 512);
    hipMemcpy(wp, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int grid = 256 * 2 * 4;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    run<0, 1>("GEMM loop only, phases pinned", wp, out, grid);
    run<0, 0>("GEMM loop only, compiler-interleaved", wp, out, grid);
    run<1, 1>("+ softplus + write-back, pinned", wp, out, grid);
    run<1, 0>("+ softplus + write-back, compiler-interleaved", wp, out, grid);
    run<0, 2>("GEMM loop only, sched_group_barrier 1 MFMA : 4 VALU", wp, out, grid);
    run<1, 2>("+ softplus + write-back, sched_group_barrier", wp, out, grid);
    return 0;
}
