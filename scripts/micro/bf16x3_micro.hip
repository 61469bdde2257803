// Exploration for the next round: can the 256-wide fp32 MLP layers run on the bf16 matrix cores at fp32
// accuracy?  x = x1 + x2 + x3 (three bf16 pieces, 24 mantissa bits), w likewise (split once at pack time); the six
// products x1w1, x1w2, x2w1, x1w3, x2w2, x3w1 accumulated in fp32 by v_mfma_f32_32x32x16_bf16 reproduce the fp32 GEMM to
// 1.1e-7 relative (scripts/micro/bf16_split_accuracy.py) at 6/16 of the fp32-MFMA instruction time.  This kernel is a
// layer chain shaped like csrc/tile.h's (TM points x 256 -> 256, weights streamed from L2, activations split and
// written back to LDS in the epilogue) to measure what that buys on MI355X including the split's VALU cost.
// Build: hipcc --offload-arch=gfx950 -O3 -DTM=64 bf16x3_micro.hip -o bf16x3_micro_64   (TM = 32 or 64)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#ifndef TM
#define TM 64
#endif
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
constexpr int RT = TM / 32;          // 32-point column tiles of the transposed product per wave
constexpr int LDB = 256 + 8;         // bf16 row stride of an activation piece in LDS
constexpr int NP = 3;

__device__ __forceinline__ float softplus_fast(float z) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(z) * (100.f * 1.44269504f));
    const float l = __builtin_amdgcn_logf(1.f + e);
    return fmaf(l, 0.0069314718f, fmaxf(z, 0.f));
}

// MODE 0: MFMA loop only; 1: + barrier and split + LDS write-back; 2: + softplus epilogue.  PIECES 3 (6 MFMAs) or 2 (3 MFMAs)
template <int MODE, int PIECES>
__global__ __launch_bounds__(256, TM == 64 ? 1 : 2) void k(const bf16x8* __restrict__ wp, float* out, int layers, int tiles) {
    __shared__ __attribute__((aligned(16))) __bf16 X[NP][TM * LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 31, kg = lane >> 5;
    for (int i = tid; i < NP * TM * LDB; i += 256) (&X[0][0])[i] = (__bf16)((float)((i * 2654435761u) >> 22) * 1e-3f);
    __syncthreads();
    f32x16 acc[2][RT];
    for (int a = 0; a < 2; ++a) for (int r = 0; r < RT; ++r) for (int q = 0; q < 16; ++q) acc[a][r][q] = 0.f;
    for (int t = 0; t < tiles; ++t)
        for (int l = 0; l < layers; ++l) {
            // W pieces packed as A operands: [layer][piece][kc 16][feature tile 8][lane 64] bf16x8
            const bf16x8* wl = wp + (size_t)(l % 8) * NP * 16 * 8 * 64 + lane;
#pragma unroll 2
            for (int kc = 0; kc < 16; ++kc) {
                bf16x8 w[2][PIECES], x[RT][PIECES];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) w[a][p] = wl[((p * 16 + kc) * 8 + (2 * wave + a)) * 64];
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int p = 0; p < PIECES; ++p)
                        x[r][p] = *reinterpret_cast<const bf16x8*>(&X[p][(r * 32 + j) * LDB + kc * 16 + 8 * kg]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        if (PIECES == 3) {
                            acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][2], x[r][0], acc[a][r], 0, 0, 0);
                            acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][1], x[r][1], acc[a][r], 0, 0, 0);
                            acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][0], x[r][2], acc[a][r], 0, 0, 0);
                        }
                        acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][1], x[r][0], acc[a][r], 0, 0, 0);
                        acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][0], x[r][1], acc[a][r], 0, 0, 0);
                        acc[a][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[a][0], x[r][0], acc[a][r], 0, 0, 0);
                    }
            }
            if (MODE >= 1) {
                __syncthreads();
                // acc[a][r][q]: feature = (2 wave + a) * 32 + 8 (q/4) + 4 kg + q%4, point = r * 32 + j
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 pc[NP];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = acc[a][r][4 * g + e] * 1e-3f;
                                if (MODE >= 2) v = softplus_fast(v);
                                float rem = v;
#pragma unroll
                                for (int p = 0; p < NP; ++p) {
                                    const __bf16 h = (__bf16)rem;
                                    pc[p][e] = h;
                                    rem -= (float)h;
                                }
                                acc[a][r][4 * g + e] = 0.f;
                            }
                            const int f0 = (2 * wave + a) * 32 + 8 * g + 4 * kg;
#pragma unroll
                            for (int p = 0; p < PIECES; ++p)
                                *reinterpret_cast<bf16x4*>(&X[p][(r * 32 + j) * LDB + f0]) = pc[p];
                        }
                __syncthreads();
            }
        }
    float s = 0.f;
    for (int a = 0; a < 2; ++a) for (int r = 0; r < RT; ++r) for (int q = 0; q < 16; ++q) s += acc[a][r][q];
    s += (float)X[0][tid];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int PIECES>
void run(const char* name, const bf16x8* wp, float* out, int grid) {
    const int layers = 8, tiles = 16;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, PIECES>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles);
    hipDeviceSynchronize();
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, PIECES>), dim3(grid), dim3(256), 0, 0, wp, out, layers, tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = 2.0 * TM * 256 * 256 * layers * tiles * grid;      // fp32-equivalent work
    printf("TM=%d %-44s %8.3f ms  %7.1f TFLOP/s fp32-equivalent (%.0f %% of the 157.3 fp32-MFMA peak)\n", TM, name, ms,
           flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100.0);
}

int main() {
    bf16x8* wp; float* out;
    const size_t nw = (size_t)8 * NP * 16 * 8 * 64;
    hipMalloc(&wp, nw * sizeof(bf16x8));
    std::vector<unsigned short> h(nw * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (i * 7919u) % 512);   // small positive bf16 values
    hipMemcpy(wp, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int grid = 256 * (TM == 64 ? 1 : 2) * 4;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    run<0, 3>("6-MFMA split, MFMA loop only", wp, out, grid);
    run<1, 3>("6-MFMA split + split/LDS write-back", wp, out, grid);
    run<2, 3>("6-MFMA split + softplus + split/write-back", wp, out, grid);
    run<0, 2>("3-MFMA split (2 pieces), MFMA loop only", wp, out, grid);
    run<2, 2>("3-MFMA split + softplus + split/write-back", wp, out, grid);
    return 0;
}
