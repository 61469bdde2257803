// Which bf16 MFMA shape delivers more in THIS loop?  (guide: MI355X_MICROARCH.md 'DVFS give-back' item 7 -- on bare loops the chip
// holds a higher clock on v_mfma_f32_16x16x32_bf16 than on 32x32x16: 1.12-1.15 x the FLOP/s at equal cycles per FLOP.)
// Both arms: the product's layer loop -- fp32 [64 x 256] LDS image, two workgroups per CU, A fragments split into three bf16
// pieces as they are fetched (pair-wise, dealt out between the MFMAs in pinned program order), six products per fp32 product,
// packed weight pieces streamed from L2, every MFMA at raised wave priority, softplus + LDS write-back per layer, random data.
//   arm 32: dh::gemm_rows_s of dynhor_amd/csrc/tile16.h itself (wave tile 64 x 64 = 2 x 2 accumulators of 32 x 32)
//   arm 16: the same scheme on 16x16x32 (wave tile 64 x 64 = 4 x 4 accumulators of 16 x 16; a 32-deep k-chunk in two halves of
//           two n-tiles each so that the operand registers fit: 64 acc + 96 A pieces + 48 B pieces + 32 raw)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_shape_micro.hip -o mfma_shape_micro
#include "../../dynhor_amd/csrc/tile16.h"
#include <cstdio>
#include <vector>
using namespace dh;

typedef __attribute__((__vector_size__(4 * sizeof(float)))) float v4f;
struct U3q { u32x4 p[3]; };

__device__ __forceinline__ float softplus_fast(float z) { return softplus100(z); }

// ---------------------------------------------------------------- arm 16
template <int STEP>   // 48 steps: pair j = STEP / 3 of 16 (m-tile j / 4, quad j % 4), stage STEP % 3
__device__ __forceinline__ void split_step16(U3q (&a)[4], const f32x4 (&lo)[4], const f32x4 (&hi)[4], f32x2 (&r1)[16]) {
    constexpr int j = STEP / 3, s = STEP % 3, m = j / 4, q = j % 4;
    if constexpr (s == 0) {
        f32x2 x;
        x[0] = q < 2 ? lo[m][2 * q] : hi[m][2 * q - 4];
        x[1] = q < 2 ? lo[m][2 * q + 1] : hi[m][2 * q - 3];
        const unsigned h = pack_bf16x2(x);
        a[m].p[0][q] = h;
        r1[j] = x - unpack_bf16x2(h);
    } else if constexpr (s == 1) {
        const unsigned mm = pack_bf16x2(r1[j]);
        a[m].p[1][q] = mm;
        r1[j] = r1[j] - unpack_bf16x2(mm);
    } else {
        a[m].p[2][q] = pack_bf16x2(r1[j]);
    }
}
template <int I, int NB, int SBASE>     // 48 MFMAs of one half: product-major over 4 m-tiles x 2 n-tiles
__device__ __forceinline__ void half16(v4f (&acc)[4][4], const U3q (&ac)[4], const Bf3 (&bc)[2], U3q (&an)[4], const f32x4 (&lo)[4],
                                       const f32x4 (&hi)[4], f32x2 (&r1)[16]) {
    if constexpr (I < 48) {
        constexpr int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};
        constexpr int p = I / 8, m = (I % 8) / 2, t = I % 2;
        __builtin_amdgcn_s_setprio(1);
        acc[m][NB + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ac[m].p[pa[p]]), bc[t].p[pb[p]], acc[m][NB + t], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I % 2 == 1) {
            split_step16<SBASE + I / 2>(an, lo, hi, r1);
            __builtin_amdgcn_sched_barrier(0);
        }
        half16<I + 1, NB, SBASE>(acc, ac, bc, an, lo, hi, r1);
    }
}
template <int I>
__device__ __forceinline__ void split_all16(U3q (&an)[4], const f32x4 (&lo)[4], const f32x4 (&hi)[4], f32x2 (&r1)[16]) {
    if constexpr (I < 48) { split_step16<I>(an, lo, hi, r1); split_all16<I + 1>(an, lo, hi, r1); }
}
// weights: bf16x8 index ((kc * 16 + nt) * 3 + piece) * 64 + lane, kc = 32-deep chunk (8 per layer), nt = 16-column tile
__device__ __forceinline__ void gemm16(v4f (&acc)[4][4], const float* xs, const bf16x8* __restrict__ wp, const int wave, const int lane) {
    const float* xrow = xs + (lane & 15) * LDX + 8 * (lane >> 4);
    const bf16x8* wl = wp + (4 * wave) * 3 * 64 + lane;
    constexpr int NKC = 8;
    U3q a0[4], a1[4];
    Bf3 b0[2], b1[2];
    f32x4 lo[4], hi[4];
    f32x2 r1[16];
    auto loadb = [&](Bf3 (&b)[2], int kc, int nb) {
        kc = kc < NKC - 1 ? kc : NKC - 1;
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 16 + nb + t) * 3 + p) * 64];
    };
    auto loada = [&](int kc) {
        kc = kc < NKC - 1 ? kc : NKC - 1;
        DH_UNROLL for (int m = 0; m < 4; ++m) {
            lo[m] = *reinterpret_cast<const f32x4*>(xrow + m * 16 * LDX + kc * 32);
            hi[m] = *reinterpret_cast<const f32x4*>(xrow + m * 16 * LDX + kc * 32 + 4);
        }
    };
    loadb(b0, 0, 0); loada(0);
    split_all16<0>(a0, lo, hi, r1);
    loada(1);
    _Pragma("unroll 1") for (int kc = 0; kc < NKC; kc += 2) {
        // chunk kc from a0: pieces of kc+1 -> a1 (raw already in lo / hi); raw of kc+2 is read after the last split step
        loadb(b1, kc, 2);
        __builtin_amdgcn_sched_barrier(0);
        half16<0, 0, 0>(acc, a0, b0, a1, lo, hi, r1);
        loadb(b0, kc + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        half16<0, 2, 24>(acc, a0, b1, a1, lo, hi, r1);
        loada(kc + 2);
        loadb(b1, kc + 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        half16<0, 0, 0>(acc, a1, b0, a0, lo, hi, r1);
        loadb(b0, kc + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        half16<0, 2, 24>(acc, a1, b1, a0, lo, hi, r1);
        loada(kc + 3);
    }
}

// STORE: 1 = the layer's output tile also goes to HBM (non-temporal, as the training forward saves it)
template <int SHAPE, int STORE>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ wp, float* out, float* __restrict__ tiles_out, int layers, int tiles) {
    __shared__ __attribute__((aligned(16))) float X[TM * LDX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TM * LDX; i += 256) {
        unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        X[i] = (float)(h & 0xffffff) * (1.f / 16777216.f);
    }
    __syncthreads();
    float sum = 0.f;
    for (int tl = 0; tl < tiles; ++tl)
        for (int l = 0; l < layers; ++l) {
            const bf16x8* wlay = wp + (size_t)(l % 8) * 3 * 16 * 8 * 64;
            if constexpr (SHAPE == 32) {
                f32x16 acc[MT][2];
                acc_zero(acc);
                gemm_rows_s(acc, X, LDX, 16, wlay, wave, lane);
                if (STORE) acc_store_native(acc, tiles_out + ((size_t)(blockIdx.x * tiles + tl) * 8 + (l % 8)) * TILE_F, wave, lane);
                __syncthreads();
                DH_UNROLL for (int m = 0; m < MT; ++m)
                    DH_UNROLL for (int t = 0; t < 2; ++t)
                        DH_UNROLL for (int q = 0; q < 16; ++q) {
                            const float v = softplus_fast(acc[m][t][q] * 0.01f - 0.3f);
                            X[acc_row(m, q, lane) * LDX + acc_col(wave, t, lane)] = v;
                            sum += v;
                        }
                __syncthreads();
            } else {
                v4f acc[4][4];
                DH_UNROLL for (int m = 0; m < 4; ++m) DH_UNROLL for (int t = 0; t < 4; ++t) DH_UNROLL for (int q = 0; q < 4; ++q) acc[m][t][q] = 0.f;
                gemm16(acc, X, wlay, wave, lane);
                __syncthreads();
                DH_UNROLL for (int m = 0; m < 4; ++m)
                    DH_UNROLL for (int t = 0; t < 4; ++t)
                        DH_UNROLL for (int q = 0; q < 4; ++q) {
                            const float v = softplus_fast(acc[m][t][q] * 0.01f - 0.3f);
                            X[(m * 16 + 4 * (lane >> 4) + q) * LDX + 64 * wave + 16 * t + (lane & 15)] = v;
                            sum += v;
                        }
                __syncthreads();
            }
        }
    out[blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE, int STORE>
void run(const char* name, const bf16x8* wp, float* out, float* tiles_out, int grid) {
    const int layers = 8, tiles = 16, reps = 40;          // ~60 ms per arm: long enough for the clock to settle
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<SHAPE, STORE>), dim3(grid), dim3(256), 0, 0, wp, out, tiles_out, layers, tiles);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<SHAPE, STORE>), dim3(grid), dim3(256), 0, 0, wp, out, tiles_out, layers, tiles);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = 2.0 * TM * 256 * 256 * layers * tiles * grid;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s fp32-equivalent\n", name, ms, flop / ms / 1e9);
}

int main() {
    bf16x8* wp; float* out;
    const size_t nw = (size_t)8 * 3 * 16 * 8 * 64;
    hipMalloc(&wp, nw * sizeof(bf16x8));
    std::vector<unsigned short> h(nw * 8);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {          // random bf16 in +-[2^-7, 1): random sign, 3 exponent bits, 7 mantissa bits
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        h[i] = (unsigned short)(((r & 1) << 15) | ((0x78 + ((r >> 1) & 7)) << 7) | ((r >> 4) & 0x7f));
    }
    hipMemcpy(wp, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int grid = 512;
    hipMalloc(&out, (size_t)grid * 256 * 4);
    float* tiles_out;
    hipMalloc(&tiles_out, (size_t)grid * 16 * 8 * TILE_F * 4);      // 4.3 GB: every (workgroup, tile, layer) its own 64 KB tile
    for (int round = 0; round < 3; ++round) {
        run<32, 0>("32x32x16 (tile16.h core)", wp, out, tiles_out, grid);
        run<16, 0>("16x16x32", wp, out, tiles_out, grid);
        run<32, 1>("32x32x16 + tile store to HBM", wp, out, tiles_out, grid);
    }
    return 0;
}
