// Tile-resident MLP chain kernels (SDF no-grad / train forward / input-gradient / colour forward and the
// backward chains).  See tile.h for the tile GEMM primitives and layouts, DESIGN.md for the math.
#include "tile.h"
#include "kernels.h"

namespace dh {

// ------------------------------------------------------------------------------------------------
// positional embedding of 128 points into the LDS aux image: [x, sin(2^k x), cos(2^k x)]_{k<6}  (App. A.1)
// 256 threads: thread handles point tid&127 and frequencies 3*(tid>>7) .. +2.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void embed_tile(const float* __restrict__ pts, int64_t base, int64_t npts, float* aux, int tid) {
    const int p = tid & 127, half = tid >> 7;
    const int64_t gp = base + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (gp < npts) { x[0] = pts[gp * 3 + 0]; x[1] = pts[gp * 3 + 1]; x[2] = pts[gp * 3 + 2]; }
    float* row = aux + p * LDA;
    if (half == 0) { row[0] = x[0]; row[1] = x[1]; row[2] = x[2]; }
    else           { row[39] = 0.f; row[40] = 0.f; row[41] = 0.f; row[42] = 0.f; row[43] = 0.f; }
    DH_UNROLL for (int kk = 0; kk < 3; ++kk) {
        const int k = half * 3 + kk;
        const float f = (float)(1 << k);
        DH_UNROLL for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            row[3 + 6 * k + c] = s;
            row[3 + 6 * k + 3 + c] = co;
        }
    }
}

template <class F>
__device__ __forceinline__ void acc_map(f32x16 (&acc)[4][2], F f) {
    DH_UNROLL for (int m = 0; m < 4; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r = 0; r < 16; ++r) acc[m][t][r] = f(m, t, r, acc[m][t][r]);
}

// per-point dot of the LDS main tile rows with a 256-vector: 2 threads per point, result valid on even threads
__device__ __forceinline__ float row_dot256(const float* main, const float* __restrict__ w, int tid) {
    const int p = tid >> 1, half = tid & 1;
    const f32x4* xr = reinterpret_cast<const f32x4*>(main + p * LDX + half * 128);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w + half * 128);
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < 32; ++i) {
        const f32x4 a = xr[i], b = wr[i];
        s = fmaf(a[0], b[0], s); s = fmaf(a[1], b[1], s); s = fmaf(a[2], b[2], s); s = fmaf(a[3], b[3], s);
    }
    s += __shfl_xor(s, 1);
    return s;
}

struct SdfPtrs {
    const f32x4* fwd_main[N_SDF];
    const f32x4* fwd_aux[N_SDF];
    const f32x4* rev_main[N_SDF];
    const f32x4* rev_aux[N_SDF];
    const float* bias[N_SDF];
    const float* w8row0;
    const float* b8_0;
};

static SdfPtrs make_sdf_ptrs(const float* packed) {
    SdfPtrs P;
    for (int l = 0; l < N_SDF; ++l) {
        P.fwd_main[l] = reinterpret_cast<const f32x4*>(packed + PACK.sdf_fwd_main[l]);
        P.fwd_aux[l] = reinterpret_cast<const f32x4*>(packed + PACK.sdf_fwd_aux[l]);
        P.rev_main[l] = reinterpret_cast<const f32x4*>(packed + PACK.sdf_rev_main[l]);
        P.rev_aux[l] = reinterpret_cast<const f32x4*>(packed + PACK.sdf_rev_aux[l]);
        P.bias[l] = packed + PACK.sdf_bias[l];
    }
    P.w8row0 = packed + PACK.sdf_w8row0;
    P.b8_0 = packed + PACK.sdf_b8_0;
    return P;
}

// ------------------------------------------------------------------------------------------------
// K1: SDF forward, no grad, sdf only (hierarchical up-sampling evaluations; SURVEY §8 a5 "no-grad").
// lin8 reduces to its row 0: a 256-long dot per point, done on the VALU.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void sdf_nograd_kernel(SdfPtrs P, const float* __restrict__ pts, int64_t npts,
                                                             float* __restrict__ sdf_out) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        embed_tile(pts, tile * TM, npts, saux, tid);
        __syncthreads();
        f32x16 acc[4][2];
        for (int l = 0; l < 8; ++l) {
            acc_zero(acc);
            if (l > 0) gemm_rows(acc, smain, LDX, l == 4 ? 28 : 32, P.fwd_main[l], wave, lane);
            if (l == 0 || l == 4) gemm_rows(acc, saux, LDA, 5, P.fwd_aux[l], wave, lane);
            const float b0 = P.bias[l][acc_col(wave, 0, lane)], b1 = P.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return softplus100(v + (t ? b1 : b0)); });
            __syncthreads();                 // every wave finished reading smain as the A operand
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        const float s = row_dot256(smain, P.w8row0, tid) + P.b8_0[0];
        const int64_t gp = tile * TM + (tid >> 1);
        if ((tid & 1) == 0 && gp < npts) sdf_out[gp] = s;
        __syncthreads();                     // smain/saux are rewritten by the next tile
    }
}

int launch_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, int grid, hipStream_t stream) {
    if (npts <= 0) return 0;
    const int64_t ntiles = (npts + TM - 1) / TM;
    const int g = (int)(ntiles < grid ? ntiles : grid);
    hipLaunchKernelGGL(sdf_nograd_kernel, dim3(g), dim3(256), 0, stream, make_sdf_ptrs(packed), pts, npts, sdf);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
