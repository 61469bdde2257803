// Helpers shared by the forward and backward MLP chain kernels.
#pragma once
#include "tile.h"
#include "tile16.h"

namespace dh {

// ------------------------------------------------------------------------------------------------
// positional embedding of 128 points into the LDS aux image: [x, sin(2^k x), cos(2^k x)]_{k<6}  (App. A.1)
// 256 threads: thread handles point tid&127 and frequencies 3*(tid>>7) .. +2.
// ------------------------------------------------------------------------------------------------
constexpr int TPP = 256 / TM;      // threads per point in the per-point VALU phases (2 or 4)
__device__ __forceinline__ void embed_tile(const float* __restrict__ pts, int64_t base, int64_t npts, float* aux, int tid) {
    const int p = tid & (TM - 1), part = tid / TM;
    const int64_t gp = base + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (gp < npts) { x[0] = pts[gp * 3 + 0]; x[1] = pts[gp * 3 + 1]; x[2] = pts[gp * 3 + 2]; }
    float* row = aux + p * LDA;
    if (part == 0) { row[0] = x[0]; row[1] = x[1]; row[2] = x[2]; }
    if (part == 1) { for (int c = 39; c < LDA; ++c) row[c] = 0.f; }
    for (int k = part; k < 6; k += TPP) {
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
__device__ __forceinline__ void acc_map(f32x16 (&acc)[MT][2], F f) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r = 0; r < 16; ++r) acc[m][t][r] = f(m, t, r, acc[m][t][r]);
}

// per-point dot of the LDS main tile rows with a 256-vector: TPP threads per point (point = tid / TPP), result
// valid on every thread of the group
__device__ __forceinline__ float row_dot256(const float* main, const float* __restrict__ w, int tid) {
    constexpr int SEG = 256 / TPP;
    const int p = tid / TPP, part = tid % TPP;
    const f32x4* xr = reinterpret_cast<const f32x4*>(main + p * LDX + part * SEG);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w + part * SEG);
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < SEG / 4; ++i) {
        const f32x4 a = xr[i], b = wr[i];
        s = fmaf(a[0], b[0], s); s = fmaf(a[1], b[1], s); s = fmaf(a[2], b[2], s); s = fmaf(a[3], b[3], s);
    }
    DH_UNROLL for (int off = 1; off < TPP; off <<= 1) s += __shfl_xor(s, off);
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

static inline SdfPtrs make_sdf_ptrs(const float* packed) {
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

struct ColPtrs {
    const f32x4* fwd_main[4];
    const f32x4* rev_main[4];
    const f32x4* fwd_aux0;
    const f32x4* rev_aux0;
    const float* bias[4];
    const float* w4;
    const float* b4;
};
static inline ColPtrs make_col_ptrs(const float* packed) {
    ColPtrs C;
    for (int l = 0; l < 4; ++l) {
        C.fwd_main[l] = reinterpret_cast<const f32x4*>(packed + PACK.col_fwd_main[l]);
        C.rev_main[l] = reinterpret_cast<const f32x4*>(packed + PACK.col_rev_main[l]);
        C.bias[l] = packed + PACK.col_bias[l];
    }
    C.fwd_aux0 = reinterpret_cast<const f32x4*>(packed + PACK.col_fwd_aux0);
    C.rev_aux0 = reinterpret_cast<const f32x4*>(packed + PACK.col_rev_aux0);
    C.w4 = packed + PACK.col_w4;
    C.b4 = packed + PACK.col_b4;
    return C;
}


// ---- pointers into the split-bf16 packed weights (layout.h PACK16)
struct Sdf16Ptrs {
    const bf16x8* main16[N_SDF];
    const bf16x8* aux16[N_SDF];
    const bf16x8* rev16[N_SDF];
    const bf16x8* revaux16[N_SDF];
    const float* bias[N_SDF];
    const float* w8row0;
    const float* b8_0;
};
static inline Sdf16Ptrs make_sdf16_ptrs(const float* packed) {
    Sdf16Ptrs P;
    for (int l = 0; l < N_SDF; ++l) {
        P.main16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.sdf_fwd_main[l]);
        P.aux16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.sdf_fwd_aux[l]);
        P.rev16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.sdf_rev_main[l]);
        P.revaux16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.sdf_rev_aux[l]);
        P.bias[l] = packed + PACK.sdf_bias[l];
    }
    P.w8row0 = packed + PACK.sdf_w8row0;
    P.b8_0 = packed + PACK.sdf_b8_0;
    return P;
}

struct Col16Ptrs {
    const bf16x8* main16[4];
    const bf16x8* rev16[4];
    const bf16x8* aux16;
    const bf16x8* revaux16;
    const float* bias[4];
    const float* w4;
    const float* b4;
};
static inline Col16Ptrs make_col16_ptrs(const float* packed) {
    Col16Ptrs C;
    for (int l = 0; l < 4; ++l) {
        C.main16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.col_fwd_main[l]);
        C.rev16[l] = reinterpret_cast<const bf16x8*>(packed + PACK16.col_rev_main[l]);
        C.bias[l] = packed + PACK.col_bias[l];
    }
    C.aux16 = reinterpret_cast<const bf16x8*>(packed + PACK16.col_fwd_aux0);
    C.revaux16 = reinterpret_cast<const bf16x8*>(packed + PACK16.col_rev_aux0);
    C.w4 = packed + PACK.col_w4;
    C.b4 = packed + PACK.col_b4;
    return C;
}


// column sums of a [TM x 256] accumulator tile -> dst[256]
__device__ __forceinline__ void tile_colsum(const f32x16 (&acc)[MT][2], float* __restrict__ dst, int wave, int lane) {
    DH_UNROLL for (int t = 0; t < 2; ++t) {
        float s = 0.f;
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int r = 0; r < 16; ++r) s += acc[m][t][r];
        s += __shfl_xor(s, 32);
        if (lane < 32) dst[64 * wave + 32 * t + lane] = s;
    }
}

}  // namespace dh
