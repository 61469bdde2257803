// Split-bf16 tile GEMM core (gfx950): fp32 activations and weights are carried as three bf16 pieces x = x1 + x2 + x3
// (exact residuals, 24 mantissa bits) and every fp32 product is formed as the six bf16 products x1w1, x1w2, x2w1, x1w3,
// x2w2, x3w1 accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 2^-24 relative, i.e. fp32 accuracy (measured
// scripts/micro/bf16_split_accuracy.py: 1.1e-7 vs fp64 against 2.9e-7 for a plain fp32 GEMM) at 6 x 32 cycles per 16-deep
// k-step instead of 8 x 64 for v_mfma_f32_32x32x2_f32.  The accumulator layout of the two instructions is the same
// 32 x 32 fp32 tile, so epilogues, native saved tiles and the dW kernels are unchanged.
//
// Two ways of feeding the A operand (both in this file; the faster one per kernel ships):
//   * split-on-fetch (gemm_rows_s / gemm_auxout_s): the LDS image stays tile.h's fp32 image (two workgroups per CU) and
//     each wave splits its A fragments as it reads them;
//   * piece planes (gemm16_rows / gemm16_auxout / acc_to_lds16 ...): three bf16 planes [TM x 256], row stride LDB = 264
//     (528 B: 16 lanes x 16 B cover all 64 banks), plus aux planes [TM x 48] (stride 56); the split happens once, in the
//     epilogue that writes the image (122 KB: one workgroup per CU).
// Packed weights (pack.hip): bf16x8 index ((kc*NT + nt)*3 + piece)*64 + lane holds
//   M[k = 16 kc + 8 (lane>>5) + s][n = 32 nt + (lane&31)], s = 0..7.
#pragma once
#include "tile.h"

namespace dh {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
struct Bf3 { bf16x8 p[3]; };

constexpr int LDB = 264;             // bf16 row stride of a main piece plane
constexpr int LDA16 = 56;            // bf16 row stride of an aux piece plane (48 valid columns)
constexpr int P_MAIN = TM * LDB;     // elements per main piece plane
constexpr int P_AUX = TM * LDA16;
constexpr int AUX_KC = 3;            // 48 / 16

__device__ __forceinline__ void split_f32(float v, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// 8 fp32 (two f32x4) -> three bf16x8 pieces
__device__ __forceinline__ Bf3 split3(const f32x4& lo, const f32x4& hi) {
    Bf3 r;
    DH_UNROLL for (int e = 0; e < 8; ++e) {
        __bf16 h1, h2, h3;
        split_f32(e < 4 ? lo[e] : hi[e - 4], h1, h2, h3);
        r.p[0][e] = h1; r.p[1][e] = h2; r.p[2][e] = h3;
    }
    return r;
}

__device__ __forceinline__ f32x16 mfma6(const Bf3& a, const Bf3& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[2], b.p[0], c, 0, 0, 0);      // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[1], b.p[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.p[0], b.p[0], c, 0, 0, 0);
    return c;
}

// acc[m][t] += X[TM x 16 nkc] * M (NT = 8).  xs: piece-0 plane in LDS (planes pstride elements apart, row stride ld).
// Operands of k-chunk kc+1 are issued before the MFMAs of kc (order pinned with sched_barrier, as in tile.h gemm_rows).
__device__ __forceinline__ void gemm16_rows(f32x16 (&acc)[MT][2], const __bf16* xs, const int pstride, const int ld,
                                            const int nkc, const bf16x8* __restrict__ wp, const int wave, const int lane) {
    const __bf16* xrow = xs + (lane & 31) * ld + 8 * (lane >> 5);
    const bf16x8* wl = wp + (2 * wave) * 3 * 64 + lane;
    Bf3 a0[MT], b0[2], a1[MT], b1[2];
    auto fetch = [&](Bf3 (&a)[MT], Bf3 (&b)[2], int kc) {
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 8 + t) * 3 + p) * 64];
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int p = 0; p < 3; ++p)
                a[m].p[p] = *reinterpret_cast<const bf16x8*>(xrow + p * pstride + m * 32 * ld + kc * 16);
    };
    auto mul = [&](const Bf3 (&a)[MT], const Bf3 (&b)[2]) {
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int t = 0; t < 2; ++t) acc[m][t] = mfma6(a[m], b[t], acc[m][t]);
    };
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, (kc + 1 < nkc) ? kc + 1 : kc);
        __builtin_amdgcn_sched_barrier(0);
        mul(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nkc) {
            fetch(a0, b0, (kc + 2 < nkc) ? kc + 2 : kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            mul(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// accumulators -> the three piece planes of the LDS main image
__device__ __forceinline__ void acc_to_lds16(const f32x16 (&acc)[MT][2], __bf16* xs, int wave, int lane) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t) {
            __bf16* base = xs + (m * 32 + 4 * (lane >> 5)) * LDB + acc_col(wave, t, lane);
            DH_UNROLL for (int r = 0; r < 16; ++r) {
                __bf16 h1, h2, h3;
                split_f32(acc[m][t][r], h1, h2, h3);
                __bf16* e = base + ((r & 3) + 8 * (r >> 2)) * LDB;
                e[0] = h1; e[P_MAIN] = h2; e[2 * P_MAIN] = h3;
            }
        }
}

// positional embedding of the tile's points into the aux piece planes (columns 39..47 zero)
__device__ __forceinline__ void embed_tile16(const float* __restrict__ pts, int64_t base, int64_t npts, __bf16* aux, int tid) {
    constexpr int TPP16 = 256 / TM;
    const int p = tid & (TM - 1), part = tid / TM;
    const int64_t gp = base + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (gp < npts) { x[0] = pts[gp * 3 + 0]; x[1] = pts[gp * 3 + 1]; x[2] = pts[gp * 3 + 2]; }
    __bf16* row = aux + p * LDA16;
    auto put = [&](int c, float v) {
        __bf16 h1, h2, h3;
        split_f32(v, h1, h2, h3);
        row[c] = h1; row[P_AUX + c] = h2; row[2 * P_AUX + c] = h3;
    };
    if (part == 0) { put(0, x[0]); put(1, x[1]); put(2, x[2]); }
    if (part == 1) { for (int c = 39; c < 48; ++c) put(c, 0.f); }
    for (int k = part; k < 6; k += TPP16) {
        const float f = (float)(1 << k);
        DH_UNROLL for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            put(3 + 6 * k + c, s);
            put(3 + 6 * k + 3 + c, co);
        }
    }
}

// per-point dot of the main image rows (pieces summed back to fp32) with a 256-vector; TPP threads per point
__device__ __forceinline__ float row_dot256_16(const __bf16* main, const float* __restrict__ w, int tid) {
    constexpr int TPP16 = 256 / TM, SEG = 256 / TPP16;
    const int p = tid / TPP16, part = tid % TPP16;
    const __bf16* xr = main + p * LDB + part * SEG;
    const float* wr = w + part * SEG;
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < SEG / 8; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + 8 * i);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + P_MAIN + 8 * i);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(xr + 2 * P_MAIN + 8 * i);
        DH_UNROLL for (int e = 0; e < 8; ++e) s = fmaf(((float)a[e] + (float)b[e]) + (float)c[e], wr[8 * i + e], s);
    }
    DH_UNROLL for (int off = 1; off < TPP16; off <<= 1) s += __shfl_xor(s, off);
    return s;
}

// acc2[.] += X[rows of this wave's m-tile][16 nkc] * M (NT = 2): the 64-wide "aux" output (tile.h gemm_auxout)
__device__ __forceinline__ void gemm16_auxout(f32x16 (&acc2)[AUX_NTW], const __bf16* xs, const int nkc,
                                              const bf16x8* __restrict__ wp, const int wave, const int lane) {
    const __bf16* xrow = xs + (32 * aux_mtile(wave) + (lane & 31)) * LDB + 8 * (lane >> 5);
    const bf16x8* wl = wp + aux_ntile(wave, 0) * 3 * 64 + lane;
    auto fetch = [&](Bf3& a, Bf3 (&b)[AUX_NTW], int kc) {
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 2 + t) * 3 + p) * 64];
        DH_UNROLL for (int p = 0; p < 3; ++p) a.p[p] = *reinterpret_cast<const bf16x8*>(xrow + p * P_MAIN + kc * 16);
    };
    Bf3 a0, a1, b0[AUX_NTW], b1[AUX_NTW];
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, (kc + 1 < nkc) ? kc + 1 : kc);
        __builtin_amdgcn_sched_barrier(0);
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma6(a0, b0[t], acc2[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nkc) {
            fetch(a0, b0, (kc + 2 < nkc) ? kc + 2 : kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma6(a1, b1[t], acc2[t]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// one fp32 value -> the three piece planes of an aux row
__device__ __forceinline__ void aux_put16(__bf16* row, int c, float v) {
    __bf16 h1, h2, h3;
    split_f32(v, h1, h2, h3);
    row[c] = h1; row[P_AUX + c] = h2; row[2 * P_AUX + c] = h3;
}

// LDS aux piece planes (cols < 40 valid) -> fp32 aux native tile in HBM (same layout as tile.h aux_lds_to_native)
__device__ __forceinline__ void aux_lds16_to_native(const __bf16* aux, float* __restrict__ tile, int wave, int lane) {
    f32x16 a2[AUX_NTW];
    DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
        const int col = aux_col(wave, tt, lane);
        DH_UNROLL for (int r = 0; r < 16; ++r) {
            float v = 0.f;
            if (col < AUXW) {
                const __bf16* e = aux + aux_row(wave, r, lane) * LDA16 + col;
                v = ((float)e[0] + (float)e[P_AUX]) + (float)e[2 * P_AUX];
            }
            a2[tt][r] = v;
        }
    }
    aux_store_native(a2, tile, wave, lane);
}

// A saved native tile as registers (same element order as the accumulators): issued BEFORE a GEMM so its HBM latency sits
// under the MFMAs -- with one workgroup per CU nothing else would hide it.
struct TileRegs { f32x4 v[MT][2][4]; };
__device__ __forceinline__ void tile_prefetch(TileRegs& t, const float* __restrict__ tile, int wave, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(tile) + (size_t)wave * MT * 8 * 64 + lane;
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int tt = 0; tt < 2; ++tt)
            DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) t.v[m][tt][r4] = p[((m * 2 + tt) * 4 + r4) * 64];
}

// ---------------------------------------------------------------- split-on-fetch form (the shipping chain kernels)
// The LDS activation image stays the fp32 image of tile.h (78 KB with the aux image -> TWO workgroups per CU, whose
// epilogues and barriers overlap each other's MFMAs); each wave splits its A fragments into bf16 pieces as it fetches
// them (two ds_read_b128 -> 8 fp32 -> three bf16x8).  The split is redone by each of the four waves, but it sits in the
// vector-issue shadow of the matrix pipe (an MFMA holds issue for 8 of its 32 cycles): scripts/micro/
// bf16x3_fp32lds_micro.hip measures 233-247 TFLOP/s fp32-equivalent for this chain WITH softplus and write-back,
// against 218 for the piece-plane image at one workgroup per CU.  No sched_barrier between the fetch/split block and the
// MFMA block: interleaving them is exactly what is wanted (micro: 247 vs 233 pinned).
__device__ __forceinline__ void gemm_rows_s(f32x16 (&acc)[MT][2], const float* xs, const int ldx, const int nkc,
                                            const bf16x8* __restrict__ wp, const int wave, const int lane) {
    const float* xrow = xs + (lane & 31) * ldx + 8 * (lane >> 5);
    const bf16x8* wl = wp + (2 * wave) * 3 * 64 + lane;
    Bf3 a0[MT], b0[2], a1[MT], b1[2];
    const int last = nkc - 1;
    auto fetch = [&](Bf3 (&a)[MT], Bf3 (&b)[2], int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 8 + t) * 3 + p) * 64];
        DH_UNROLL for (int m = 0; m < MT; ++m) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16 + 4);
            a[m] = split3(lo, hi);
        }
    };
    auto mul = [&](const Bf3 (&a)[MT], const Bf3 (&b)[2]) {
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int t = 0; t < 2; ++t) acc[m][t] = mfma6(a[m], b[t], acc[m][t]);
    };
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, kc + 1);
        mul(a0, b0);
        if (kc + 1 < nkc) {
            fetch(a0, b0, kc + 2);
            mul(a1, b1);
        }
    }
}

// 64-wide "aux" output from the fp32 main image (tile.h gemm_auxout), split-on-fetch
__device__ __forceinline__ void gemm_auxout_s(f32x16 (&acc2)[AUX_NTW], const float* xs, const int nkc,
                                              const bf16x8* __restrict__ wp, const int wave, const int lane) {
    const float* xrow = xs + (32 * aux_mtile(wave) + (lane & 31)) * LDX + 8 * (lane >> 5);
    const bf16x8* wl = wp + aux_ntile(wave, 0) * 3 * 64 + lane;
    const int last = nkc - 1;
    auto fetch = [&](Bf3& a, Bf3 (&b)[AUX_NTW], int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 2 + t) * 3 + p) * 64];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + kc * 16);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + kc * 16 + 4);
        a = split3(lo, hi);
    };
    Bf3 a0, a1, b0[AUX_NTW], b1[AUX_NTW];
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, kc + 1);
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma6(a0, b0[t], acc2[t]);
        if (kc + 1 < nkc) {
            fetch(a0, b0, kc + 2);
            DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma6(a1, b1[t], acc2[t]);
        }
    }
}

// ---------------------------------------------------------------- 8-wave variant (512 threads, two waves per SIMD)
// Same tile, same LDS image; wave w8 = 0..7 owns the single 32-column tile ct = w8 (native index: wave = ct >> 1,
// t = ct & 1), so acc is [MT][1] and the B operand per wave halves while every wave still reads the whole A image.
// The second wave on each SIMD covers the epilogue (activation, split, 16-bit LDS writes) of the first.
__device__ __forceinline__ void gemm16_rows_w8(f32x16 (&acc)[MT], const __bf16* xs, const int pstride, const int ld,
                                               const int nkc, const bf16x8* __restrict__ wp, const int w8, const int lane) {
    const __bf16* xrow = xs + (lane & 31) * ld + 8 * (lane >> 5);
    const bf16x8* wl = wp + w8 * 3 * 64 + lane;
    Bf3 a0[MT], b0, a1[MT], b1;
    auto fetch = [&](Bf3 (&a)[MT], Bf3& b, int kc) {
        DH_UNROLL for (int p = 0; p < 3; ++p) b.p[p] = wl[(kc * 8 * 3 + p) * 64];
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int p = 0; p < 3; ++p)
                a[m].p[p] = *reinterpret_cast<const bf16x8*>(xrow + p * pstride + m * 32 * ld + kc * 16);
    };
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, (kc + 1 < nkc) ? kc + 1 : kc);
        __builtin_amdgcn_sched_barrier(0);
        DH_UNROLL for (int m = 0; m < MT; ++m) acc[m] = mfma6(a0[m], b0, acc[m]);
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < nkc) {
            fetch(a0, b0, (kc + 2 < nkc) ? kc + 2 : kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            DH_UNROLL for (int m = 0; m < MT; ++m) acc[m] = mfma6(a1[m], b1, acc[m]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ void acc_to_lds16_w8(const f32x16 (&acc)[MT], __bf16* xs, int w8, int lane) {
    DH_UNROLL for (int m = 0; m < MT; ++m) {
        __bf16* base = xs + (m * 32 + 4 * (lane >> 5)) * LDB + 32 * w8 + (lane & 31);
        DH_UNROLL for (int r = 0; r < 16; ++r) {
            __bf16 h1, h2, h3;
            split_f32(acc[m][r], h1, h2, h3);
            __bf16* e = base + ((r & 3) + 8 * (r >> 2)) * LDB;
            e[0] = h1; e[P_MAIN] = h2; e[2 * P_MAIN] = h3;
        }
    }
}

// embedding with 512 threads: thread (p = tid & 63, part = tid >> 6): part 0 x, part 1 zero pad, parts 0..5 one frequency each
__device__ __forceinline__ void embed_tile16_w8(const float* __restrict__ pts, int64_t base, int64_t npts, __bf16* aux, int tid) {
    static_assert(TM == 64, "8-wave kernels are written for 64-point tiles");
    const int p = tid & 63, part = tid >> 6;
    const int64_t gp = base + p;
    float x[3] = {0.f, 0.f, 0.f};
    if (gp < npts) { x[0] = pts[gp * 3 + 0]; x[1] = pts[gp * 3 + 1]; x[2] = pts[gp * 3 + 2]; }
    __bf16* row = aux + p * LDA16;
    if (part == 6) { aux_put16(row, 0, x[0]); aux_put16(row, 1, x[1]); aux_put16(row, 2, x[2]); }
    if (part == 7) { for (int c = 39; c < 48; ++c) aux_put16(row, c, 0.f); }
    if (part < 6) {
        const float f = (float)(1 << part);
        DH_UNROLL for (int c = 0; c < 3; ++c) {
            float s, co;
            sincosf(x[c] * f, &s, &co);
            aux_put16(row, 3 + 6 * part + c, s);
            aux_put16(row, 3 + 6 * part + 3 + c, co);
        }
    }
}

// per-point dot with 8 threads per point (point = tid >> 3)
__device__ __forceinline__ float row_dot256_16_w8(const __bf16* main, const float* __restrict__ w, int tid) {
    const int p = tid >> 3, part = tid & 7;
    const __bf16* xr = main + p * LDB + part * 32;
    const float* wr = w + part * 32;
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < 4; ++i) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(xr + 8 * i);
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(xr + P_MAIN + 8 * i);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(xr + 2 * P_MAIN + 8 * i);
        DH_UNROLL for (int e = 0; e < 8; ++e) s = fmaf(((float)a[e] + (float)b[e]) + (float)c[e], wr[8 * i + e], s);
    }
    DH_UNROLL for (int off = 1; off < 8; off <<= 1) s += __shfl_xor(s, off);
    return s;
}

}  // namespace dh
