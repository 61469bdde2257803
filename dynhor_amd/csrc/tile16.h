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
//   * piece planes (gemm16_rows / gemm16_auxout / acc_to_lds16): three bf16 planes [TM x 256], row stride LDB = 264
//     (528 B: 16 lanes x 16 B cover all 64 banks); the split happens once, in the epilogue that writes the image
//     (one workgroup per CU).  Ships for the reverse chain and the colour backward, whose epilogue inputs are
//     prefetched into registers under the GEMM (tile_prefetch).
// Packed weights (pack.hip): bf16x8 index ((kc*NT + nt)*3 + piece)*64 + lane holds
//   M[k = 16 kc + 8 (lane>>5) + s][n = 32 nt + (lane&31)], s = 0..7.
#pragma once
#include "tile.h"

namespace dh {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
struct Bf3 { bf16x8 p[3]; };

constexpr int LDB = 264;             // bf16 row stride of a main piece plane
constexpr int P_MAIN = TM * LDB;     // elements per main piece plane
constexpr int AUX_KC = 3;            // k-chunks of the aux image: 48 / 16

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

}  // namespace dh
