// Split-bf16 tile GEMM core (gfx950): fp32 activations and weights are carried as three bf16 pieces x = x1 + x2 + x3
// (exact residuals, 24 mantissa bits) and every fp32 product is formed as the six bf16 products x1w1, x1w2, x2w1, x1w3,
// x2w2, x3w1 accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 2^-24 relative, i.e. fp32 accuracy (measured
// scripts/micro/bf16_split_accuracy.py: 1.1e-7 vs fp64 against 2.9e-7 for a plain fp32 GEMM) at 6 x 32 cycles per 16-deep
// k-step instead of 8 x 64 for v_mfma_f32_32x32x2_f32.  The accumulator layout of the two instructions is the same
// 32 x 32 fp32 tile, so epilogues, native saved tiles and the dW kernels are unchanged.
//
// The A operand is fed split-on-fetch (gemm_rows_s / gemm_auxout_s): the LDS image stays tile.h's fp32 image (two
// workgroups per CU) and each wave splits its A fragments as it reads them.  (A piece-plane form -- three bf16 planes in LDS,
// split once in the epilogue, one workgroup per CU -- shipped for two kernels until the GEMM core below got its pinned order
// and MFMA priority; it is in git history, its measurements in DESIGN.md section 3.)
// Packed weights (pack.hip): bf16x8 index ((kc*NT + nt)*3 + piece)*64 + lane holds
//   M[k = 16 kc + 8 (lane>>5) + s][n = 32 nt + (lane&31)], s = 0..7.
#pragma once
#include "tile.h"

namespace dh {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
struct Bf3 { bf16x8 p[3]; };

constexpr int AUX_KC = 3;            // k-chunks of the aux image: 48 / 16

__device__ __forceinline__ void split_f32(float v, __bf16& h1, __bf16& h2, __bf16& h3) {
    h1 = (__bf16)v;
    const float r1 = v - (float)h1;
    h2 = (__bf16)r1;
    h3 = (__bf16)(r1 - (float)h2);
}

// 8 fp32 (two f32x4) -> three bf16x8 pieces, written pair-wise: per pair 3 v_cvt_pk_bf16_f32 + 4 unpack (shift / and) + 2
// subtractions of a float pair (one v_pk_add_f32 each, which the compiler turns back into two v_add_f32 where it sits in an
// MFMA's shadow: packed fp32 ops do not issue beside the matrix pipe) = 4.5 - 5.5 vector ops per value; the element-wise form
// left the pairing to the SLP vectoriser, which got 5.8 with a third of the conversions unpaired.  Same values as split_f32.
typedef __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16 bf16x2;
typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
__device__ __forceinline__ unsigned pack_bf16x2(f32x2 v) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 unpack_bf16x2(unsigned h) {
    f32x2 r;
    r[0] = __builtin_bit_cast(float, h << 16);
    r[1] = __builtin_bit_cast(float, h & 0xffff0000u);
    return r;
}
__device__ __forceinline__ Bf3 split3(const f32x4& lo, const f32x4& hi) {
    u32x4 p0, p1, p2;
    DH_UNROLL for (int j = 0; j < 4; ++j) {
        f32x2 x;
        x[0] = j < 2 ? lo[2 * j] : hi[2 * j - 4];
        x[1] = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
        const unsigned h = pack_bf16x2(x);
        const f32x2 r1 = x - unpack_bf16x2(h);
        const unsigned m = pack_bf16x2(r1);
        const f32x2 r2 = r1 - unpack_bf16x2(m);
        p0[j] = h; p1[j] = m; p2[j] = pack_bf16x2(r2);
    }
    Bf3 r;
    r.p[0] = __builtin_bit_cast(bf16x8, p0);
    r.p[1] = __builtin_bit_cast(bf16x8, p1);
    r.p[2] = __builtin_bit_cast(bf16x8, p2);
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

// ---------------------------------------------------------------- split-on-fetch form (the shipping chain kernels)
// The LDS activation image stays the fp32 image of tile.h (78 KB with the aux image -> TWO workgroups per CU, whose
// epilogues and barriers overlap each other's MFMAs); each wave splits its A fragments into bf16 pieces as it fetches
// them (two ds_read_b128 -> 8 fp32 -> three bf16x8).  The split is redone by each of the four waves.
//
// Program order is written out and pinned with sched_barrier(0) (round 2, same-box A/B of six schedules):
//   * raw fp32 A fragments are read two k-chunks ahead (two register sets), weight pieces one k-chunk ahead;
//   * the 24 MFMAs of a k-chunk go product-major (consecutive ones hit different accumulators; per accumulator the order
//     is mfma6's, smallest terms first), and after MFMA i comes one third of the split of pair i/3 of the NEXT k-chunk
//     (5 / 5 / 1 vector ops) -- left to itself the compiler put ~90 vector ops in front of the first MFMA and ran 24 bare;
//   * every MFMA is issued at raised wave priority (s_setprio 1 ... 0): with two waves per SIMD the arbiter otherwise lets
//     the other wave's split ops in ahead of a ready MFMA.  This alone is -8 % on sdf_forward (1.65 -> 1.52 ms), -9 % on
//     the tangent chain and -12 % on the reverse chain; priority held over the whole GEMM instead: nothing.
struct U3 { u32x4 p[3]; };
struct RawA { f32x4 lo[MT], hi[MT]; };
struct SplitState { f32x2 r1[4 * MT]; };
template <int STEP>
__device__ __forceinline__ void split_step(U3 (&a)[MT], const RawA& r, SplitState& st) {
    constexpr int j = STEP / 3, s = STEP % 3, m = j / 4, q = j % 4;
    if constexpr (s == 0) {
        f32x2 x;
        x[0] = q < 2 ? r.lo[m][2 * q] : r.hi[m][2 * q - 4];
        x[1] = q < 2 ? r.lo[m][2 * q + 1] : r.hi[m][2 * q - 3];
        const unsigned h = pack_bf16x2(x);
        a[m].p[0][q] = h;
        st.r1[j] = x - unpack_bf16x2(h);
    } else if constexpr (s == 1) {
        const unsigned mm = pack_bf16x2(st.r1[j]);
        a[m].p[1][q] = mm;
        st.r1[j] = st.r1[j] - unpack_bf16x2(mm);
    } else {
        a[m].p[2][q] = pack_bf16x2(st.r1[j]);
    }
}
template <int I>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[MT][2], const U3 (&a)[MT], const Bf3 (&b)[2]) {
    constexpr int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};      // mfma6's product order
    constexpr int p = I / (2 * MT), mt = I % (2 * MT), m = mt / 2, t = mt % 2;
    __builtin_amdgcn_s_setprio(1);
    acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[m].p[pa[p]]), b[t].p[pb[p]], acc[m][t], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}
template <int I, int N>
__device__ __forceinline__ void phase_steps(f32x16 (&acc)[MT][2], const U3 (&ac)[MT], const Bf3 (&bc)[2], U3 (&an)[MT],
                                            const RawA& rn, SplitState& st) {
    if constexpr (I < N) {
        mfma_step<I>(acc, ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        split_step<I>(an, rn, st);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps<I + 1, N>(acc, ac, bc, an, rn, st);
    }
}
template <int I, int N>
__device__ __forceinline__ void mfma_only(f32x16 (&acc)[MT][2], const U3 (&ac)[MT], const Bf3 (&bc)[2]) {
    if constexpr (I < N) { mfma_step<I>(acc, ac, bc); mfma_only<I + 1, N>(acc, ac, bc); }
}
template <int I, int N>
__device__ __forceinline__ void split_only(U3 (&an)[MT], const RawA& rn, SplitState& st) {
    if constexpr (I < N) { split_step<I>(an, rn, st); split_only<I + 1, N>(an, rn, st); }
}
// acc[m][t] += X[TM x 16 nkc] * M.  xs: the fp32 LDS image (row stride ldx floats), wp: packed weight pieces (NT = 8)
__device__ __forceinline__ void gemm_rows_s(f32x16 (&acc)[MT][2], const float* xs, const int ldx, const int nkc,
                                            const bf16x8* __restrict__ wp, const int wave, const int lane) {
    static_assert(MT == 2, "24 MFMAs per k-chunk = 8 pairs x 3 split steps");
    const float* xrow = xs + (lane & 31) * ldx + 8 * (lane >> 5);
    const bf16x8* wl = wp + (2 * wave) * 3 * 64 + lane;
    const int last = nkc - 1;
    U3 a0[MT], a1[MT];
    Bf3 b0[2], b1[2];
    RawA r0, r1;
    SplitState st;
    auto loadb = [&](Bf3 (&b)[2], int kc) {                 // past the end: clamped (a harmless re-read), no branch
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 3; ++p) b[t].p[p] = wl[((kc * 8 + t) * 3 + p) * 64];
    };
    auto loada = [&](RawA& r, int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int m = 0; m < MT; ++m) {
            r.lo[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16);
            r.hi[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16 + 4);
        }
    };
    loadb(b0, 0); loada(r0, 0); loada(r1, 1);
    split_only<0, 24>(a0, r0, st);
    _Pragma("unroll 1") for (int kc = 0; kc + 1 < nkc; kc += 2) {
        loadb(b1, kc + 1); loada(r0, kc + 2);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps<0, 24>(acc, a0, b0, a1, r1, st);
        loadb(b0, kc + 2); loada(r1, kc + 3);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps<0, 24>(acc, a1, b1, a0, r0, st);
    }
    if (nkc & 1) mfma_only<0, 24>(acc, a0, b0);
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
