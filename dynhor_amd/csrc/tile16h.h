// Two-piece fp16 tile GEMM core (gfx950; round 4): every fp32 product from THREE v_mfma_f32_32x32x16_f16 instead of tile16.h's six
// bf16 ones.
//
//   X = S x with S a power of two;  X = Xh + Xl,  Xh = fp16(X),  Xl = fp16(X - Xh)  (the UNSCALED residual)
//   x w  <-  (Xh Wl + Xl Wh + Xh Wh) / (S_x S_w)        fp32 accumulation, smallest terms first
//
// Xh carries 11 significant bits, the residual (exact in fp32: v_fma_mix_f32 straight from the f16 half) is <= 2^-11 |X| and is
// rounded to 11 bits again, so X is represented to 2^-22 |X| in the worst case and 2^-24.6 |X| rms (fp32 itself: 2^-24 / 2^-25.8)
// -- as long as Xl is a normal fp16 number.  Where it is an fp16 SUBNORMAL (the matrix cores do not flush them:
// scripts/micro/tchain2_micro.hip gemm_check) the representation error is 2^-25 ABSOLUTE in units of X.  The scheme is therefore fp32-accurate once S puts the operand's
// dominant magnitudes at >= O(1): measured on the hardware 1.9e-7 relative L2 vs fp64 against 2.3e-7 for the exact fp32 MFMA and
// 2.5e-7 for bf16x3 (profiles/r04_ab_f16x2_chain_micro.json; tests/test_cpu_f16_split.py restates it on the CPU), and wrong by
// orders of magnitude for UNSCALED 1e-6 operands.  Hence the scales:
//   * weights: per linear, from max |W| (pack.hip -> layout.h PACKH.wabs): scaled maximum in [8, 16);
//   * O(1) operands (softplus / ReLU activations, embeddings, features): the constant H2_XS = 16 (overflow beyond 4094);
//   * everything else (adjoints, tangents, reverse-chain values): per 64-point tile and layer, from the tile's own maximum
//     (tile_scale below): scaled maximum in [256, 512), 128 x headroom to fp16's 65504.
// The LDS activation image of the tile-resident chains holds the SCALED fp32 values; each wave splits its A fragments as it
// fetches them (2 vector ops per value instead of 4.5): v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32 per pair.
// Packed weights (pack.hip packh_kernel): f16x8 index ((kc*NT + nt)*2 + piece)*64 + lane holds
//   S_w M[k = 16 kc + 8 (lane>>5) + s][n = 32 nt + (lane&31)], s = 0..7  (piece 0 = hi, 1 = lo).
#pragma once
#include "tile16.h"

namespace dh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
struct H2 { u32x4 p[2]; };                     // hi, lo pieces of 8 values (one MFMA operand each)

__device__ __forceinline__ unsigned pack_f16x2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v)); }
// v - float(h.lo), v - float(h.hi): one v_fma_mix_f32 each, the f16 half read in place (exact: |v - h| <= ulp_f16 / 2)
__device__ __forceinline__ f32x2 resid_f16x2(f32x2 v, unsigned h) {
    f32x2 r;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(h), "v"(v[0]));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(h), "v"(v[1]));
    return r;
}
__device__ __forceinline__ f32x16 mfma_h(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma3(const H2& a, const H2& b, f32x16 c) {
    c = mfma_h(a.p[0], b.p[1], c);             // smallest terms first
    c = mfma_h(a.p[1], b.p[0], c);
    c = mfma_h(a.p[0], b.p[0], c);
    return c;
}
// 8 fp32 (already scaled) -> hi / lo pieces
__device__ __forceinline__ H2 split2(const f32x4& lo, const f32x4& hi) {
    H2 r;
    DH_UNROLL for (int j = 0; j < 4; ++j) {
        f32x2 x;
        x[0] = j < 2 ? lo[2 * j] : hi[2 * j - 4];
        x[1] = j < 2 ? lo[2 * j + 1] : hi[2 * j - 3];
        const unsigned h = pack_f16x2(x);
        r.p[0][j] = h;
        r.p[1][j] = pack_f16x2(resid_f16x2(x, h));
    }
    return r;
}

// ---------------------------------------------------------------- power-of-two scales
// S = 2^k with S * m in [2^T, 2^(T+1)) for a maximum m given by its fp32 bits; k clamped to +-60 (m = 0, inf, nan stay finite)
__host__ __device__ inline unsigned pow2_scale_bits(unsigned maxbits, int T) {
    const int e = (int)((maxbits >> 23) & 0xffu);          // biased exponent: floor(log2 m) = e - 127
    int k = T - (e - 127);
    k = k < -60 ? -60 : (k > 60 ? 60 : k);
    return (unsigned)(127 + k) << 23;
}
__host__ __device__ inline unsigned pow2_inv_bits(unsigned sbits) { return (254u << 23) - sbits; }
constexpr int H2_WT = 3;            // weights: scaled maximum in [8, 16)
constexpr int H2_AT = 8;            // dynamically scaled operands: scaled maximum in [256, 512)
__device__ __forceinline__ float wscale_from_bits(unsigned wabs) { return __builtin_bit_cast(float, pow2_scale_bits(wabs, H2_WT)); }
__device__ __forceinline__ float winv_from_bits(unsigned wabs) { return __builtin_bit_cast(float, pow2_inv_bits(pow2_scale_bits(wabs, H2_WT))); }

struct TileScale { float S, inv; unsigned maxbits; };
__device__ __forceinline__ TileScale scale_for_max(float m) {
    TileScale t;
    t.maxbits = __builtin_bit_cast(unsigned, m);
    const unsigned sb = pow2_scale_bits(t.maxbits, H2_AT);
    t.S = __builtin_bit_cast(float, sb);
    t.inv = __builtin_bit_cast(float, pow2_inv_bits(sb));
    return t;
}
__device__ __forceinline__ float wave_max(float m) {
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return m;
}
// largest |value| of this wave's [TM x 64] slice of a tile
__device__ __forceinline__ float acc_absmax(const f32x16 (&acc)[MT][2]) {
    float m0 = 0.f, m1 = 0.f;
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r = 0; r < 16; r += 2) { m0 = fmaxf(m0, fabsf(acc[m][t][r])); m1 = fmaxf(m1, fabsf(acc[m][t][r + 1])); }
    return wave_max(fmaxf(m0, m1));
}
// Publishing a tile's maximum: every wave writes its own to sred[wave] BEFORE a workgroup barrier, everyone reads the four AFTER it
// (the chains already have that barrier: the one that protects the LDS image against its readers).  lmax: this workgroup's running
// maximum of the tile class (an LDS word, nullptr: none) -- the weight-gradient kernel scales its operands by the per-launch
// maximum of each saved-tile class, which a workgroup posts ONCE, at its end (post_class_max), not per tile: 32 k atomics on one
// address would serialise at ~12 ns each.
__device__ __forceinline__ void tile_max_publish(float* sred, int wave, int lane, float m) { if (lane == 0) sred[wave] = m; }
__device__ __forceinline__ float tile_max_read(const float* sred) { return fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3])); }
__device__ __forceinline__ TileScale tile_scale(const float* sred, float* lmax, int tid) {
    const float m = tile_max_read(sred);
    if (lmax && tid == 0) *lmax = fmaxf(*lmax, m);
    return scale_for_max(m);
}
// workspace.h: class slot c lives 64 words (256 B) from the next
constexpr int ABSMAX_STRIDE = 64;
__device__ __forceinline__ void post_class_max(unsigned* absmax, int cls, float m) {
    atomicMax(absmax + cls * ABSMAX_STRIDE, __builtin_bit_cast(unsigned, m));
}
// accumulators -> LDS main tile, scaled
__device__ __forceinline__ void acc_to_lds_scaled(const f32x16 (&acc)[MT][2], float* xs, int wave, int lane, float S) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t) {
            float* base = xs + (m * 32 + 4 * (lane >> 5)) * LDX + acc_col(wave, t, lane);
            DH_UNROLL for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * LDX] = acc[m][t][r] * S;
        }
}

// ---------------------------------------------------------------- split-on-fetch GEMM (the chains' core), tile16.h's scheme
// Program order pinned as in gemm_rows_s: raw fp32 A fragments two k-chunks ahead, weight pieces one; the 12 MFMAs of a k-chunk
// product-major at raised priority, and after MFMA i two of the 24 split micro-steps of the NEXT k-chunk (pair j = 8 m + q:
// convert | two residuals | convert).
// SC: the image is UNSCALED and the scale st.sc is applied as the values are fetched (the small aux images)
struct SplitStateH { f32x2 r[4 * MT]; float sc; };
template <int STEP, bool SC>
__device__ __forceinline__ void split_step_h(H2 (&a)[MT], const RawA& raw, SplitStateH& st) {
    constexpr int j = STEP / 3, s = STEP % 3, m = j / 4, q = j % 4;
    if constexpr (s == 0) {
        f32x2 x;
        x[0] = q < 2 ? raw.lo[m][2 * q] : raw.hi[m][2 * q - 4];
        x[1] = q < 2 ? raw.lo[m][2 * q + 1] : raw.hi[m][2 * q - 3];
        if constexpr (SC) { x[0] *= st.sc; x[1] *= st.sc; }
        a[m].p[0][q] = pack_f16x2(x);
        st.r[j] = x;
    } else if constexpr (s == 1) {
        st.r[j] = resid_f16x2(st.r[j], a[m].p[0][q]);
    } else {
        a[m].p[1][q] = pack_f16x2(st.r[j]);
    }
}
template <int I>
__device__ __forceinline__ void mfma_step_h(f32x16 (&acc)[MT][2], const H2 (&a)[MT], const H2 (&b)[2]) {
    constexpr int pa[3] = {0, 1, 0}, pb[3] = {1, 0, 0};                      // mfma3's product order
    constexpr int p = I / (2 * MT), mt = I % (2 * MT), m = mt / 2, t = mt % 2;
    __builtin_amdgcn_s_setprio(1);
    acc[m][t] = mfma_h(a[m].p[pa[p]], b[t].p[pb[p]], acc[m][t]);
    __builtin_amdgcn_s_setprio(0);
}
template <int I, int N, bool SC>
__device__ __forceinline__ void phase_steps_h(f32x16 (&acc)[MT][2], const H2 (&ac)[MT], const H2 (&bc)[2], H2 (&an)[MT],
                                              const RawA& rn, SplitStateH& st) {
    if constexpr (I < N) {
        mfma_step_h<I>(acc, ac, bc);
        __builtin_amdgcn_sched_barrier(0);
        split_step_h<2 * I, SC>(an, rn, st);
        split_step_h<2 * I + 1, SC>(an, rn, st);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps_h<I + 1, N, SC>(acc, ac, bc, an, rn, st);
    }
}
template <int I, int N>
__device__ __forceinline__ void mfma_only_h(f32x16 (&acc)[MT][2], const H2 (&ac)[MT], const H2 (&bc)[2]) {
    if constexpr (I < N) { mfma_step_h<I>(acc, ac, bc); mfma_only_h<I + 1, N>(acc, ac, bc); }
}
template <int I, int N, bool SC>
__device__ __forceinline__ void split_only_h(H2 (&an)[MT], const RawA& rn, SplitStateH& st) {
    if constexpr (I < N) { split_step_h<I, SC>(an, rn, st); split_only_h<I + 1, N, SC>(an, rn, st); }
}
// acc[m][t] += X[TM x 16 nkc] * M.  xs: the SCALED fp32 LDS image (row stride ldx floats; SC: unscaled, `sc` applied on fetch), wp:
// packed fp16 weight pieces (NT = 8).  The result carries the product of the two scales.
template <bool SC = false>
__device__ __forceinline__ void gemm_rows_h(f32x16 (&acc)[MT][2], const float* xs, const int ldx, const int nkc,
                                            const u32x4* __restrict__ wp, const int wave, const int lane, const float sc = 1.f) {
    static_assert(MT == 2, "12 MFMAs per k-chunk = 8 pairs x 3 split steps / 2");
    const float* xrow = xs + (lane & 31) * ldx + 8 * (lane >> 5);
    const u32x4* wl = wp + (2 * wave) * 2 * 64 + lane;
    const int last = nkc - 1;
    H2 a0[MT], a1[MT];
    H2 b0[2], b1[2];
    RawA r0, r1;
    SplitStateH st;
    st.sc = sc;
    auto loadb = [&](H2 (&b)[2], int kc) {                  // past the end: clamped (a harmless re-read), no branch
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p) b[t].p[p] = wl[((kc * 8 + t) * 2 + p) * 64];
    };
    auto loada = [&](RawA& r, int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int m = 0; m < MT; ++m) {
            r.lo[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16);
            r.hi[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + kc * 16 + 4);
        }
    };
    loadb(b0, 0); loada(r0, 0); loada(r1, 1);
    split_only_h<0, 24, SC>(a0, r0, st);
    _Pragma("unroll 1") for (int kc = 0; kc + 1 < nkc; kc += 2) {
        loadb(b1, kc + 1); loada(r0, kc + 2);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps_h<0, 12, SC>(acc, a0, b0, a1, r1, st);
        loadb(b0, kc + 2); loada(r1, kc + 3);
        __builtin_amdgcn_sched_barrier(0);
        phase_steps_h<0, 12, SC>(acc, a1, b1, a0, r0, st);
    }
    if (nkc & 1) mfma_only_h<0, 12>(acc, a0, b0);
}

// 64-wide "aux" output from the scaled main image (tile16.h gemm_auxout_s)
__device__ __forceinline__ void gemm_auxout_h(f32x16 (&acc2)[AUX_NTW], const float* xs, const int nkc,
                                              const u32x4* __restrict__ wp, const int wave, const int lane) {
    const float* xrow = xs + (32 * aux_mtile(wave) + (lane & 31)) * LDX + 8 * (lane >> 5);
    const u32x4* wl = wp + aux_ntile(wave, 0) * 2 * 64 + lane;
    const int last = nkc - 1;
    auto fetch = [&](H2& a, H2 (&b)[AUX_NTW], int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p) b[t].p[p] = wl[((kc * 2 + t) * 2 + p) * 64];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + kc * 16);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + kc * 16 + 4);
        a = split2(lo, hi);
    };
    H2 a0, a1, b0[AUX_NTW], b1[AUX_NTW];
    fetch(a0, b0, 0);
    _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
        fetch(a1, b1, kc + 1);
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma3(a0, b0[t], acc2[t]);
        if (kc + 1 < nkc) {
            fetch(a0, b0, kc + 2);
            DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) acc2[t] = mfma3(a1, b1[t], acc2[t]);
        }
    }
}

}  // namespace dh
