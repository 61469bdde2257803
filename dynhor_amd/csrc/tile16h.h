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
// The split costs 2 vector ops per value (v_cvt_pk_f16_f32, 2 x v_fma_mix_f32, v_cvt_pk_f16_f32 per pair; bf16x3: 4.5).
// The LDS activation image of the tile-resident chains holds the PIECES: two fp16 planes [TM x 256] (row stride LDH halves), 2 x
// 2 bytes per value = the 4 bytes of the fp32 image they replace, so two workgroups per CU still fit -- which three bf16 planes did
// not (tile16.h).  The wave that PRODUCES a value splits it once in its epilogue (acc_to_lds_split: the tile's scale is known there,
// lds_handoff), and the GEMM loop is ds_read_b128 of ready MFMA operands + weight loads + MFMAs: none of the split-on-fetch form's
// 32 vector ops per k-chunk, redone by each of the four waves.  Only the 48-wide aux images stay fp32 and are split as they are
// fetched (gemm_rows_aux_h: three k-chunks).
// Packed weights (pack.hip packh_kernel): f16x8 index ((kc*NT + nt)*2 + piece)*64 + lane holds
//   S_w M[k = 16 kc + 8 (lane>>5) + s][n = 32 nt + (lane&31)], s = 0..7  (piece 0 = hi, 1 = lo).
#pragma once
#include "tile16.h"
#include "workspace.h"

namespace dh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
struct H2 { u32x4 p[2]; };
__device__ __forceinline__ unsigned pack_f16x2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v)); }
// v - float(h.lo), v - float(h.hi): one v_fma_mix_f32 each, the f16 half read in place (exact: |v - h| <= ulp_f16 / 2)
__device__ __forceinline__ f32x2 resid_f16x2(f32x2 v, unsigned h) {
    f32x2 r;
#ifdef H2_VISIBLE_RESID
    // (development: the same residual from instructions the compiler sees -- its hazard recognizer does not look inside inline asm)
    const f16x2v hv = __builtin_bit_cast(f16x2v, h);
    r[0] = v[0] - (float)hv[0];
    r[1] = v[1] - (float)hv[1];
#else
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(h), "v"(v[0]));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(h), "v"(v[1]));
#endif
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
// maximum over the wave's 64 lanes, valid on EVERY lane.  Round 5: six DPP steps on the vector ALU (row_shr 1 / 2 / 4 / 8, row_bcast 15 /
// 31: the classic wave64 reduction, the result lands in lane 63) + one v_readlane, instead of six __shfl_xor = six dependent LDS round
// trips (ds_bpermute, ~150 cycles each): the phase stamps of the fp16 chains showed 2,300 cycles per layer between the epilogue and
// barrier 1 of the image hand-off (profiles/r05_chain_phase_stamps.json).  Inputs are >= 0 (absolute values): 0 is the identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float wave_max_step(float m) {
    // lanes without a source (and rows outside ROW_MASK) read the identity 0.0f
    const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m), CTRL, ROW_MASK, 0xf, false);
    return fmaxf(m, __builtin_bit_cast(float, o));
}
__device__ __forceinline__ float wave_max(float m) {
#ifdef DH_WAVE_MAX_SHFL                      // (development: rounds 1-4's form, for the A/B of profiles/r05_ab_chain_experiments.json)
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    return m;
#endif
    m = wave_max_step<0x111, 0xf>(m);        // row_shr:1
    m = wave_max_step<0x112, 0xf>(m);        // row_shr:2
    m = wave_max_step<0x114, 0xf>(m);        // row_shr:4
    m = wave_max_step<0x118, 0xf>(m);        // row_shr:8   -> lane 15 of every 16-lane row holds the row's maximum
    m = wave_max_step<0x142, 0xa>(m);        // row_bcast:15 into rows 1 and 3
    m = wave_max_step<0x143, 0xc>(m);        // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63));
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
// (workspace.h ABSMAX_STRIDE: class slot c lives 64 words = 256 B from the next)
__device__ __forceinline__ void post_class_max(unsigned* absmax, int cls, float m) {
    atomicMax(absmax + cls * ABSMAX_STRIDE, __builtin_bit_cast(unsigned, m));
}
// ---------------------------------------------------------------- the piece-plane LDS image
constexpr int LDH = 264;                 // row stride of a plane in halves: 528 B = 132 dwords, 132 % 64 == 4 like LDX -> conflict-free b128 reads
constexpr int PLANE_H = TM * LDH;        // halves per plane; the image is hi plane, then lo plane
constexpr int IMG_H = 2 * PLANE_H;
// accumulators -> scaled, split, into the two planes.  A lane holds 16 rows of ONE column per accumulator: a converted pair is two
// rows of that column, written as two 16-bit stores per plane (ds_write_b16 / ds_write_b16_d16_hi).
// (Round 5 built the 32-bit form -- lanes 2i / 2i + 1 trade halves by one DPP exchange + one v_perm_b32 per word, pairs (r, r + 4) so
// that the even and odd lanes' rows lie 32 banks apart: 64 ds_write_b32 instead of 128 ds_write_b16 per layer, bit-identical planes --
// and measured it at the same stage times: the 128 extra vector operations cost what the 64 writes save.  Removed;
// profiles/r05_ab_chain_experiments.json.)
__device__ __forceinline__ void acc_to_lds_split(const f32x16 (&acc)[MT][2], _Float16* img, int wave, int lane, float S) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t) {
            _Float16* base = img + (m * 32 + 4 * (lane >> 5)) * LDH + acc_col(wave, t, lane);
            DH_UNROLL for (int r = 0; r < 16; r += 2) {
                f32x2 x;
                x[0] = acc[m][t][r] * S;
                x[1] = acc[m][t][r + 1] * S;
                const unsigned h = pack_f16x2(x);
                const unsigned l = pack_f16x2(resid_f16x2(x, h));
                const f16x2v hv = __builtin_bit_cast(f16x2v, h), lv = __builtin_bit_cast(f16x2v, l);
                _Float16* q = base + ((r & 3) + 8 * (r >> 2)) * LDH;
                q[0] = hv[0]; q[LDH] = hv[1];
                q[PLANE_H] = lv[0]; q[PLANE_H + LDH] = lv[1];
#ifdef H_PROBE_EXTRA_WRITES                              // timing probe (results unchanged): the lo plane's 64 ds_write_b16 per layer issued a second time
                asm volatile("ds_write_b16 %0, %1 offset:%2\n\tds_write_b16_d16_hi %0, %1 offset:%3" ::"v"((unsigned)(uintptr_t)q), "v"(l), "n"(PLANE_H * 2), "n"((PLANE_H + LDH) * 2) : "memory");
#endif
            }
        }
}
// ---------------------------------------------------------------- split micro-steps (the weight-gradient kernel deals them between its MFMAs)
// pair j = 8 m + q of the 8 values of an operand fragment: convert | two residuals | convert.
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

// ---------------------------------------------------------------- MFMA issue order of one k-chunk
// the 12 MFMAs of a k-chunk (two m-tiles x two n-tiles x three products) product-major.  (Rounds 2-4 raised the wave priority around
// every MFMA -- a gain for the six-product bf16 chains; in the three-product chains it costs 0.5-1 %, and 10 % in a synthetic two-wave
// loop: scripts/micro/mfma_issue_micro.hip, profiles/r05_ab_chain_experiments.json -- removed in round 5.)
template <int I>
__device__ __forceinline__ void mfma_step_h(f32x16 (&acc)[MT][2], const H2 (&a)[MT], const H2 (&b)[2]) {
    constexpr int pa[3] = {0, 1, 0}, pb[3] = {1, 0, 0};                      // mfma3's product order
    constexpr int p = I / (2 * MT), mt = I % (2 * MT), m = mt / 2, t = mt % 2;
    acc[m][t] = mfma_h(a[m].p[pa[p]], b[t].p[pb[p]], acc[m][t]);
}
template <int I, int N>
__device__ __forceinline__ void mfma_only_h(f32x16 (&acc)[MT][2], const H2 (&ac)[MT], const H2 (&bc)[2]) {
    if constexpr (I < N) { mfma_step_h<I>(acc, ac, bc); mfma_only_h<I + 1, N>(acc, ac, bc); }
}

// ---------------------------------------------------------------- saved tiles and weights through buffer descriptors
// A chain touches up to four saved-tile streams per layer plus the weight stream; as flat pointers each costs 64-bit vector address
// pairs (one per 4 KB of reach), and the backward kernels spilled.  Through a buffer descriptor the base is scalar (four SGPRs built
// per tile and layer with scalar arithmetic) and ONE 32-bit lane offset serves every stream.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t tile_rsrc(const float* tile) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, TILE_F * 4, 0x00020000);
}
__device__ __forceinline__ int tile_loff(int wave, int lane) { return (wave * MT * 8 * 64 + lane) * 16; }
// float4 idx = (m*2 + t)*4 + r4 of this lane's share of a native tile (tile.h); non-temporal like DH_TILE_LD / DH_TILE_ST
__device__ __forceinline__ f32x4 tile_ld(rsrc_t r, int loff, int idx) {
    const int off = idx * 1024;
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, loff + (off & 4095), off & ~4095, 2));
}
// STORES carry their whole offset in the vector offset + immediate, never in a scalar soffset: with a REGISTER soffset hipcc (ROCm
// 7.2) schedules a vector write to the store's data registers directly behind a buffer_store_dwordx4 -- it only guards the
// immediate-soffset form -- and on gfx950 the store then sends the new values (measured: a tile class wrong by O(1), differently on
// every run; located with scripts/cmp_libs.py, record in profiles/r04_ab_chain_io.json).
__device__ __forceinline__ void tile_st(rsrc_t r, int loff, int idx, const f32x4& v) {
    // (cache policy 2 = nt.  Round 5 re-measured plain write-back / sc1 / sc0 stores: 2 % slower, the tiles evict the weights from L2)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, loff + idx * 1024, 0, 2);
}
__device__ __forceinline__ void acc_store_native_b(const f32x16 (&acc)[MT][2], rsrc_t r, int loff) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
                v[0] = acc[m][t][4 * r4 + 0]; v[1] = acc[m][t][4 * r4 + 1]; v[2] = acc[m][t][4 * r4 + 2]; v[3] = acc[m][t][4 * r4 + 3];
                tile_st(r, loff, (m * 2 + t) * 4 + r4, v);
            }
}
__device__ __forceinline__ void acc_load_native_b(f32x16 (&acc)[MT][2], rsrc_t r, int loff) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 v = tile_ld(r, loff, (m * 2 + t) * 4 + r4);
                acc[m][t][4 * r4 + 0] = v[0]; acc[m][t][4 * r4 + 1] = v[1]; acc[m][t][4 * r4 + 2] = v[2]; acc[m][t][4 * r4 + 3] = v[3];
            }
}
// one m-slab (8 float4: t, r4) of a saved tile: what an epilogue holds in flight per stream
struct Slab { f32x4 v[8]; };
__device__ __forceinline__ void slab_ld(Slab& s, rsrc_t r, int loff, int m) {
    DH_UNROLL for (int i = 0; i < 8; ++i) s.v[i] = tile_ld(r, loff, m * 8 + i);
}
__device__ __forceinline__ rsrc_t weight_rsrc(const u32x4* wp) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wp), 0, 0x40000000, 0x00020000);
}

// ---------------------------------------------------------------- piece-plane GEMM (the chains' core)
// acc[m][t] += X[TM x 16 nkc] * M from the piece planes: per k-chunk a wave issues 4 ds_read_b128 (its two m-tiles, hi and lo) and 4
// weight loads (two n-tiles, hi and lo; L2), then the 12 MFMAs of the chunk product-major at raised priority.  A fragments one chunk
// ahead, weight pieces two chunks ahead in four rotating buffers.  nkc even, >= 6.  The result carries the product of the two scales.
// `pre`: called once, at the top of the THIRD-LAST chunk, right after the last weight load of the GEMM has been issued -- the place
// for the epilogue's first saved-tile loads: vector-memory loads return in order, so a tile load issued any earlier would stand
// between the remaining weight loads and their consumers, and one issued in the epilogue is exposed in full.  Three chunks = 36
// MFMAs cover it (a fourth would need all four weight buffers live at once: the two-stream kernels spilled).
struct NoPre { __device__ __forceinline__ void operator()() const {} };
// LEAN: two weight buffers one chunk ahead instead of four two ahead (32 registers less; the kernels with two saved-tile input
// streams per epilogue, whose register file is full), `pre` at the top of the last chunk.
template <bool LEAN = false, class Pre = NoPre>
__device__ __forceinline__ void gemm_rows_hp(f32x16 (&acc)[MT][2], const _Float16* img, const int nkc,
                                             const u32x4* __restrict__ wp, const int wave, const int lane, Pre&& pre = Pre()) {
    static_assert(MT == 2, "");
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const rsrc_t wr = weight_rsrc(wp);
    const int woff = ((2 * wave) * 2 * 64 + lane) * 16;
    auto loadb = [&](H2 (&b)[2], int kc) {
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p)
                b[t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, kc * (8 * 2 * 64 * 16), 0);
    };
    auto loada = [&](H2 (&a)[MT], int kc) {
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int p = 0; p < 2; ++p)
                a[m].p[p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE_H + m * 32 * LDH + kc * 16);
    };
    H2 a0[MT], a1[MT];
    if constexpr (LEAN) {
        H2 b0[2], b1[2];
        loadb(b0, 0); loada(a0, 0);
        _Pragma("unroll 1") for (int kc = 0; kc < nkc; kc += 2) {
            loadb(b1, kc + 1); loada(a1, kc + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_only_h<0, 12>(acc, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (kc + 2 < nkc) { loadb(b0, kc + 2); loada(a0, kc + 2); }
            else pre();
            __builtin_amdgcn_sched_barrier(0);
            mfma_only_h<0, 12>(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    H2 bA[2], bB[2], bC[2], bD[2];
    // chunks c, c+1 from (b0, b1); (b2, b3) <- c+2, c+3
    auto step2 = [&](int c, H2 (&b0)[2], H2 (&b1)[2], H2 (&b2)[2], H2 (&b3)[2]) {
        loadb(b2, c + 2); loada(a1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        loadb(b3, c + 3); loada(a0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // the last four chunks: the remaining weights, then the caller's prefetch
    auto tail4 = [&](int c, H2 (&b0)[2], H2 (&b1)[2], H2 (&b2)[2], H2 (&b3)[2]) {
        loadb(b2, c + 2); loada(a1, c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        loadb(b3, c + 3);
        pre();
        loada(a0, c + 2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        loada(a1, c + 3);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a0, b2);
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a1, b3);
        __builtin_amdgcn_sched_barrier(0);
    };
    loadb(bA, 0); loadb(bB, 1); loada(a0, 0);
    int kc = 0;
    _Pragma("unroll 1") for (; kc + 8 <= nkc; kc += 4) {
        step2(kc, bA, bB, bC, bD);
        step2(kc + 2, bC, bD, bA, bB);
    }
    if (nkc - kc == 6) {
        step2(kc, bA, bB, bC, bD);
        tail4(kc + 2, bC, bD, bA, bB);
    } else {
        tail4(kc, bA, bB, bC, bD);
    }
}

// acc[m][t] += sc * X[TM x 48] * M from the small fp32 aux image (row stride LDA; embedding / colour extras, unscaled in LDS): three
// k-chunks, split as they are fetched.  All three chunks' weight pieces are requested up front.
__device__ __forceinline__ void gemm_rows_aux_h(f32x16 (&acc)[MT][2], const float* xs, const u32x4* __restrict__ wp,
                                                const int wave, const int lane, const float sc) {
    static_assert(AUX_KC == 3, "");
    const float* xrow = xs + (lane & 31) * LDA + 8 * (lane >> 5);
    const rsrc_t wr = weight_rsrc(wp);
    const int woff = ((2 * wave) * 2 * 64 + lane) * 16;
    H2 b[AUX_KC][2];
    asm volatile("" ::: "memory");           // (keeps these loads from being hoisted above a main GEMM that precedes this one)
    DH_UNROLL for (int kc = 0; kc < AUX_KC; ++kc)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p)
                b[kc][t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, kc * (8 * 2 * 64 * 16), 0);
    DH_UNROLL for (int kc = 0; kc < AUX_KC; ++kc) {
        H2 a[MT];
        DH_UNROLL for (int m = 0; m < MT; ++m) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(xrow + m * 32 * LDA + kc * 16);
            const f32x4 hi = *reinterpret_cast<const f32x4*>(xrow + m * 32 * LDA + kc * 16 + 4);
            a[m] = split2(lo * sc, hi * sc);
        }
        mfma_only_h<0, 12>(acc, a, b[kc]);
    }
}

// 64-wide "aux" output from the piece planes (one m-tile per wave)
__device__ __forceinline__ void gemm_auxout_hp(f32x16 (&acc2)[AUX_NTW], const _Float16* img, const int nkc,
                                               const u32x4* __restrict__ wp, const int wave, const int lane) {
    const _Float16* xrow = img + (32 * aux_mtile(wave) + (lane & 31)) * LDH + 8 * (lane >> 5);
    const rsrc_t wr = weight_rsrc(wp);
    const int woff = (aux_ntile(wave, 0) * 2 * 64 + lane) * 16;
    const int last = nkc - 1;
    auto fetch = [&](H2& a, H2 (&b)[AUX_NTW], int kc) {
        kc = kc < last ? kc : last;
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p)
                b[t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, kc * (2 * 2 * 64 * 16), 0);
        DH_UNROLL for (int p = 0; p < 2; ++p) a.p[p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE_H + kc * 16);
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

// per-point dot of the image rows (hi + lo) with a 256-vector: 256 / TM threads per point (point = tid / TPP), result valid on
// every thread of the group
__device__ __forceinline__ float row_dot256_hp(const _Float16* img, const float* __restrict__ w, int tid) {
    constexpr int TPP_ = 256 / TM, SEG = 256 / TPP_;
    const int p = tid / TPP_, part = tid % TPP_;
    const f16x8* xh = reinterpret_cast<const f16x8*>(img + p * LDH + part * SEG);
    const f16x8* xl = reinterpret_cast<const f16x8*>(img + PLANE_H + p * LDH + part * SEG);
    const f32x4* wr = reinterpret_cast<const f32x4*>(w + part * SEG);
    float s = 0.f;
    DH_UNROLL for (int i = 0; i < SEG / 8; ++i) {
        const f16x8 h = xh[i], l = xl[i];
        const f32x4 b0 = wr[2 * i], b1 = wr[2 * i + 1];
        DH_UNROLL for (int j = 0; j < 4; ++j) s = fmaf((float)h[j] + (float)l[j], b0[j], s);
        DH_UNROLL for (int j = 0; j < 4; ++j) s = fmaf((float)h[4 + j] + (float)l[4 + j], b1[j], s);
    }
    DH_UNROLL for (int off = 1; off < TPP_; off <<= 1) s += __shfl_xor(s, off);
    return s;
}

}  // namespace dh
