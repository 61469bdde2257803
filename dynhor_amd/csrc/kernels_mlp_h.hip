// The five tile-resident MLP chains in the two-piece fp16 arithmetic (DH_ARITH_SPLIT_F16; round 4): colour forward, input-gradient
// (reverse) chain, colour backward, tangent chain, SDF backward.  Same saved tiles, outputs and tile-partial sums as the
// `<name>_s_kernel` forms in kernels_mlp.hip / kernels_mlp_bwd.hip (whose headers hold the maths); what differs is the GEMM core
// (tile16h.h: three v_mfma_f32_32x32x16_f16 per fp32 product instead of six bf16 ones), the LDS image, the tile I/O and the operand
// scaling:
//   * the LDS main image holds the PIECES of S x: two fp16 planes, written once by the wave that produces a value (lds_handoff ->
//     acc_to_lds_split), read as ready MFMA operands; S = the power of two that puts the tile's own maximum into [256, 512): every
//     wave publishes the maximum of its slice before the barrier the chains already have, everyone derives S behind it;
//   * the small aux images stay unscaled fp32 in LDS and are scaled and split as they are fetched (gemm_rows_aux_h);
//   * a GEMM's result carries S * S_w (S_w: the linear's weight scale, layout.h PACKH.wabs): its reciprocal -- a power of two,
//     so the product is exact -- multiplies the accumulator in the epilogue's first operation;
//   * saved tiles and weights go through buffer descriptors (tile_rsrc / tile_ld / tile_st, weight_rsrc): scalar bases, one 32-bit
//     lane offset for every stream; the one-stream kernels request their epilogue's first m-slab inside the GEMM's tail;
//   * the per-launch maximum of every saved-tile class the weight-gradient kernel reads (workspace.h absmax) is kept as a running
//     maximum per workgroup and posted with one atomicMax per class at the workgroup's end; per-tile maxima of the heavy-tailed
//     classes go to workspace.h tmax.
#include "tile.h"
#include "kernels.h"
#include "mlp_common.h"
#include "tile16h.h"
#include "workspace.h"
#include "stamps.h"

DH_STAMP_READER(dh_dev_read_stamps_h)

namespace dh {

#ifndef H2_LEAN2
#define H2_LEAN2 true            // development macro: the GEMM form of the two kernels with two saved-tile input streams (tile16h.h)
#endif

// tile partial-sum slots (workspace.h: tpart [nt][N_TILE_PART][256]) -- as kernels_mlp_bwd.hip
enum : int { TP_SDF_B0 = 0, TP_SDF_B8 = 8, TP_W8ROW0_T = 9, TP_W8ROW0_S = 10, TP_SCAL = 11, TP_COL_B0 = 12,
             TP_COL_W4 = 16, TP_COL_B4 = 19 };

struct SdfHPtrs {
    const u32x4* main[N_SDF];
    const u32x4* aux[N_SDF];
    const u32x4* rev[N_SDF];
    const u32x4* revaux[N_SDF];
    const float* bias[N_SDF];
    const float* w8row0;
    const float* b8_0;
    const unsigned* wabs;          // [0..8]
};
static inline SdfHPtrs make_sdfh_ptrs(const float* packed) {
    SdfHPtrs P;
    for (int l = 0; l < N_SDF; ++l) {
        P.main[l] = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_fwd_main[l]);
        P.aux[l] = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_fwd_aux[l]);
        P.rev[l] = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_rev_main[l]);
        P.revaux[l] = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_rev_aux[l]);
        P.bias[l] = packed + PACK.sdf_bias[l];
    }
    P.w8row0 = packed + PACK.sdf_w8row0;
    P.b8_0 = packed + PACK.sdf_b8_0;
    P.wabs = reinterpret_cast<const unsigned*>(packed + PACKH.wabs);
    return P;
}
struct ColHPtrs {
    const u32x4* main[4];
    const u32x4* rev[4];
    const u32x4* aux;
    const u32x4* revaux;
    const float* bias[4];
    const float* w4;
    const float* b4;
    const unsigned* wabs;          // [0..3] (colour linears: PACKH.wabs + N_SDF)
};
static inline ColHPtrs make_colh_ptrs(const float* packed) {
    ColHPtrs C;
    for (int l = 0; l < 4; ++l) {
        C.main[l] = reinterpret_cast<const u32x4*>(packed + PACKH.col_fwd_main[l]);
        C.rev[l] = reinterpret_cast<const u32x4*>(packed + PACKH.col_rev_main[l]);
        C.bias[l] = packed + PACK.col_bias[l];
    }
    C.aux = reinterpret_cast<const u32x4*>(packed + PACKH.col_fwd_aux0);
    C.revaux = reinterpret_cast<const u32x4*>(packed + PACKH.col_rev_aux0);
    C.w4 = packed + PACK.col_w4;
    C.b4 = packed + PACK.col_b4;
    C.wabs = reinterpret_cast<const unsigned*>(packed + PACKH.wabs) + N_SDF;
    return C;
}

// shared LDS bookkeeping of the H2 chains: the four waves' maxima and the workgroup's running class maxima
struct HScratch { float sred[4]; float sred2[4]; float lmax[12]; };      // sred2: a second set for the aux images' maxima
__device__ __forceinline__ void hs_init(HScratch& h, int tid) { if (tid < 12) h.lmax[tid] = 0.f; }
// the image hand-off of every layer: publish this wave's maximum, wait until every wave has left the previous image, derive the
// tile's scale, write the scaled image.  Returns the scale (S, 1 / S).
// extra_max / extra_lds: a second operand that will share the accumulator (and hence the scale) of the GEMM that reads this image
// tslot: where this tile's maximum goes for the weight-gradient kernel (workspace.h tmax), or nullptr
__device__ __forceinline__ TileScale lds_handoff(const f32x16 (&acc)[MT][2], _Float16* smain, HScratch& hs, float* lmax, int tid,
                                                 int wave, int lane, float extra_max = 0.f, const float* extra_lds = nullptr,
                                                 unsigned* tslot = nullptr, int sit = -1, int slayer = 0) {
    (void)sit; (void)slayer;                          // (-DDH_STAMPS: phase stamps 3..6 of layer slayer; stamps.h)
    tile_max_publish(hs.sred, wave, lane, acc_absmax(acc));
    DH_STAMP(sit, slayer, 3);
    __syncthreads();
    DH_STAMP(sit, slayer, 4);
    const float m = tile_max_read(hs.sred);
    if (tid == 0) {
        if (lmax) *lmax = fmaxf(*lmax, m);
        if (tslot) *tslot = __builtin_bit_cast(unsigned, m);
    }
    if (extra_lds) extra_max = fmaxf(extra_max, *extra_lds);
    const TileScale ts = scale_for_max(fmaxf(m, extra_max));
    acc_to_lds_split(acc, smain, wave, lane, ts.S);
    DH_STAMP(sit, slayer, 5);
    __syncthreads();
    DH_STAMP(sit, slayer, 6);
    return ts;
}

#ifdef DH_STAMPS
// (diagnostic build only) the colour forward GEMM as a plain unrolled loop with four weight buffers that stamps the start of every
// k-chunk of layer 1 into stamp layers 5 / 6: where inside a GEMM do the cycles go?
template <int KC>
__device__ __forceinline__ void gemm_stamped_steps(f32x16 (&acc)[MT][2], H2 (&a)[2][MT], H2 (&b)[4][2], const _Float16* xrow, rsrc_t wr, int woff, int it, int wave,
                                                   int lane, bool on) {
    if constexpr (KC < 16) {
        if (on) DH_STAMP(it, 5 + (KC >> 3), KC & 7);
        if constexpr (KC + 3 < 16) {
            DH_UNROLL for (int t = 0; t < 2; ++t)
                DH_UNROLL for (int p = 0; p < 2; ++p)
                    b[(KC + 3) & 3][t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, (KC + 3) * (8 * 2 * 64 * 16), 0);
        }
        if constexpr (KC + 1 < 16) {
            DH_UNROLL for (int m = 0; m < MT; ++m)
                DH_UNROLL for (int p = 0; p < 2; ++p)
                    a[(KC + 1) & 1][m].p[p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE_H + m * 32 * LDH + (KC + 1) * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_only_h<0, 12>(acc, a[KC & 1], b[KC & 3]);
        __builtin_amdgcn_sched_barrier(0);
        gemm_stamped_steps<KC + 1>(acc, a, b, xrow, wr, woff, it, wave, lane, on);
    }
}
__device__ __forceinline__ void gemm_rows_stamped(f32x16 (&acc)[MT][2], const _Float16* img, const u32x4* __restrict__ wp, int wave, int lane, int it, bool on) {
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const rsrc_t wr = weight_rsrc(wp);
    const int woff = ((2 * wave) * 2 * 64 + lane) * 16;
    H2 a[2][MT], b[4][2];
    DH_UNROLL for (int kc = 0; kc < 3; ++kc)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int p = 0; p < 2; ++p)
                b[kc][t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, kc * (8 * 2 * 64 * 16), 0);
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int p = 0; p < 2; ++p) a[0][m].p[p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE_H + m * 32 * LDH);
    gemm_stamped_steps<0>(acc, a, b, xrow, wr, woff, it, wave, lane, on);
    if (on) DH_STAMP(it, 7, 0);
}
#endif

// ------------------------------------------------------------------------------------------------ colour forward
__global__ __launch_bounds__(256, 2) void color_fwd_h_kernel(ColHPtrs C, const float* __restrict__ pts, const float* __restrict__ dirs,
                                                            int n_per_ray, const float* __restrict__ normals,
                                                            const float* __restrict__ feat, int64_t npts,
                                                            float* __restrict__ color, float* __restrict__ cact,
                                                            float* __restrict__ caux, int save, unsigned* __restrict__ absmax) {
    __shared__ __attribute__((aligned(16))) _Float16 smain[IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    __shared__ HScratch hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM;
    hs_init(hs, tid);
    float winv[4];
    DH_UNROLL for (int l = 0; l < 4; ++l) winv[l] = winv_from_bits(C.wabs[l]);
    int it = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        (void)it;
        if (tid < TM) {                                  // (TM == 64: exactly wave 0)
            const int64_t gp = tile * TM + tid;
            float* row = saux + tid * LDA;
            float mx = 1.f;                              // sin / cos of the view embedding
            // The view embedding depends on the RAY only.  Where the tile's 64 points belong to one ray (training: 128 samples per ray)
            // twelve lanes evaluate its twelve sincosf once and every row copies them through the wave -- same arguments, same bits
            // (round 6: per point they cost this wave ~12 x 64 evaluations' worth of vector-ALU time per tile).
            const int64_t first = tile * TM, last = first + TM - 1 < npts ? first + TM - 1 : npts - 1;
            const bool one_ray = first / n_per_ray == last / n_per_ray;       // (wave-uniform)
            float esin = 0.f, ecos = 0.f;
            if (one_ray && lane < 12) sincosf(dirs[(first / n_per_ray) * 3 + lane % 3] * (float)(1 << (lane / 3)), &esin, &ecos);
            if (gp < npts) {
                const int64_t ray = gp / n_per_ray;
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = dirs[ray * 3 + c], x = pts[gp * 3 + c], nn = normals[gp * 3 + c];
                    row[c] = x;
                    row[3 + c] = d;
                    DH_UNROLL for (int k = 0; k < 4; ++k) {
                        float s, co;
                        if (one_ray) {
                            s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, esin), 3 * k + c));
                            co = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ecos), 3 * k + c));
                        } else {
                            sincosf(d * (float)(1 << k), &s, &co);
                        }
                        row[6 + 6 * k + c] = s;
                        row[6 + 6 * k + 3 + c] = co;
                    }
                    row[30 + c] = nn;
                    mx = fmaxf(mx, fmaxf(fabsf(x), fmaxf(fabsf(d), fabsf(nn))));
                }
            } else {
                DH_UNROLL for (int c = 0; c < CAUX; ++c) row[c] = 0.f;
            }
            DH_UNROLL for (int c = CAUX; c < LDA; ++c) row[c] = 0.f;
            mx = wave_max(mx);
            if (lane == 0) hs.sred2[0] = mx;
        }
        f32x16 acc[MT][2];
        acc_load_native_b(acc, tile_rsrc(feat + tile * TILE_F), loff);
        // (the barrier inside also publishes saux; the previous tile ended with one, so the images are free).  The extras [p,
        // embed(view), n] share layer 0's accumulator with feat: one scale for both, from the larger of the two maxima
        TileScale ts = lds_handoff(acc, smain, hs, save ? &hs.lmax[4] : nullptr, tid, wave, lane, 0.f, &hs.sred2[0]);
        if (save && tid == 0) hs.lmax[5] = fmaxf(hs.lmax[5], hs.sred2[0]);      // the extras' class maximum (sred2[0]: rewritten a tile away)
        if (save) aux_lds_to_native(saux, caux + tile * AUXT_F, wave, lane);
        for (int l = 0; l < 4; ++l) {
            DH_STAMP(it, l, 0);
            acc_zero(acc);
#ifdef DH_STAMPS
            gemm_rows_stamped(acc, smain, C.main[l], wave, lane, it, l == 1);
#else
            gemm_rows_hp(acc, smain, 16, C.main[l], wave, lane);
#endif
            const float inv = ts.inv * winv[l];
            if (l == 0) gemm_rows_aux_h(acc, saux, C.aux, wave, lane, ts.S);
            DH_STAMP(it, l, 1);
            const float b0 = C.bias[l][acc_col(wave, 0, lane)], b1 = C.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return fmaxf(fmaf(v, inv, t ? b1 : b0), 0.f); });
            if (save) acc_store_native_b(acc, tile_rsrc(cact + ((int64_t)l * ntiles + tile) * TILE_F), loff);
            DH_STAMP(it, l, 2);
            ts = lds_handoff(acc, smain, hs, save ? &hs.lmax[l] : nullptr, tid, wave, lane, 0.f, nullptr, nullptr, it, l);
        }
        DH_STAMP(it, 4, 0);
        const int64_t gp = tile * TM + tid / TPP;
        DH_UNROLL for (int j = 0; j < 3; ++j) {
            const float raw = fmaf(row_dot256_hp(smain, C.w4 + j * 256, tid), ts.inv, C.b4[j]);
            if (tid % TPP == 0 && gp < npts) color[gp * 3 + j] = 1.f / (1.f + __expf(-raw));
        }
        __syncthreads();
    }
    if (save && absmax) {
        if (tid < 4) post_class_max(absmax, ABSMAX_CACT + tid, hs.lmax[tid]);
        if (tid == 4) post_class_max(absmax, ABSMAX_FEAT, hs.lmax[4]);
        if (tid == 5) post_class_max(absmax, ABSMAX_CAUX, hs.lmax[5]);
    }
}

// ------------------------------------------------------------------------------------------------ n = d sdf / d x (reverse chain)
__global__ __launch_bounds__(256, 2) void sdf_grad_h_kernel(SdfHPtrs P, const float* __restrict__ pts, int64_t npts,
                                                           const float* __restrict__ act, float* __restrict__ asave,
                                                           float* __restrict__ normals, int save, float* __restrict__ gesave,
                                                           unsigned* __restrict__ absmax) {
    __shared__ __attribute__((aligned(16))) _Float16 smain[IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    __shared__ HScratch hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM;
    hs_init(hs, tid);
    // RANGE WATCH of the forward chain (chain_t.hip carries its softplus activations at the constant scale H2_XS in fp16; it has no
    // register left to watch them itself): this kernel reads every activation tile the forward saved -- as fp32, valid even where a
    // piece overflowed -- and keeps their maximum: the class maximum of `act` for the weight-gradient kernel (workspace.h ABSMAX_ACT)
    // and what dh_range_words exposes.  One v_max3_f32 per two values.
    float hmax = 0.f;
    int it = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        (void)it;
        f32x16 acc[MT][2];
        f32x16 ge[AUX_NTW];
        aux_zero(ge);
        TileScale ts;
        // a_7 = W8[0,:] * sigma'(z_7)
        {
            const float w0 = P.w8row0[acc_col(wave, 0, lane)], w1 = P.w8row0[acc_col(wave, 1, lane)];
            acc_load_native_b(acc, tile_rsrc(act + ((int64_t)7 * ntiles + tile) * TILE_F), loff);
            DH_UNROLL for (int m = 0; m < MT; ++m) DH_UNROLL for (int t = 0; t < 2; ++t)
                DH_UNROLL for (int r = 0; r < 16; r += 2) hmax = fmaxf(hmax, fmaxf(acc[m][t][r], acc[m][t][r + 1]));
            acc_map(acc, [&](int, int t, int, float h) { float s, em; softplus_deriv_from_h(h, s, em); return (t ? w1 : w0) * s; });
            if (save) acc_store_native_b(acc, tile_rsrc(asave + ((int64_t)7 * ntiles + tile) * TILE_F), loff);
            ts = lds_handoff(acc, smain, hs, &hs.lmax[7], tid, wave, lane);
        }
        for (int l = 7; l >= 1; --l) {
            DH_STAMP(it, l, 0);
            acc_zero(acc);
            // sigma' comes from act[l-1] == the input of layer l: its first m-slab is requested inside the GEMM's last chunks
            const rsrc_t hr = tile_rsrc(act + ((int64_t)(l - 1) * ntiles + tile) * TILE_F);
            Slab h0, h1;
            gemm_rows_hp<false>(acc, smain, 16, P.rev[l], wave, lane, [&] { slab_ld(h0, hr, loff, 0); });      // u_l = a_l W_l
            DH_STAMP(it, l, 1);
            slab_ld(h1, hr, loff, 1);
            const float inv = ts.inv * winv_from_bits(P.wabs[l]);
            if (l == 4) {                                                         // skip path -> ge (true units)
                gemm_auxout_hp(ge, smain, 16, P.revaux[4], wave, lane);
                DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) DH_UNROLL for (int r = 0; r < 16; ++r) ge[tt][r] *= inv;
            }
            // a_{l-1} = u_l * sigma'(z_{l-1})
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                const Slab& hs_ = m ? h1 : h0;
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = hs_.v[t * 4 + r4];
                        hmax = fmaxf(hmax, fmaxf(h[0], h[1]));
                        hmax = fmaxf(hmax, fmaxf(h[2], h[3]));
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            acc[m][t][4 * r4 + rr] *= s * inv;
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (save) acc_store_native_b(acc, tile_rsrc(asave + ((int64_t)(l - 1) * ntiles + tile) * TILE_F), loff);
            DH_STAMP(it, l, 2);
            ts = lds_handoff(acc, smain, hs, &hs.lmax[l - 1], tid, wave, lane, 0.f, nullptr, nullptr, it, l);
        }
        DH_STAMP(it, 0, 0);
        {                                                                         // ge += a_0 W_0
            f32x16 g0[AUX_NTW];
            aux_zero(g0);
            gemm_auxout_hp(g0, smain, 16, P.revaux[0], wave, lane);
            const float inv = ts.inv * winv_from_bits(P.wabs[0]);
            DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) DH_UNROLL for (int r = 0; r < 16; ++r) ge[tt][r] = fmaf(g0[tt][r], inv, ge[tt][r]);
        }
        // ge -> LDS aux image
        DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
            const int col = aux_col(wave, tt, lane);
            if (col < AUXW) {
                DH_UNROLL for (int r = 0; r < 16; ++r) saux[aux_row(wave, r, lane) * LDA + col] = ge[tt][r];
            }
        }
        __syncthreads();
        if (tid < TM) {
            const int64_t gp = tile * TM + tid;
            if (gp < npts) {
                const float* g = saux + tid * LDA;
                float n[3];
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float x = pts[gp * 3 + c];
                    float v = g[c];
                    DH_UNROLL for (int k = 0; k < 6; ++k) {
                        const float f = (float)(1 << k);
                        float s, co; sincosf(x * f, &s, &co);
                        v += f * (co * g[3 + 6 * k + c] - s * g[3 + 6 * k + 3 + c]);
                    }
                    n[c] = v;
                }
                normals[gp * 3 + 0] = n[0]; normals[gp * 3 + 1] = n[1]; normals[gp * 3 + 2] = n[2];
                if (save == 2) { for (int c = 0; c < 40; ++c) gesave[gp * 40 + c] = c < EMB ? g[c] : 0.f; }     // pose refinement
            }
        }
        __syncthreads();
    }
    if (save && absmax && tid < 8) post_class_max(absmax, ABSMAX_ASAVE + tid, hs.lmax[tid]);
    if (absmax) {                                      // (also forward-only renders: save == 0)
        hmax = wave_max(hmax);
        if (lane == 0) post_class_max(absmax, ABSMAX_ACT, hmax);
    }
}

// ------------------------------------------------------------------------------------------------ colour backward
template <bool RAYS>
__global__ __launch_bounds__(256, 2) void color_bwd_h_kernel(ColHPtrs C, const float* __restrict__ colors,
                                                            const float* __restrict__ d_colors, int64_t npts,
                                                            const float* __restrict__ cact, float* __restrict__ czbar,
                                                            float* __restrict__ featbar, float* __restrict__ d_normals,
                                                            float* __restrict__ tpart, const float* __restrict__ dirs,
                                                            int n_per_ray, float* __restrict__ d_pts,
                                                            float* __restrict__ d_dirs_pts, unsigned* __restrict__ absmax,
                                                            unsigned* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) _Float16 smain[IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: craw [128][4]
    __shared__ HScratch hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM;
    hs_init(hs, tid);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float* tp = tpart + tile * N_TILE_PART * 256;
        if (tid < TM) {
            const int64_t gp = tile * TM + tid;
            DH_UNROLL for (int j = 0; j < 3; ++j) {
                float v = 0.f;
                if (gp < npts) { const float c = colors[gp * 3 + j]; v = d_colors[gp * 3 + j] * c * (1.f - c); }
                saux[tid * 4 + j] = v;
            }
            saux[tid * 4 + 3] = 0.f;
        }
        __syncthreads();
        if (tid < 3) {                                               // db4
            float s = 0.f;
            for (int r = 0; r < TM; ++r) s += saux[r * 4 + tid];
            tp[TP_COL_B4 * 256 + tid] = s;
        }
        f32x16 acc[MT][2];
        // lin4: dW4 partials, zbar_3 = (craw W4) * [h4 > 0]
        acc_load_native_b(acc, tile_rsrc(cact + ((int64_t)3 * ntiles + tile) * TILE_F), loff);
        {
            const int col0 = acc_col(wave, 0, lane), col1 = acc_col(wave, 1, lane);
            float w4[3][2];
            DH_UNROLL for (int j = 0; j < 3; ++j) { w4[j][0] = C.w4[j * 256 + col0]; w4[j][1] = C.w4[j * 256 + col1]; }
            float dw[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            DH_UNROLL for (int m = 0; m < MT; ++m)
                DH_UNROLL for (int r = 0; r < 16; ++r) {
                    const f32x4 cr = *reinterpret_cast<const f32x4*>(saux + acc_row(m, r, lane) * 4);
                    DH_UNROLL for (int t = 0; t < 2; ++t) {
                        const float h = acc[m][t][r];
                        DH_UNROLL for (int j = 0; j < 3; ++j) dw[j][t] = fmaf(cr[j], h, dw[j][t]);
                        const float hb = cr[0] * w4[0][t] + cr[1] * w4[1][t] + cr[2] * w4[2][t];
                        acc[m][t][r] = h > 0.f ? hb : 0.f;
                    }
                }
            DH_UNROLL for (int j = 0; j < 3; ++j)
                DH_UNROLL for (int t = 0; t < 2; ++t) {
                    float s = dw[j][t];
                    s += __shfl_xor(s, 32);
                    if (lane < 32) tp[(TP_COL_W4 + j) * 256 + 64 * wave + 32 * t + lane] = s;
                }
        }
        acc_store_native_b(acc, tile_rsrc(czbar + ((int64_t)3 * ntiles + tile) * TILE_F), loff);
        tile_colsum(acc, tp + (TP_COL_B0 + 3) * 256, wave, lane);
        TileScale ts = lds_handoff(acc, smain, hs, &hs.lmax[3], tid, wave, lane, 0.f, nullptr, tmax + (TMAX_CZBAR + 3) * ntiles + tile);
        for (int l = 3; l >= 1; --l) {
            acc_zero(acc);
            const rsrc_t hr = tile_rsrc(cact + ((int64_t)(l - 1) * ntiles + tile) * TILE_F);
            Slab h0, h1;
            gemm_rows_hp<false>(acc, smain, 16, C.rev[l], wave, lane, [&] { slab_ld(h0, hr, loff, 0); });      // hbar_l = zbar_l W_l
            slab_ld(h1, hr, loff, 1);
            const float inv = ts.inv * winv_from_bits(C.wabs[l]);
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                const Slab& hs_ = m ? h1 : h0;
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = hs_.v[t * 4 + r4];
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr)
                            acc[m][t][4 * r4 + rr] = h[rr] > 0.f ? acc[m][t][4 * r4 + rr] * inv : 0.f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc_store_native_b(acc, tile_rsrc(czbar + ((int64_t)(l - 1) * ntiles + tile) * TILE_F), loff);
            tile_colsum(acc, tp + (TP_COL_B0 + l - 1) * 256, wave, lane);
            ts = lds_handoff(acc, smain, hs, &hs.lmax[l - 1], tid, wave, lane, 0.f, nullptr, tmax + (TMAX_CZBAR + l - 1) * ntiles + tile);
        }
        // lin0: featbar = zbar_0 W0[:,33:] ; extras adjoint = zbar_0 W0[:,:33] (only the normal columns 30..32 matter)
        const float inv0 = ts.inv * winv_from_bits(C.wabs[0]);
        acc_zero(acc);
        gemm_rows_hp(acc, smain, 16, C.rev[0], wave, lane);
        acc_map(acc, [&](int, int, int, float v) { return v * inv0; });
        acc_store_native_b(acc, tile_rsrc(featbar + tile * TILE_F), loff);
        tile_max_publish(hs.sred2, wave, lane, acc_absmax(acc));       // featbar's maxima: read behind the barrier that ends the tile
        f32x16 a2[AUX_NTW];
        aux_zero(a2);
        gemm_auxout_hp(a2, smain, 16, C.revaux, wave, lane);
        DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) DH_UNROLL for (int r = 0; r < 16; ++r) a2[tt][r] *= inv0;
        DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
            const int col = aux_col(wave, tt, lane);
            if (col >= 30 && col < 33) {
                DH_UNROLL for (int r = 0; r < 16; ++r) {
                    const int64_t gp = tile * TM + aux_row(wave, r, lane);
                    if (gp < npts) d_normals[gp * 3 + (col - 30)] += a2[tt][r];
                }
            }
        }
        if (RAYS) {
            // a2 columns 0..32 -> LDS (the craw scratch in saux is dead by now), then one thread per point
            __syncthreads();
            DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
                const int col = aux_col(wave, tt, lane);
                if (col < CAUX) {
                    DH_UNROLL for (int r = 0; r < 16; ++r) saux[aux_row(wave, r, lane) * LDA + col] = a2[tt][r];
                }
            }
            __syncthreads();
            if (tid < TM) {
                const int64_t gp = tile * TM + tid;
                if (gp < npts) {
                    const float* row = saux + tid * LDA;
                    const int64_t ray = gp / n_per_ray;
                    DH_UNROLL for (int c = 0; c < 3; ++c) {
                        d_pts[gp * 3 + c] = row[c];
                        const float dv = dirs[ray * 3 + c];
                        float v = row[3 + c];
                        DH_UNROLL for (int kf = 0; kf < 4; ++kf) {
                            const float f = (float)(1 << kf);
                            float sn, co; sincosf(dv * f, &sn, &co);
                            v += f * (co * row[6 + 6 * kf + c] - sn * row[6 + 6 * kf + 3 + c]);
                        }
                        d_dirs_pts[gp * 3 + c] = v;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {                                  // (sred2: the next write to it is a whole tile away)
            const float m = tile_max_read(hs.sred2);
            hs.lmax[4] = fmaxf(hs.lmax[4], m);
            tmax[TMAX_FEATBAR * ntiles + tile] = __builtin_bit_cast(unsigned, m);
        }
    }
    if (absmax) {
        if (tid < 4) post_class_max(absmax, ABSMAX_CZBAR + tid, hs.lmax[tid]);
        if (tid == 4) post_class_max(absmax, ABSMAX_FEATBAR, hs.lmax[4]);
    }
}

// ------------------------------------------------------------------------------------------------ tangent chain
__global__ __launch_bounds__(256, 2) void sdf_tangent_h_kernel(SdfHPtrs P, const float* __restrict__ pts,
                                                              const float* __restrict__ d_normals, int64_t npts,
                                                              const float* __restrict__ act, const float* __restrict__ asave,
                                                              float* __restrict__ t0aux, float* __restrict__ tsave,
                                                              float* __restrict__ rsave, float* __restrict__ tpart,
                                                              unsigned* __restrict__ absmax, unsigned* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) _Float16 smain[IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    __shared__ HScratch hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM;
    hs_init(hs, tid);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float* tp = tpart + tile * N_TILE_PART * 256;
        {   // tt_0 = J_e(x) nbar
            const int p = tid & (TM - 1), part = tid / TM;
            const int64_t gp = tile * TM + p;
            float x[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f};
            if (gp < npts) {
                DH_UNROLL for (int c = 0; c < 3; ++c) { x[c] = pts[gp * 3 + c]; nb[c] = d_normals[gp * 3 + c]; }
            }
            float* row = saux + p * LDA;
            float mx = 0.f;
            if (part == 0) { row[0] = nb[0]; row[1] = nb[1]; row[2] = nb[2]; mx = fmaxf(fmaxf(fabsf(nb[0]), fabsf(nb[1])), fabsf(nb[2])); }
            if (part == 1) { for (int c = 39; c < LDA; ++c) row[c] = 0.f; }
            for (int k = part; k < 6; k += TPP) {
                const float f = (float)(1 << k);
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    float s, co;
                    sincosf(x[c] * f, &s, &co);
                    const float v0 = f * co * nb[c], v1 = -f * s * nb[c];
                    row[3 + 6 * k + c] = v0;
                    row[3 + 6 * k + 3 + c] = v1;
                    mx = fmaxf(mx, fmaxf(fabsf(v0), fabsf(v1)));
                }
            }
            tile_max_publish(hs.sred2, wave, lane, wave_max(mx));
        }
        __syncthreads();
        const float m_aux = tile_max_read(hs.sred2);                 // max |t_0| of the tile
        if (tid == 0) { hs.lmax[7] = fmaxf(hs.lmax[7], m_aux); tmax[TMAX_T0AUX * ntiles + tile] = __builtin_bit_cast(unsigned, m_aux); }
        const TileScale ts_aux = scale_for_max(m_aux);
        aux_lds_to_native(saux, t0aux + tile * AUXT_F, wave, lane);
        f32x16 acc[MT][2];
        TileScale ts = ts_aux;
        for (int l = 0; l < 8; ++l) {
            acc_zero(acc);
            const rsrc_t hr = tile_rsrc(act + ((int64_t)l * ntiles + tile) * TILE_F);
            const rsrc_t ar = tile_rsrc(asave + ((int64_t)l * ntiles + tile) * TILE_F);
            const rsrc_t rr_ = tile_rsrc(rsave + ((int64_t)l * ntiles + tile) * TILE_F);
            // l == 4: the main image (t_4) was written at the scale of max(|t_4|, |t_0|) so that both GEMMs share one accumulator
            if (l > 0) gemm_rows_hp<H2_LEAN2>(acc, smain, l == 4 ? 14 : 16, P.main[l], wave, lane);
            if (l == 0 || l == 4) gemm_rows_aux_h(acc, saux, P.aux[l], wave, lane, ts.S);     // abar_l
            const float inv = ts.inv * winv_from_bits(P.wabs[l]);
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                // (two input streams: one m-slab of each in flight is what the register file holds beside the accumulators)
                Slab hs_, as_;
                slab_ld(hs_, hr, loff, m); slab_ld(as_, ar, loff, m);
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = hs_.v[t * 4 + r4], a = as_.v[t * 4 + r4];
                        f32x4 rv;
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            const float ab = acc[m][t][4 * r4 + rr] * inv;
                            rv[rr] = ab * a[rr] * (SOFTPLUS_BETA * em);
                            acc[m][t][4 * r4 + rr] = s * ab;
                        }
                        tile_st(rr_, loff, (m * 2 + t) * 4 + r4, rv);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l < 7) {
                acc_store_native_b(acc, tile_rsrc(tsave + ((int64_t)l * ntiles + tile) * TILE_F), loff);    // t_{l+1}
                ts = lds_handoff(acc, smain, hs, &hs.lmax[l], tid, wave, lane, l == 3 ? m_aux : 0.f, nullptr, tmax + (TMAX_TSAVE + l) * ntiles + tile);
            } else {
                tile_colsum(acc, tp + TP_W8ROW0_T * 256, wave, lane);                                 // colsum t_8
            }
        }
        __syncthreads();
    }
    if (absmax) {
        if (tid < 7) post_class_max(absmax, ABSMAX_TSAVE + tid, hs.lmax[tid]);
        if (tid == 7) post_class_max(absmax, ABSMAX_T0AUX, hs.lmax[7]);
    }
}

// ------------------------------------------------------------------------------------------------ SDF backward chain
template <bool RAYS>
__global__ __launch_bounds__(256, 2) void sdf_bwd_h_kernel(SdfHPtrs P, const float* __restrict__ d_sdf, int64_t npts,
                                                          const float* __restrict__ act, const float* __restrict__ rsave,
                                                          const float* __restrict__ featbar, float* __restrict__ zbar,
                                                          float* __restrict__ tpart, const float* __restrict__ pts,
                                                          const float* __restrict__ d_normals, const float* __restrict__ gesave,
                                                          float* __restrict__ d_pts, unsigned* __restrict__ absmax,
                                                          unsigned* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) _Float16 smain[IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: sdfbar [128]
    __shared__ HScratch hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM;
    hs_init(hs, tid);
    const float w0c0 = P.w8row0[acc_col(wave, 0, lane)], w0c1 = P.w8row0[acc_col(wave, 1, lane)];
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        float* tp = tpart + tile * N_TILE_PART * 256;
        if (tid < TM) {
            const int64_t gp = tile * TM + tid;
            saux[tid] = gp < npts ? d_sdf[gp] : 0.f;
        }
        f32x16 acc[MT][2];
        f32x16 eb[AUX_NTW];
        if (RAYS) aux_zero(eb);
        acc_load_native_b(acc, tile_rsrc(featbar + tile * TILE_F), loff);
        tile_colsum(acc, tp + TP_SDF_B8 * 256, wave, lane);
        TileScale ts = lds_handoff(acc, smain, hs, nullptr, tid, wave, lane);      // (featbar's class maximum: the colour backward)
        if (wave == 0) {                                       // sum of sdfbar -> bbar_8[0]
            float s = 0.f;
            for (int i = lane; i < TM; i += 64) s += saux[i];
            DH_UNROLL for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) tp[TP_SCAL * 256] = s;
        }
        // hbar_8 = featbar W8[1:,:] + sdfbar (x) W8[0,:]
        acc_zero(acc);
        gemm_rows_hp<H2_LEAN2>(acc, smain, 16, P.rev[8], wave, lane);
        {
            const float inv = ts.inv * winv_from_bits(P.wabs[8]);
            DH_UNROLL for (int m = 0; m < MT; ++m)
                DH_UNROLL for (int r = 0; r < 16; ++r) {
                    const float sb = saux[acc_row(m, r, lane)];
                    acc[m][0][r] = fmaf(sb, w0c0, acc[m][0][r] * inv);
                    acc[m][1][r] = fmaf(sb, w0c1, acc[m][1][r] * inv);
                }
        }
        for (int l = 7; l >= 0; --l) {
            float ws0 = 0.f, ws1 = 0.f;                         // sum_rows sdfbar * h_8 (l == 7 only)
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                Slab hs_, rs_;
                slab_ld(hs_, tile_rsrc(act + ((int64_t)l * ntiles + tile) * TILE_F), loff, m);
                slab_ld(rs_, tile_rsrc(rsave + ((int64_t)l * ntiles + tile) * TILE_F), loff, m);
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = hs_.v[t * 4 + r4], rv = rs_.v[t * 4 + r4];
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            if (l == 7) {
                                const float sb = saux[acc_row(m, 4 * r4 + rr, lane)];
                                if (t == 0) ws0 = fmaf(sb, h[rr], ws0); else ws1 = fmaf(sb, h[rr], ws1);
                            }
                            acc[m][t][4 * r4 + rr] = fmaf(acc[m][t][4 * r4 + rr], s, rv[rr]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l == 7) {
                ws0 += __shfl_xor(ws0, 32); ws1 += __shfl_xor(ws1, 32);
                if (lane < 32) {
                    tp[TP_W8ROW0_S * 256 + 64 * wave + lane] = ws0;
                    tp[TP_W8ROW0_S * 256 + 64 * wave + 32 + lane] = ws1;
                }
            }
            acc_store_native_b(acc, tile_rsrc(zbar + ((int64_t)l * ntiles + tile) * TILE_F), loff);
            tile_colsum(acc, tp + (TP_SDF_B0 + l) * 256, wave, lane);
            if (l > 0) {
                ts = lds_handoff(acc, smain, hs, &hs.lmax[l], tid, wave, lane, 0.f, nullptr, tmax + (TMAX_ZBAR + l) * ntiles + tile);
                const float inv = ts.inv * winv_from_bits(P.wabs[l]);
                if (RAYS && l == 4) {                                                // skip path -> ebar (true units)
                    gemm_auxout_hp(eb, smain, 16, P.revaux[4], wave, lane);
                    DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) DH_UNROLL for (int r = 0; r < 16; ++r) eb[tt][r] *= inv;
                }
                acc_zero(acc);
                gemm_rows_hp<H2_LEAN2>(acc, smain, 16, P.rev[l], wave, lane);             // hbar_l = zbar_l W_l
                acc_map(acc, [&](int, int, int, float v) { return v * inv; });
            } else if (RAYS) {
                ts = lds_handoff(acc, smain, hs, &hs.lmax[0], tid, wave, lane, 0.f, nullptr, tmax + TMAX_ZBAR * ntiles + tile);      // zbar_0
                {
                    f32x16 e0[AUX_NTW];
                    aux_zero(e0);
                    gemm_auxout_hp(e0, smain, 16, P.revaux[0], wave, lane);           // ebar += zbar_0 W_0
                    const float inv = ts.inv * winv_from_bits(P.wabs[0]);
                    DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) DH_UNROLL for (int r = 0; r < 16; ++r) eb[tt][r] = fmaf(e0[tt][r], inv, eb[tt][r]);
                }
                DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
                    const int col = aux_col(wave, tt, lane);
                    if (col < AUXW) {
                        DH_UNROLL for (int r = 0; r < 16; ++r) saux[aux_row(wave, r, lane) * LDA + col] = eb[tt][r];
                    }
                }
                __syncthreads();
                if (tid < TM) {
                    const int64_t gp = tile * TM + tid;
                    if (gp < npts) {
                        const float* e = saux + tid * LDA;
                        const float* ge = gesave + gp * 40;
                        DH_UNROLL for (int c = 0; c < 3; ++c) {
                            const float x = pts[gp * 3 + c], nb = d_normals[gp * 3 + c];
                            float v = e[c], dn = 0.f;
                            DH_UNROLL for (int k = 0; k < 6; ++k) {
                                const float f = (float)(1 << k);
                                float sn, co; sincosf(x * f, &sn, &co);
                                v += f * (co * e[3 + 6 * k + c] - sn * e[3 + 6 * k + 3 + c]);
                                dn -= f * f * (sn * ge[3 + 6 * k + c] + co * ge[3 + 6 * k + 3 + c]);
                            }
                            d_pts[gp * 3 + c] += v + nb * dn;
                        }
                    }
                }
            } else {
                // zbar_0 feeds only the weight-gradient kernel: its maxima, read behind the barrier that ends the tile
                tile_max_publish(hs.sred2, wave, lane, acc_absmax(acc));
            }
        }
        __syncthreads();
        if (!RAYS && tid == 0) {                         // (sred2: the next write to it is a whole tile away)
            const float m = tile_max_read(hs.sred2);
            hs.lmax[0] = fmaxf(hs.lmax[0], m);
            tmax[TMAX_ZBAR * ntiles + tile] = __builtin_bit_cast(unsigned, m);
        }
    }
    if (absmax && tid < 8) post_class_max(absmax, ABSMAX_ZBAR + tid, hs.lmax[tid]);
}

// ------------------------------------------------------------------------------------------------ launchers
static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }
static inline int grid_for(int64_t npts, int grid) {
    const int64_t ntiles = (npts + TM - 1) / TM;
    return (int)(ntiles < grid ? ntiles : grid);
}
int launch_sdf_grad_h(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals,
                      int save, float* gesave, unsigned* absmax, int grid, hipStream_t stream) {
    hipLaunchKernelGGL(sdf_grad_h_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_sdfh_ptrs(packed), pts, npts, act,
                       asave, normals, save, gesave, absmax);
    return ok();
}
int launch_color_fwd_h(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                       const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, unsigned* absmax,
                       int grid, hipStream_t stream) {
    hipLaunchKernelGGL(color_fwd_h_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_colh_ptrs(packed), pts, dirs,
                       n_per_ray, normals, feat, npts, color, cact, caux, save, absmax);
    return ok();
}
int launch_color_bwd_h(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                       int64_t npts, const float* cact, float* czbar, float* featbar, float* d_normals, float* tpart,
                       float* d_pts, float* d_dirs_pts, unsigned* absmax, unsigned* tmax, int grid, hipStream_t st) {
    if (d_pts) hipLaunchKernelGGL(color_bwd_h_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_colh_ptrs(packed), colors,
                                  d_colors, npts, cact, czbar, featbar, d_normals, tpart, dirs, n_per_ray, d_pts, d_dirs_pts, absmax, tmax);
    else hipLaunchKernelGGL(color_bwd_h_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_colh_ptrs(packed), colors,
                            d_colors, npts, cact, czbar, featbar, d_normals, tpart, nullptr, 1, nullptr, nullptr, absmax, tmax);
    return ok();
}
int launch_sdf_tangent_h(const float* packed, const float* pts, const float* d_normals, int64_t npts, const float* act,
                         const float* asave, float* t0aux, float* tsave, float* rsave, float* tpart, unsigned* absmax, unsigned* tmax,
                         int grid, hipStream_t st) {
    hipLaunchKernelGGL(sdf_tangent_h_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdfh_ptrs(packed), pts, d_normals,
                       npts, act, asave, t0aux, tsave, rsave, tpart, absmax, tmax);
    return ok();
}
int launch_sdf_bwd_h(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                     const float* act, const float* rsave, const float* featbar, const float* gesave, float* zbar, float* tpart,
                     float* d_pts, unsigned* absmax, unsigned* tmax, int grid, hipStream_t st) {
    if (d_pts) hipLaunchKernelGGL(sdf_bwd_h_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdfh_ptrs(packed), d_sdf, npts,
                                  act, rsave, featbar, zbar, tpart, pts, d_normals, gesave, d_pts, absmax, tmax);
    else hipLaunchKernelGGL(sdf_bwd_h_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdfh_ptrs(packed), d_sdf, npts,
                            act, rsave, featbar, zbar, tpart, nullptr, nullptr, nullptr, nullptr, absmax, tmax);
    return ok();
}

}  // namespace dh
