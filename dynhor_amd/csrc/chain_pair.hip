// The tile-PAIR forms of the two-piece fp16 chains (pair16h.h; round 6): one workgroup per CU owns two 64-point tiles, holds a layer's
// weight slice in registers across both tiles' GEMMs, and deals one tile's epilogue + image hand-off under the other tile's MFMAs.
// Each kernel here computes what its kernels_mlp_h.hip twin computes -- same saved tiles, same outputs, same scale tables, bit for
// bit (the maths live in kernels_mlp.hip / kernels_mlp_bwd.hip's headers) -- and is chosen by the launcher for launches of at least
// PAIR_MIN_TILES tiles (fewer would leave CUs idle: a pair occupies a CU that two tile workgroups would share).
#include "tile.h"
#include "kernels.h"
#include "mlp_common.h"
#include "tile16h.h"
#include "pair16h.h"
#include "workspace.h"

namespace dh {

// kernels_mlp_h.hip
struct HScratchP { float sred[4]; float sred2[4]; float lmax[12]; };
struct ColHPtrs;
struct ColPPtrs {                  // (plain pointers: a run-time-indexed pointer ARRAY in a kernel argument goes to scratch -- NEXT_STEPS trap 1)
    const u32x4* main0;            // colour lin0's forward weights; lin l at main0 + l * COL_MAIN_STRIDE
    const u32x4* aux;
    const float* bias0;            // lin l at bias0 + l * COL_BIAS_STRIDE
    const float* w4;
    const float* b4;
    const unsigned* wabs;          // [0..3]
};
constexpr int64_t COL_MAIN_STRIDE = (PACKH.col_fwd_main[1] - PACKH.col_fwd_main[0]) / 4;      // in u32x4
constexpr int64_t COL_BIAS_STRIDE = PACK.col_bias[1] - PACK.col_bias[0];
static_assert(PACKH.col_fwd_main[2] - PACKH.col_fwd_main[1] == PACKH.col_fwd_main[1] - PACKH.col_fwd_main[0] &&
              PACKH.col_fwd_main[3] - PACKH.col_fwd_main[2] == PACKH.col_fwd_main[1] - PACKH.col_fwd_main[0] &&
              (PACKH.col_fwd_main[1] - PACKH.col_fwd_main[0]) % 4 == 0, "colour forward weights: one stride");
static_assert(PACK.col_bias[2] - PACK.col_bias[1] == COL_BIAS_STRIDE && PACK.col_bias[3] - PACK.col_bias[2] == COL_BIAS_STRIDE, "colour biases: one stride");
static inline ColPPtrs make_colp_ptrs(const float* packed) {
    ColPPtrs C;
    C.main0 = reinterpret_cast<const u32x4*>(packed + PACKH.col_fwd_main[0]);
    C.aux = reinterpret_cast<const u32x4*>(packed + PACKH.col_fwd_aux0);
    C.bias0 = packed + PACK.col_bias[0];
    C.w4 = packed + PACK.col_w4;
    C.b4 = packed + PACK.col_b4;
    C.wabs = reinterpret_cast<const unsigned*>(packed + PACKH.wabs) + N_SDF;
    return C;
}

// ------------------------------------------------------------------------------------------------ colour forward
// epilogue of one colour layer for one tile (bias, ReLU, the saved tile, the running maximum) + the hand-off into the tile's image
template <bool SAVE>
struct ColFwdEpi : PEpi<ColFwdEpi<SAVE>, 3, 4> {
    f32x16 (&acc)[MT][2];
    HScratchP& hs;
    float* lmax;                   // the workgroup's running class maximum (an LDS word), or nullptr
    const int wave, lane, tid;
    const float inv, b0, b1;
    const rsrc_t st;               // where the layer's activation tile goes (a 0-record descriptor where this tile stores nothing)
    const int loff;
    __device__ __forceinline__ ColFwdEpi(f32x16 (&acc_)[MT][2], _Float16* img_, HScratchP& hs_, float* lmax_, int wave_, int lane_, int tid_,
                                         float inv_, float b0_, float b1_, rsrc_t st_, int loff_, bool)
        : acc(acc_), hs(hs_), lmax(lmax_), wave(wave_), lane(lane_), tid(tid_), inv(inv_), b0(b0_), b1(b1_), st(st_), loff(loff_) {
        this->handoff_init(img_, wave_, lane_);
    }
    template <int G, int SUB>
    __device__ __forceinline__ void elem() {
        constexpr int m = G / 8, t = (G / 4) % 2, r4 = G % 4;
        const float b = t ? b1 : b0;
        if constexpr (SUB == 0 || SUB == 1) {
            constexpr int r = 4 * r4 + 2 * SUB;
            acc[m][t][r] = fmaxf(fmaf(acc[m][t][r], inv, b), 0.f);
            acc[m][t][r + 1] = fmaxf(fmaf(acc[m][t][r + 1], inv, b), 0.f);
        } else {
            if constexpr (SAVE) {
                f32x4 v;
                v[0] = acc[m][t][4 * r4 + 0]; v[1] = acc[m][t][4 * r4 + 1]; v[2] = acc[m][t][4 * r4 + 2]; v[3] = acc[m][t][4 * r4 + 3];
                tile_st(st, loff, G, v);
            }
            this->m0 = fmaxf(this->m0, fmaxf(fabsf(acc[m][t][4 * r4 + 0]), fabsf(acc[m][t][4 * r4 + 1])));
            this->m1 = fmaxf(this->m1, fmaxf(fabsf(acc[m][t][4 * r4 + 2]), fabsf(acc[m][t][4 * r4 + 3])));
        }
    }
    __device__ __forceinline__ void publish(float wm) { tile_max_publish(hs.sred, wave, lane, wm); }
    __device__ __forceinline__ float read_tile_max() { return tile_max_read(hs.sred); }
    __device__ __forceinline__ void on_tile_max(float m) { if (tid == 0 && lmax) *lmax = fmaxf(*lmax, m); }
};

// the colour head: three per-point dots of the image rows (hi + lo) with the rows of lin4 staged in LDS (sw4 [3][256]) -- tile16h.h
// row_dot256_hp's products in its order, per output; the row segment is read once for the three
__device__ __forceinline__ void row_dot256x3_hp(const _Float16* img, const float* sw4, int tid, float (&out)[3]) {
    constexpr int TPP_ = 256 / TM, SEG = 256 / TPP_;
    const int p = tid / TPP_, part = tid % TPP_;
    const f16x8* xh = reinterpret_cast<const f16x8*>(img + p * LDH + part * SEG);
    const f16x8* xl = reinterpret_cast<const f16x8*>(img + PLANE_H + p * LDH + part * SEG);
    float s[3] = {0.f, 0.f, 0.f};
    DH_UNROLL for (int i = 0; i < SEG / 8; ++i) {
        const f16x8 h = xh[i], l = xl[i];
        float x[8];
        DH_UNROLL for (int k = 0; k < 8; ++k) x[k] = (float)h[k] + (float)l[k];
        DH_UNROLL for (int j = 0; j < 3; ++j) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(sw4 + j * 256 + part * SEG);
            const f32x4 b0 = wr[2 * i], b1 = wr[2 * i + 1];
            DH_UNROLL for (int k = 0; k < 4; ++k) s[j] = fmaf(x[k], b0[k], s[j]);
            DH_UNROLL for (int k = 0; k < 4; ++k) s[j] = fmaf(x[4 + k], b1[k], s[j]);
        }
    }
    DH_UNROLL for (int j = 0; j < 3; ++j) {
        DH_UNROLL for (int off = 1; off < TPP_; off <<= 1) s[j] += __shfl_xor(s[j], off);
        out[j] = s[j];
    }
}

// the exposed hand-off of a pair's first image (the feature tile): kernels_mlp_h.hip lds_handoff, on this file's scratch
__device__ __forceinline__ TileScale pair_handoff_exposed(const f32x16 (&acc)[MT][2], _Float16* img, HScratchP& hs, float* lmax, int tid,
                                                          int wave, int lane, const float* extra_lds) {
    tile_max_publish(hs.sred, wave, lane, acc_absmax(acc));
    __syncthreads();
    const float m = tile_max_read(hs.sred);
    if (tid == 0 && lmax) *lmax = fmaxf(*lmax, m);
    const TileScale ts = scale_for_max(fmaxf(m, *extra_lds));
    acc_to_lds_split(acc, img, wave, lane, ts.S);
    __syncthreads();
    return ts;
}

template <bool SAVE>
__global__ __launch_bounds__(256, 1) void color_fwd_p_kernel(ColPPtrs C, const float* __restrict__ pts, const float* __restrict__ dirs,
                                                            int n_per_ray, const float* __restrict__ normals,
                                                            const float* __restrict__ feat, int64_t npts, float* __restrict__ color,
                                                            float* __restrict__ cact, float* __restrict__ caux,
                                                            unsigned* __restrict__ absmax) {
    __shared__ __attribute__((aligned(16))) _Float16 simg[2][IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[2][TM * LDA];
    __shared__ HScratchP hs;
    __shared__ float sfb[64];                                     // the view embedding of a one-ray tile (two tiles x [24])
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    if (tid < 12) hs.lmax[tid] = 0.f;
    float winv0 = winv_from_bits(C.wabs[0]), winv1 = winv_from_bits(C.wabs[1]), winv2 = winv_from_bits(C.wabs[2]), winv3 = winv_from_bits(C.wabs[3]);
    // this lane's two bias columns of the four layers
    float bs[4][2];
    DH_UNROLL for (int l = 0; l < 4; ++l)
        DH_UNROLL for (int t = 0; t < 2; ++t) bs[l][t] = C.bias0[l * COL_BIAS_STRIDE + acc_col(wave, t, lane)];
    PW<P_NKC> W;
    _Float16* imgA = simg[0];
    _Float16* imgB = simg[1];
    int pit = 0;
    for (int64_t pr = blockIdx.x; pr < npairs; pr += gridDim.x, ++pit) {
        const bool son = pit == 2;                                // (-DDH_STAMPS: the workgroup's third pair is stamped)
        (void)son;
        PSTAMP(son, 0);
        const int64_t tileA = 2 * pr;
        const bool okB = tileA + 1 < ntiles;                      // an odd last tile: B replays A with every output switched off
        const int64_t tileB = okB ? tileA + 1 : tileA;
        // the feature tiles first: their latency runs under the extras
        f32x16 accA[MT][2], accB[MT][2];
        acc_load_native_b(accA, tile_rsrc(feat + tileA * TILE_F), loff);
        acc_load_native_b(accB, tile_rsrc(feat + tileB * TILE_F), loff);
        {   // the extras [p, embed(view), n]: waves 0, 1 -> tile A, waves 2, 3 -> tile B.  The view embedding depends on the RAY only: where
            // a tile's 64 points belong to one ray (training: 128 samples per ray) twelve lanes evaluate its twelve sincosf once and every
            // row copies them -- the same arguments, the same bits; a tile that spans rays evaluates per point (a tile's two waves split
            // the four frequencies).  (One wave per SIMD hides nothing: 12 sincosf per point cost a pair ~10,000 exposed cycles.)
            const int which = tid >> 7, half = (tid >> 6) & 1, p = tid & 63;
            const int64_t t0 = (which ? tileB : tileA) * TM;
            const int64_t gp = t0 + p;
            auto single_ray = [&](int64_t first) {
                const int64_t last = first + TM - 1 < npts ? first + TM - 1 : npts - 1;
                return first / n_per_ray == last / n_per_ray;
            };
            const bool oneA = single_ray(tileA * TM), oneB = single_ray(tileB * TM);      // (workgroup-uniform)
            const bool one_ray = which ? oneB : oneA;
            float* row = saux[which] + p * LDA;
            float* emb = sfb + which * 32;                                    // [24] sin / cos of the tile's one ray
            if (one_ray && half == 0 && lane < 12) {
                const int c = lane % 3, k = lane / 3;
                float sn, co; sincosf(dirs[(t0 / n_per_ray) * 3 + c] * (float)(1 << k), &sn, &co);
                emb[6 * k + c] = sn;
                emb[6 * k + 3 + c] = co;
            }
            float mx = 1.f;                                      // sin / cos of the view embedding
            if (gp < npts) {
                const int64_t ray = gp / n_per_ray;
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = dirs[ray * 3 + c];
                    if (!one_ray) {
                        DH_UNROLL for (int kk = 0; kk < 2; ++kk) {
                            const int k = 2 * half + kk;
                            float sn, co; sincosf(d * (float)(1 << k), &sn, &co);
                            row[6 + 6 * k + c] = sn;
                            row[6 + 6 * k + 3 + c] = co;
                        }
                    }
                    if (half == 0) {
                        const float x = pts[gp * 3 + c], nn = normals[gp * 3 + c];
                        row[c] = x;
                        row[3 + c] = d;
                        row[30 + c] = nn;
                        mx = fmaxf(mx, fmaxf(fabsf(x), fmaxf(fabsf(d), fabsf(nn))));
                    }
                }
            } else {
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    DH_UNROLL for (int kk = 0; kk < 2; ++kk) { row[6 + 6 * (2 * half + kk) + c] = 0.f; row[6 + 6 * (2 * half + kk) + 3 + c] = 0.f; }
                    if (half == 0) { row[c] = 0.f; row[3 + c] = 0.f; row[30 + c] = 0.f; }
                }
            }
            if (half == 0) {
                DH_UNROLL for (int c = CAUX; c < LDA; ++c) row[c] = 0.f;
                mx = wave_max(mx);
                if (lane == 0) hs.sred2[which] = mx;
            }
            if (oneA || oneB) {                                  // (uniform) the tile's other wave must see emb: copy behind a barrier
                pair_barrier();
                if (one_ray && gp < npts) { DH_UNROLL for (int j = 0; j < 12; ++j) row[6 + 12 * half + j] = emb[12 * half + j]; }
            }
        }
        // (the barriers inside also publish saux; the previous pair ended with one, so the images are free).  The extras share layer
        // 0's accumulator with feat: one scale for both, from the larger of the two maxima
        PSTAMP(son, 88);
        TileScale tsA = pair_handoff_exposed(accA, imgA, hs, SAVE ? &hs.lmax[4] : nullptr, tid, wave, lane, &hs.sred2[0]);
        PSTAMP(son, 89);
        TileScale tsB = pair_handoff_exposed(accB, imgB, hs, SAVE ? &hs.lmax[4] : nullptr, tid, wave, lane, &hs.sred2[1]);
        PSTAMP(son, 90);
        if (SAVE && tid == 0) hs.lmax[5] = fmaxf(hs.lmax[5], fmaxf(hs.sred2[0], hs.sred2[1]));
        if (SAVE) {
            aux_lds_to_native(saux[0], caux + tileA * AUXT_F, wave, lane);
            if (okB) aux_lds_to_native(saux[1], caux + tileB * AUXT_F, wave, lane);
        }
        PSTAMP(son, 91);
        // layer 0's weight slice: requested here, not by the previous pair's last phase -- 256 registers held across the output stage and
        // the next prologue (sincosf, two tile loads, two hand-offs) were spilled to scratch and back
        pw_load_all(W, C.main0, wave, lane);
        PSTAMP(son, 1);
        _Pragma("unroll 1") for (int l = 0; l < 4; ++l) {
            const float wl = l == 0 ? winv0 : l == 1 ? winv1 : l == 2 ? winv2 : winv3;
            const float wp = l == 1 ? winv0 : l == 2 ? winv1 : winv2;                          // layer l - 1 (l >= 1)
            const float c0 = l == 0 ? bs[0][0] : l == 1 ? bs[1][0] : l == 2 ? bs[2][0] : bs[3][0];
            const float c1 = l == 0 ? bs[0][1] : l == 1 ? bs[1][1] : l == 2 ? bs[2][1] : bs[3][1];
            const float p0 = l == 1 ? bs[0][0] : l == 2 ? bs[1][0] : bs[2][0];
            const float p1 = l == 1 ? bs[0][1] : l == 2 ? bs[1][1] : bs[2][1];
            // ---- phase 1: G_A(l) over B's epilogue of layer l - 1
            acc_zero(accA);
            if (l == 0) {
                PNoEpi e;
                pair_phase<false>(accA, imgA, W, C.main0, C.main0, wave, lane, e);
                gemm_rows_aux_h(accA, saux[0], C.aux, wave, lane, tsA.S);
            } else {
                ColFwdEpi<SAVE> e(accB, imgB, hs, SAVE ? &hs.lmax[l - 1] : nullptr, wave, lane, tid, tsB.inv * wp, p0, p1,
                                  tile_rsrc_if(cact + ((int64_t)(l - 1) * ntiles + tileB) * TILE_F, okB), loff, okB);
                if (son && l == 2) e.stamp = 18;
                pair_phase<false>(accA, imgA, W, C.main0 + (int64_t)l * COL_MAIN_STRIDE, C.main0, wave, lane, e);
                tsB.S = e.S; tsB.inv = e.inv_S;
            }
            PSTAMP(son, 2 + 4 * l);
            pair_barrier();
            PSTAMP(son, 3 + 4 * l);
            // ---- phase 2: G_B(l) over A's epilogue of layer l; W takes layer l + 1 (after the last layer: nothing -- a 0-record reload)
            acc_zero(accB);
            {
                ColFwdEpi<SAVE> e(accA, imgA, hs, SAVE ? &hs.lmax[l] : nullptr, wave, lane, tid, tsA.inv * wl, c0, c1,
                                  tile_rsrc(cact + ((int64_t)l * ntiles + tileA) * TILE_F), loff, true);
                if (son && l == 2) e.stamp = 52;
                pair_phase<true>(accB, imgB, W, C.main0 + (int64_t)l * COL_MAIN_STRIDE, C.main0 + (int64_t)(l < 3 ? l + 1 : 3) * COL_MAIN_STRIDE, wave, lane, e, l < 3);
                if (l == 0) gemm_rows_aux_h(accB, saux[1], C.aux, wave, lane, tsB.S);
                tsA.S = e.S; tsA.inv = e.inv_S;
            }
            PSTAMP(son, 4 + 4 * l);
            pair_barrier();
            PSTAMP(son, 5 + 4 * l);
        }
        // the colour head's three rows go to LDS for the output stage (the extras' image of tile A is dead since layer 0): requested
        // here, written behind B's last epilogue
        const float w4a = C.w4[tid], w4b = C.w4[256 + tid], w4c = C.w4[512 + tid];
        {   // B's epilogue of the last layer, with nothing above it
            ColFwdEpi<SAVE> e(accB, imgB, hs, SAVE ? &hs.lmax[3] : nullptr, wave, lane, tid, tsB.inv * winv3, bs[3][0], bs[3][1],
                              tile_rsrc_if(cact + ((int64_t)3 * ntiles + tileB) * TILE_F, okB), loff, okB);
            pair_epi_alone_from<0>(e);
            tsB.S = e.S; tsB.inv = e.inv_S;
        }
        float* sw4 = saux[0];
        sw4[tid] = w4a; sw4[256 + tid] = w4b; sw4[512 + tid] = w4c;
        pair_barrier();
        PSTAMP(son, 86);
        DH_UNROLL for (int which = 0; which < 2; ++which) {
            const int64_t gp = (which ? tileB : tileA) * TM + tid / TPP;
            const float sinv = which ? tsB.inv : tsA.inv;
            float dots[3];
            row_dot256x3_hp(which ? imgB : imgA, sw4, tid, dots);
            DH_UNROLL for (int j = 0; j < 3; ++j) {
                const float raw = fmaf(dots[j], sinv, C.b4[j]);
                if (tid % TPP == 0 && gp < npts && (which == 0 || okB)) color[gp * 3 + j] = 1.f / (1.f + __expf(-raw));
            }
        }
        pair_barrier();
        PSTAMP(son, 87);
    }
    if (SAVE && absmax) {
        if (tid < 4) post_class_max(absmax, ABSMAX_CACT + tid, hs.lmax[tid]);
        if (tid == 4) post_class_max(absmax, ABSMAX_FEAT, hs.lmax[4]);
        if (tid == 5) post_class_max(absmax, ABSMAX_CAUX, hs.lmax[5]);
    }
}

// ------------------------------------------------------------------------------------------------ n = d sdf / d x (reverse chain)
struct SdfGradPPtrs {              // (separate members, selected with ?: -- no run-time-indexed arrays)
    const u32x4 *rev1, *rev2, *rev3, *rev4, *rev5, *rev6, *rev7;
    const u32x4 *revaux0, *revaux4;
    const float* w8row0;
    const unsigned* wabs;          // [0..8]
};
static inline SdfGradPPtrs make_sdfgradp_ptrs(const float* packed) {
    SdfGradPPtrs P;
    auto rv = [&](int l) { return reinterpret_cast<const u32x4*>(packed + PACKH.sdf_rev_main[l]); };
    P.rev1 = rv(1); P.rev2 = rv(2); P.rev3 = rv(3); P.rev4 = rv(4); P.rev5 = rv(5); P.rev6 = rv(6); P.rev7 = rv(7);
    P.revaux0 = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_rev_aux[0]);
    P.revaux4 = reinterpret_cast<const u32x4*>(packed + PACKH.sdf_rev_aux[4]);
    P.w8row0 = packed + PACK.sdf_w8row0;
    P.wabs = reinterpret_cast<const unsigned*>(packed + PACKH.wabs);
    return P;
}
// epilogue behind the GEMM u_l = a_l W_l:  a_{l-1} = u_l * sigma'(z_{l-1}) with sigma' from the saved activation tile act[l-1] (its
// float4s arrive through a register ring, requested RING groups ahead), the saved tile asave[l-1], the running maxima
// (of the tile for the hand-off, of the activations for the range watch), then the hand-off into the tile's image
template <bool SAVE>
struct SdfGradEpi : PEpi<SdfGradEpi<SAVE>, 5, 3> {
    static constexpr int RING = 5;  // float4s of the activation tile in flight (5 groups = 25 MFMA gaps ahead)
    f32x16 (&acc)[MT][2];
    HScratchP& hs;
    float* lmax;
    const int wave, lane, tid;
    const float inv;
    const rsrc_t hr, st;           // act[l-1] (read), asave[l-1] (written; a 0-record descriptor where this tile stores nothing)
    const int loff;
    float& hmax;
    float sa = 0.f, sb = 0.f;
    f32x4 hq[RING];
    __device__ __forceinline__ SdfGradEpi(f32x16 (&acc_)[MT][2], _Float16* img_, HScratchP& hs_, float* lmax_, int wave_, int lane_, int tid_,
                                          float inv_, rsrc_t hr_, rsrc_t st_, int loff_, bool, float& hmax_)
        : acc(acc_), hs(hs_), lmax(lmax_), wave(wave_), lane(lane_), tid(tid_), inv(inv_), hr(hr_), st(st_), loff(loff_), hmax(hmax_) {
        this->handoff_init(img_, wave_, lane_);
        DH_UNROLL for (int g = 0; g < RING; ++g) hq[g] = tile_ld(hr, loff, g);       // groups 0 .. RING-1 (float4 index == group index)
    }
    template <int G, int SUB>
    __device__ __forceinline__ void elem() {
        constexpr int m = G / 8, t = (G / 4) % 2, r4 = G % 4;
        constexpr float C = -SOFTPLUS_BETA * 1.44269504088896f;
        const f32x4& h = hq[G % RING];
        if constexpr (SUB == 0 || SUB == 2) {
            constexpr int i = SUB;                                               // values i, i + 1 of the group
            if constexpr (SUB == 0) { hmax = fmaxf(hmax, fmaxf(h[0], h[1])); hmax = fmaxf(hmax, fmaxf(h[2], h[3])); }
            sa = 1.f - __builtin_amdgcn_exp2f(h[i] * C);                         // softplus_deriv_from_h
            sb = 1.f - __builtin_amdgcn_exp2f(h[i + 1] * C);
        } else if constexpr (SUB == 1 || SUB == 3) {
            constexpr int r = 4 * r4 + (SUB - 1);
            acc[m][t][r] *= sa * inv;
            acc[m][t][r + 1] *= sb * inv;
        } else {
            if constexpr (SAVE) {
                f32x4 v;
                v[0] = acc[m][t][4 * r4 + 0]; v[1] = acc[m][t][4 * r4 + 1]; v[2] = acc[m][t][4 * r4 + 2]; v[3] = acc[m][t][4 * r4 + 3];
                tile_st(st, loff, G, v);
            }
            if constexpr (G + RING < 16) hq[G % RING] = tile_ld(hr, loff, G + RING);
            this->m0 = fmaxf(this->m0, fmaxf(fabsf(acc[m][t][4 * r4 + 0]), fabsf(acc[m][t][4 * r4 + 1])));
            this->m1 = fmaxf(this->m1, fmaxf(fabsf(acc[m][t][4 * r4 + 2]), fabsf(acc[m][t][4 * r4 + 3])));
        }
    }
    __device__ __forceinline__ void publish(float wm) { tile_max_publish(hs.sred, wave, lane, wm); }
    __device__ __forceinline__ float read_tile_max() { return tile_max_read(hs.sred); }
    __device__ __forceinline__ void on_tile_max(float m) { if (tid == 0 && lmax) *lmax = fmaxf(*lmax, m); }
};

#ifndef GRAD_NREG
#define GRAD_NREG 10                // k-chunks of a layer's weight slice held in registers (pair16h.h PW): what this kernel's epilogue leaves room for
#endif
template <bool SAVE>
__global__ __launch_bounds__(256, 1) void sdf_grad_p_kernel(SdfGradPPtrs P, const float* __restrict__ pts, int64_t npts,
                                                           const float* __restrict__ act, float* __restrict__ asave,
                                                           float* __restrict__ normals, int save, float* __restrict__ gesave,
                                                           unsigned* __restrict__ absmax) {
    __shared__ __attribute__((aligned(16))) _Float16 simg[2][IMG_H];
    __shared__ __attribute__((aligned(16))) float saux[2][TM * LDA];
    __shared__ HScratchP hs;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    if (tid < 12) hs.lmax[tid] = 0.f;
    float wi[8];
    DH_UNROLL for (int l = 0; l < 8; ++l) wi[l] = winv_from_bits(P.wabs[l]);
    auto winv = [&](int l) { return l == 0 ? wi[0] : l == 1 ? wi[1] : l == 2 ? wi[2] : l == 3 ? wi[3] : l == 4 ? wi[4] : l == 5 ? wi[5] : l == 6 ? wi[6] : wi[7]; };
    auto rev = [&](int l) { return l == 1 ? P.rev1 : l == 2 ? P.rev2 : l == 3 ? P.rev3 : l == 4 ? P.rev4 : l == 5 ? P.rev5 : l == 6 ? P.rev6 : P.rev7; };
    const float w0 = P.w8row0[acc_col(wave, 0, lane)], w1 = P.w8row0[acc_col(wave, 1, lane)];
    float hmax = 0.f;                                             // RANGE WATCH of the forward chain: kernels_mlp_h.hip sdf_grad_h_kernel
    PW<GRAD_NREG> W;
    pw_load_all(W, P.rev7, wave, lane);
    _Float16* imgA = simg[0];
    _Float16* imgB = simg[1];
    int pit = 0;
    for (int64_t pr = blockIdx.x; pr < npairs; pr += gridDim.x, ++pit) {
        const bool son = pit == 2;                                // (-DDH_STAMPS: the workgroup's third pair is stamped)
        (void)son;
        PSTAMP(son, 0);
        const int64_t tileA = 2 * pr;
        const bool okB = tileA + 1 < ntiles;
        const int64_t tileB = okB ? tileA + 1 : tileA;
        f32x16 accA[MT][2], accB[MT][2];
        TileScale tsA, tsB;
        // the skip path's contribution to ge (true units) waits in the tile's LDS aux image from layer 4 to the end of the chain
        // (32 registers per lane that the phases do not have): every lane re-reads the words it wrote itself
        auto skip_to_lds = [&](int which, _Float16* img, float inv) {
            f32x16 ge[AUX_NTW];
            aux_zero(ge);
            gemm_auxout_hp(ge, img, 16, P.revaux4, wave, lane);
            DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
                const int col = aux_col(wave, tt, lane);
                if (col < AUXW) {
                    DH_UNROLL for (int r = 0; r < 16; ++r) saux[which][aux_row(wave, r, lane) * LDA + col] = ge[tt][r] * inv;
                }
            }
        };
        // a_7 = W8[0,:] * sigma'(z_7), both tiles (exposed)
        DH_UNROLL for (int which = 0; which < 2; ++which) {
            f32x16 (&acc)[MT][2] = which ? accB : accA;
            const int64_t tile = which ? tileB : tileA;
            acc_load_native_b(acc, tile_rsrc(act + ((int64_t)7 * ntiles + tile) * TILE_F), loff);
            DH_UNROLL for (int m = 0; m < MT; ++m) DH_UNROLL for (int t = 0; t < 2; ++t)
                DH_UNROLL for (int r = 0; r < 16; r += 2) hmax = fmaxf(hmax, fmaxf(acc[m][t][r], acc[m][t][r + 1]));
            acc_map(acc, [&](int, int t, int, float h) { float s, em; softplus_deriv_from_h(h, s, em); return (t ? w1 : w0) * s; });
            if (SAVE && (which == 0 || okB)) acc_store_native_b(acc, tile_rsrc(asave + ((int64_t)7 * ntiles + tile) * TILE_F), loff);
            tile_max_publish(hs.sred, wave, lane, acc_absmax(acc));
            __syncthreads();
            const float mx = tile_max_read(hs.sred);
            if (tid == 0) hs.lmax[7] = fmaxf(hs.lmax[7], mx);
            const TileScale ts = scale_for_max(mx);
            acc_to_lds_split(acc, which ? imgB : imgA, wave, lane, ts.S);
            __syncthreads();
            if (which) tsB = ts; else tsA = ts;
        }
        PSTAMP(son, 1);
        _Pragma("unroll 1") for (int l = 7; l >= 1; --l) {
            // ---- phase 1: G_A(l) over B's epilogue of GEMM l + 1 (-> a_l of B)
            acc_zero(accA);
            if (l == 7) {
                PNoEpi e;
                pair_phase<false>(accA, imgA, W, P.rev7, P.rev7, wave, lane, e);
            } else {
                SdfGradEpi<SAVE> e(accB, imgB, hs, &hs.lmax[l], wave, lane, tid, tsB.inv * winv(l + 1),
                                   tile_rsrc(act + ((int64_t)l * ntiles + tileB) * TILE_F),
                                   tile_rsrc_if(asave + ((int64_t)l * ntiles + tileB) * TILE_F, okB), loff, okB, hmax);
                if (son && l == 5) e.stamp = 32;
                pair_phase<false>(accA, imgA, W, rev(l), P.rev7, wave, lane, e);
                tsB.S = e.S; tsB.inv = e.inv_S;
            }
            if (l == 4) skip_to_lds(0, imgA, tsA.inv * wi[4]);    // skip path of A; image A still holds a_4
            PSTAMP(son, 2 + 4 * (7 - l));
            pair_barrier();
            PSTAMP(son, 3 + 4 * (7 - l));
            // ---- phase 2: G_B(l) over A's epilogue of GEMM l (-> a_{l-1} of A); W takes rev[l - 1] (after l = 1: the next pair's rev[7])
            acc_zero(accB);
            {
                SdfGradEpi<SAVE> e(accA, imgA, hs, &hs.lmax[l - 1], wave, lane, tid, tsA.inv * winv(l),
                                   tile_rsrc(act + ((int64_t)(l - 1) * ntiles + tileA) * TILE_F),
                                   tile_rsrc(asave + ((int64_t)(l - 1) * ntiles + tileA) * TILE_F), loff, true, hmax);
                if (son && l == 5) e.stamp = 64;
                pair_phase<true>(accB, imgB, W, rev(l), l == 1 ? P.rev7 : rev(l - 1), wave, lane, e);
                if (l == 4) skip_to_lds(1, imgB, tsB.inv * wi[4]);    // skip path of B; image B holds a_4 until B's next epilogue
                tsA.S = e.S; tsA.inv = e.inv_S;
            }
            PSTAMP(son, 4 + 4 * (7 - l));
            pair_barrier();
            PSTAMP(son, 5 + 4 * (7 - l));
        }
        {   // B's epilogue of GEMM 1 (-> a_0), with nothing above it
            SdfGradEpi<SAVE> e(accB, imgB, hs, &hs.lmax[0], wave, lane, tid, tsB.inv * wi[1],
                               tile_rsrc(act + tileB * TILE_F), tile_rsrc_if(asave + tileB * TILE_F, okB), loff, okB, hmax);
            pair_epi_alone_from<0>(e);
            tsB.S = e.S; tsB.inv = e.inv_S;
        }
        __syncthreads();
        PSTAMP(son, 30);
        DH_UNROLL for (int which = 0; which < 2; ++which) {       // ge += a_0 W_0, then ge -> the tile's LDS aux image
            f32x16 g0[AUX_NTW];
            aux_zero(g0);
            gemm_auxout_hp(g0, which ? imgB : imgA, 16, P.revaux0, wave, lane);
            const float inv = (which ? tsB.inv : tsA.inv) * wi[0];
            DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
                const int col = aux_col(wave, tt, lane);
                if (col < AUXW) {
                    DH_UNROLL for (int r = 0; r < 16; ++r) {
                        float* q = &saux[which][aux_row(wave, r, lane) * LDA + col];
                        *q = fmaf(g0[tt][r], inv, *q);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 2 * TM) {
            const int which = tid >> 6, p = tid & 63;
            const int64_t gp = (which ? tileB : tileA) * TM + p;
            if (gp < npts && (which == 0 || okB)) {
                const float* g = saux[which] + p * LDA;
                float n[3];
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float x = pts[gp * 3 + c];
                    float v = g[c];
                    DH_UNROLL for (int k = 0; k < 6; ++k) {
                        const float f = (float)(1 << k);
                        float sn, co; sincosf(x * f, &sn, &co);
                        v += f * (co * g[3 + 6 * k + c] - sn * g[3 + 6 * k + 3 + c]);
                    }
                    n[c] = v;
                }
                normals[gp * 3 + 0] = n[0]; normals[gp * 3 + 1] = n[1]; normals[gp * 3 + 2] = n[2];
                if (save == 2) { for (int c = 0; c < 40; ++c) gesave[gp * 40 + c] = c < EMB ? g[c] : 0.f; }     // pose refinement
            }
        }
        __syncthreads();
        PSTAMP(son, 31);
    }
    if (SAVE && absmax && tid < 8) post_class_max(absmax, ABSMAX_ASAVE + tid, hs.lmax[tid]);
    if (absmax) {                                      // (also forward-only renders)
        hmax = wave_max(hmax);
        if (lane == 0) post_class_max(absmax, ABSMAX_ACT, hmax);
    }
}

// ------------------------------------------------------------------------------------------------ colour backward
enum : int { PTP_COL_B0 = 12, PTP_COL_W4 = 16, PTP_COL_B4 = 19 };           // kernels_mlp_h.hip's tile partial-sum slots (workspace.h tpart)
struct ColBwdPPtrs {
    const u32x4 *rev0, *rev1, *rev2, *rev3;
    const u32x4* revaux;
    const float* w4;
    const unsigned* wabs;          // [0..3]
};
static inline ColBwdPPtrs make_colbwdp_ptrs(const float* packed) {
    ColBwdPPtrs C;
    auto rv = [&](int l) { return reinterpret_cast<const u32x4*>(packed + PACKH.col_rev_main[l]); };
    C.rev0 = rv(0); C.rev1 = rv(1); C.rev2 = rv(2); C.rev3 = rv(3);
    C.revaux = reinterpret_cast<const u32x4*>(packed + PACKH.col_rev_aux0);
    C.w4 = packed + PACK.col_w4;
    C.wabs = reinterpret_cast<const unsigned*>(packed + PACKH.wabs) + N_SDF;
    return C;
}
// epilogue behind the GEMM hbar_l = zbar_l W_l:  l >= 1: zbar_{l-1} = hbar_l * [h_l > 0] (h_l = the saved ReLU activation cact[l-1], through
// a register ring), the saved tile czbar[l-1], its column sums (bias gradient), the tile maximum (hand-off scale, tmax, class maximum)
// and the hand-off; LAST (l = 0): featbar = hbar_0 / (S S_w), stored, its maximum published to sred2 -- no hand-off
template <bool LAST>
struct ColBwdEpi : PEpi<ColBwdEpi<LAST>, 4, 3, !LAST> {
    static constexpr int RING = 5;
    f32x16 (&acc)[MT][2];
    HScratchP& hs;
    float* lmax;
    unsigned* tslot;
    float* csum;                   // tpart row of the column sums
    const int wave, lane, tid;
    const float inv;
    const rsrc_t hr, st;           // cact[l-1] (read; unused when LAST), czbar[l-1] / featbar (written)
    const int loff;
    const bool store;
    float cs0 = 0.f, cs1 = 0.f;
    f32x4 hq[RING];
    __device__ __forceinline__ ColBwdEpi(f32x16 (&acc_)[MT][2], _Float16* img_, HScratchP& hs_, float* lmax_, unsigned* tslot_, float* csum_,
                                         int wave_, int lane_, int tid_, float inv_, rsrc_t hr_, rsrc_t st_, int loff_, bool store_)
        : acc(acc_), hs(hs_), lmax(lmax_), tslot(tslot_), csum(csum_), wave(wave_), lane(lane_), tid(tid_), inv(inv_), hr(hr_),
          st(st_), loff(loff_), store(store_) {
        if constexpr (!LAST) {
            this->handoff_init(img_, wave_, lane_);
            DH_UNROLL for (int g = 0; g < RING; ++g) hq[g] = tile_ld(hr, loff, g);
        }
    }
    template <int G, int SUB>
    __device__ __forceinline__ void elem() {
        constexpr int m = G / 8, t = (G / 4) % 2, r4 = G % 4;
        if constexpr (SUB == 0 || SUB == 1) {
            constexpr int i = 2 * SUB, r = 4 * r4 + i;
            if constexpr (LAST) {
                acc[m][t][r] *= inv;
                acc[m][t][r + 1] *= inv;
            } else {
                const f32x4& h = hq[G % RING];
                acc[m][t][r] = h[i] > 0.f ? acc[m][t][r] * inv : 0.f;
                acc[m][t][r + 1] = h[i + 1] > 0.f ? acc[m][t][r + 1] * inv : 0.f;
            }
        } else if constexpr (SUB == 2) {
            f32x4 v;
            v[0] = acc[m][t][4 * r4 + 0]; v[1] = acc[m][t][4 * r4 + 1]; v[2] = acc[m][t][4 * r4 + 2]; v[3] = acc[m][t][4 * r4 + 3];
            tile_st(st, loff, G, v);
            if constexpr (!LAST && G + RING < 16) hq[G % RING] = tile_ld(hr, loff, G + RING);
            if constexpr (!LAST) {                              // tile_colsum's order: m, then r, per column t
                float& cs = t ? cs1 : cs0;
                cs += acc[m][t][4 * r4 + 0]; cs += acc[m][t][4 * r4 + 1]; cs += acc[m][t][4 * r4 + 2]; cs += acc[m][t][4 * r4 + 3];
            }
        } else {
            this->m0 = fmaxf(this->m0, fmaxf(fabsf(acc[m][t][4 * r4 + 0]), fabsf(acc[m][t][4 * r4 + 1])));
            this->m1 = fmaxf(this->m1, fmaxf(fabsf(acc[m][t][4 * r4 + 2]), fabsf(acc[m][t][4 * r4 + 3])));
        }
    }
    __device__ __forceinline__ void publish(float wm) {
        if constexpr (LAST) {
            tile_max_publish(hs.sred2, wave, lane, wm);        // read behind the barrier that ends the pair
        } else {
            cs0 += __shfl_xor(cs0, 32); cs1 += __shfl_xor(cs1, 32);
            if (lane < 32 && store) { csum[64 * wave + lane] = cs0; csum[64 * wave + 32 + lane] = cs1; }
            tile_max_publish(hs.sred, wave, lane, wm);
        }
    }
    __device__ __forceinline__ float read_tile_max() { return tile_max_read(hs.sred); }
    __device__ __forceinline__ void on_tile_max(float m) {
        if (tid == 0) {
            *lmax = fmaxf(*lmax, m);
            if (store) *tslot = __builtin_bit_cast(unsigned, m);
        }
    }
};

#ifndef COLBWD_NREG
#define COLBWD_NREG 8
#endif
__global__ __launch_bounds__(256, 1) void color_bwd_p_kernel(ColBwdPPtrs C, const float* __restrict__ colors, const float* __restrict__ d_colors,
                                                            int64_t npts, const float* __restrict__ cact, float* __restrict__ czbar,
                                                            float* __restrict__ featbar, float* __restrict__ d_normals,
                                                            float* __restrict__ tpart, unsigned* __restrict__ absmax,
                                                            unsigned* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) _Float16 simg[2][IMG_H];
    __shared__ __attribute__((aligned(16))) float scraw[2][TM * 4];
    __shared__ HScratchP hs;
    __shared__ float sfb[2][4];                                   // featbar maxima of the pair's two tiles
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int loff = tile_loff(wave, lane);
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    if (tid < 12) hs.lmax[tid] = 0.f;
    const float wi0 = winv_from_bits(C.wabs[0]), wi1 = winv_from_bits(C.wabs[1]), wi2 = winv_from_bits(C.wabs[2]), wi3 = winv_from_bits(C.wabs[3]);
    auto winv = [&](int l) { return l == 0 ? wi0 : l == 1 ? wi1 : l == 2 ? wi2 : wi3; };
    auto rev = [&](int l) { return l == 0 ? C.rev0 : l == 1 ? C.rev1 : l == 2 ? C.rev2 : C.rev3; };
    PW<COLBWD_NREG> W;
    _Float16* imgA = simg[0];
    _Float16* imgB = simg[1];
    for (int64_t pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
        const int64_t tileA = 2 * pr;
        const bool okB = tileA + 1 < ntiles;
        const int64_t tileB = okB ? tileA + 1 : tileA;
        float* tpA = tpart + tileA * N_TILE_PART * 256;
        float* tpB = tpart + tileB * N_TILE_PART * 256;
        if (tid < 2 * TM) {
            const int which = tid >> 6, p = tid & 63;
            const int64_t gp = (which ? tileB : tileA) * TM + p;
            DH_UNROLL for (int j = 0; j < 3; ++j) {
                float v = 0.f;
                if (gp < npts) { const float c = colors[gp * 3 + j]; v = d_colors[gp * 3 + j] * c * (1.f - c); }
                scraw[which][p * 4 + j] = v;
            }
            scraw[which][p * 4 + 3] = 0.f;
        }
        __syncthreads();
        if (tid < 3 || (tid >= 64 && tid < 67)) {                 // db4
            const int which = tid >> 6, j = tid & 63;
            float sum = 0.f;
            for (int r = 0; r < TM; ++r) sum += scraw[which][r * 4 + j];
            if (which == 0 || okB) (which ? tpB : tpA)[PTP_COL_B4 * 256 + j] = sum;
        }
        f32x16 accA[MT][2], accB[MT][2];
        TileScale tsA, tsB;
        DH_UNROLL for (int which = 0; which < 2; ++which) {       // lin4: dW4 partials, zbar_3 = (craw W4) * [h4 > 0]; both tiles (exposed)
            f32x16 (&acc)[MT][2] = which ? accB : accA;
            const int64_t tile = which ? tileB : tileA;
            float* tp = which ? tpB : tpA;
            const bool on = which == 0 || okB;
            acc_load_native_b(acc, tile_rsrc(cact + ((int64_t)3 * ntiles + tile) * TILE_F), loff);
            {
                const int col0 = acc_col(wave, 0, lane), col1 = acc_col(wave, 1, lane);
                float w4[3][2];
                DH_UNROLL for (int j = 0; j < 3; ++j) { w4[j][0] = C.w4[j * 256 + col0]; w4[j][1] = C.w4[j * 256 + col1]; }
                float dw[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
                DH_UNROLL for (int m = 0; m < MT; ++m)
                    DH_UNROLL for (int r = 0; r < 16; ++r) {
                        const f32x4 cr = *reinterpret_cast<const f32x4*>(scraw[which] + acc_row(m, r, lane) * 4);
                        DH_UNROLL for (int t = 0; t < 2; ++t) {
                            const float h = acc[m][t][r];
                            DH_UNROLL for (int j = 0; j < 3; ++j) dw[j][t] = fmaf(cr[j], h, dw[j][t]);
                            const float hb = cr[0] * w4[0][t] + cr[1] * w4[1][t] + cr[2] * w4[2][t];
                            acc[m][t][r] = h > 0.f ? hb : 0.f;
                        }
                    }
                DH_UNROLL for (int j = 0; j < 3; ++j)
                    DH_UNROLL for (int t = 0; t < 2; ++t) {
                        float sum = dw[j][t];
                        sum += __shfl_xor(sum, 32);
                        if (lane < 32 && on) tp[(PTP_COL_W4 + j) * 256 + 64 * wave + 32 * t + lane] = sum;
                    }
            }
            if (on) {
                acc_store_native_b(acc, tile_rsrc(czbar + ((int64_t)3 * ntiles + tile) * TILE_F), loff);
                tile_colsum(acc, tp + (PTP_COL_B0 + 3) * 256, wave, lane);
            }
            tile_max_publish(hs.sred, wave, lane, acc_absmax(acc));
            __syncthreads();
            const float mx = tile_max_read(hs.sred);
            if (tid == 0) {
                hs.lmax[3] = fmaxf(hs.lmax[3], mx);
                if (on) tmax[(TMAX_CZBAR + 3) * ntiles + tile] = __builtin_bit_cast(unsigned, mx);
            }
            const TileScale ts = scale_for_max(mx);
            acc_to_lds_split(acc, which ? imgB : imgA, wave, lane, ts.S);
            __syncthreads();
            if (which) tsB = ts; else tsA = ts;
        }
        // the first layer's weight slice: requested here, not by the previous pair's last phase -- held across the lin4 prologue above (its
        // dW4 partials beside two accumulator sets) the slice was spilled to scratch and back (250 registers)
        pw_load_all(W, C.rev3, wave, lane);
        _Pragma("unroll 1") for (int l = 3; l >= 1; --l) {
            // ---- phase 1: G_A(l) over B's epilogue of GEMM l + 1
            acc_zero(accA);
            if (l == 3) {
                PNoEpi e;
                pair_phase<false>(accA, imgA, W, C.rev3, C.rev3, wave, lane, e);
            } else {
                ColBwdEpi<false> e(accB, imgB, hs, &hs.lmax[l], tmax + (TMAX_CZBAR + l) * ntiles + tileB, tpB + (PTP_COL_B0 + l) * 256, wave, lane, tid,
                                   tsB.inv * winv(l + 1), tile_rsrc(cact + ((int64_t)l * ntiles + tileB) * TILE_F),
                                   tile_rsrc_if(czbar + ((int64_t)l * ntiles + tileB) * TILE_F, okB), loff, okB);
                pair_phase<false>(accA, imgA, W, rev(l), C.rev3, wave, lane, e);
                tsB.S = e.S; tsB.inv = e.inv_S;
            }
            pair_barrier();
            // ---- phase 2: G_B(l) over A's epilogue of GEMM l; W takes rev[l - 1]
            acc_zero(accB);
            {
                ColBwdEpi<false> e(accA, imgA, hs, &hs.lmax[l - 1], tmax + (TMAX_CZBAR + l - 1) * ntiles + tileA, tpA + (PTP_COL_B0 + l - 1) * 256, wave,
                                   lane, tid, tsA.inv * winv(l), tile_rsrc(cact + ((int64_t)(l - 1) * ntiles + tileA) * TILE_F),
                                   tile_rsrc(czbar + ((int64_t)(l - 1) * ntiles + tileA) * TILE_F), loff, true);
                pair_phase<true>(accB, imgB, W, rev(l), rev(l - 1), wave, lane, e);
                tsA.S = e.S; tsA.inv = e.inv_S;
            }
            pair_barrier();
        }
        // lin0 (peeled out of the loop: its second phase neither reloads the weights nor hands an image off -- as a branch inside the
        // loop the two forms of phase 2 met with different register assignments of the slice and the compiler swapped it through scratch)
        acc_zero(accA);
        {
            ColBwdEpi<false> e(accB, imgB, hs, &hs.lmax[0], tmax + TMAX_CZBAR * ntiles + tileB, tpB + PTP_COL_B0 * 256, wave, lane, tid,
                               tsB.inv * wi1, tile_rsrc(cact + tileB * TILE_F), tile_rsrc_if(czbar + tileB * TILE_F, okB), loff, okB);
            pair_phase<false>(accA, imgA, W, C.rev0, C.rev0, wave, lane, e);
            tsB.S = e.S; tsB.inv = e.inv_S;
        }
        pair_barrier();
        acc_zero(accB);
        {
            ColBwdEpi<true> e(accA, imgA, hs, nullptr, nullptr, nullptr, wave, lane, tid, tsA.inv * wi0, tile_rsrc(featbar + tileA * TILE_F),
                              tile_rsrc(featbar + tileA * TILE_F), loff, true);
            pair_phase<false>(accB, imgB, W, C.rev0, C.rev0, wave, lane, e);
            if (lane == 0) sfb[0][wave] = hs.sred2[wave];         // (each wave moves its own word: sred2 is B's next)
        }
        pair_barrier();
        {   // B's featbar, with nothing above it
            ColBwdEpi<true> e(accB, imgB, hs, nullptr, nullptr, nullptr, wave, lane, tid, tsB.inv * wi0, tile_rsrc(featbar + tileB * TILE_F),
                              tile_rsrc_if(featbar + tileB * TILE_F, okB), loff, okB);
            pair_epi_alone_from<0>(e);
            if (lane == 0) sfb[1][wave] = hs.sred2[wave];
        }
        // extras adjoint = zbar_0 W0[:, :33] (only the normal columns 30..32 matter); the images still hold zbar_0
        DH_UNROLL for (int which = 0; which < 2; ++which) {
            f32x16 a2[AUX_NTW];
            aux_zero(a2);
            gemm_auxout_hp(a2, which ? imgB : imgA, 16, C.revaux, wave, lane);
            const float inv0 = (which ? tsB.inv : tsA.inv) * wi0;
            DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
                const int col = aux_col(wave, tt, lane);
                if (col >= 30 && col < 33 && (which == 0 || okB)) {
                    DH_UNROLL for (int r = 0; r < 16; ++r) {
                        const int64_t gp = (which ? tileB : tileA) * TM + aux_row(wave, r, lane);
                        if (gp < npts) d_normals[gp * 3 + (col - 30)] += a2[tt][r] * inv0;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 2) {                                           // featbar's maxima
            const float m = fmaxf(fmaxf(sfb[tid][0], sfb[tid][1]), fmaxf(sfb[tid][2], sfb[tid][3]));
            if (tid == 0 || okB) {
                atomicMax(reinterpret_cast<unsigned*>(&hs.lmax[4]), __builtin_bit_cast(unsigned, m));      // (>= 0: the bit patterns order like the values)
                tmax[TMAX_FEATBAR * ntiles + (tid ? tileB : tileA)] = __builtin_bit_cast(unsigned, m);
            }
        }
        __syncthreads();
    }
    if (absmax) {
        if (tid < 4) post_class_max(absmax, ABSMAX_CZBAR + tid, hs.lmax[tid]);
        if (tid == 4) post_class_max(absmax, ABSMAX_FEATBAR, hs.lmax[4]);
    }
}

}  // namespace dh
#ifdef DH_STAMPS
extern "C" int dh_dev_read_stamps_p(unsigned long long* host, long long n) {
    const long long total = (long long)(sizeof(dh::dh_pstamps) / sizeof(unsigned long long));
    if (n > total) n = total;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dh::dh_pstamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#endif
namespace dh {
// ------------------------------------------------------------------------------------------------ launchers
static inline int okp() { return hipGetLastError() == hipSuccess ? 0 : -3; }
// one workgroup per CU (156 KB of LDS): what the device offers, asked once per device
static int pair_cus() {
    static int cache[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
    if (cache[dev] == 0) {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess) lds = 0;
        cache[dev] = (lds >= PAIR_LDS_BYTES && cus > 0) ? cus : -1;
    }
    return cache[dev] > 0 ? cache[dev] : 0;
}
bool pair_form_available() { return pair_cus() > 0; }
int pair_form_cus() { return pair_cus(); }
int launch_color_fwd_p(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                       const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, unsigned* absmax,
                       hipStream_t stream) {
    const int cus = pair_cus();
    if (cus <= 0) return -2;
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    const int g = (int)(npairs < cus ? npairs : cus);
    if (save) hipLaunchKernelGGL(color_fwd_p_kernel<true>, dim3(g), dim3(256), 0, stream, make_colp_ptrs(packed), pts, dirs, n_per_ray, normals,
                                 feat, npts, color, cact, caux, absmax);
    else hipLaunchKernelGGL(color_fwd_p_kernel<false>, dim3(g), dim3(256), 0, stream, make_colp_ptrs(packed), pts, dirs, n_per_ray, normals,
                            feat, npts, color, cact, caux, absmax);
    return okp();
}

}  // namespace dh

namespace dh {
int launch_sdf_grad_p(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals, int save,
                      float* gesave, unsigned* absmax, hipStream_t stream) {
    const int cus = pair_cus();
    if (cus <= 0) return -2;
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    const int g = (int)(npairs < cus ? npairs : cus);
    if (save) hipLaunchKernelGGL(sdf_grad_p_kernel<true>, dim3(g), dim3(256), 0, stream, make_sdfgradp_ptrs(packed), pts, npts, act, asave,
                                 normals, save, gesave, absmax);
    else hipLaunchKernelGGL(sdf_grad_p_kernel<false>, dim3(g), dim3(256), 0, stream, make_sdfgradp_ptrs(packed), pts, npts, act, asave,
                            normals, save, gesave, absmax);
    return okp();
}
}  // namespace dh

namespace dh {
int launch_color_bwd_p(const float* packed, const float* colors, const float* d_colors, int64_t npts, const float* cact, float* czbar,
                       float* featbar, float* d_normals, float* tpart, unsigned* absmax, unsigned* tmax, hipStream_t stream) {
    const int cus = pair_cus();
    if (cus <= 0) return -2;
    const int64_t ntiles = (npts + TM - 1) / TM, npairs = (ntiles + 1) / 2;
    const int g = (int)(npairs < cus ? npairs : cus);
    hipLaunchKernelGGL(color_bwd_p_kernel, dim3(g), dim3(256), 0, stream, make_colbwdp_ptrs(packed), colors, d_colors, npts, cact, czbar,
                       featbar, d_normals, tpart, absmax, tmax);
    return okp();
}
}  // namespace dh
