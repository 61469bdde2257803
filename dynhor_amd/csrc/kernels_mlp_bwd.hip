// Backward MLP chains (DESIGN.md "Backward").  Notation for the SDF net, layer l:
//   forward   z_l = W_l in_l + b_l,  h_{l+1} = sigma(z_l)            (saved: act[l] = h_{l+1})
//   reverse   a_l = u_{l+1} * sigma'(z_l),  u_l = a_l W_l             (saved: asave[l] = a_l)      -> n = J_e^T ge
// Given the adjoints nbar (d loss/d n), sdfbar, featbar the backward needs, per layer,
//   tangent   abar_l = tt_l W_l^T,  t_{l+1} = sigma'(z_l) * abar_l    (tt_0 = J_e nbar; forward direction)
//             r_l = abar_l * a_l * beta * (1 - sigma'(z_l))           (second-order term: u_{l+1} sigma''(z_l) abar_l)
//   backward  zbar_l = hbar_{l+1} * sigma'(z_l) + r_l,  hbar_l = zbar_l W_l
//   weights   Wbar_l = zbar_l^T in_l + a_l^T tt_l  (dw.hip),  bbar_l = colsum(zbar_l)
// because the adjoint of the reverse-mode pass u -> a -> u is a forward-mode (tangent) pass in direction J_e nbar.
#include "tile.h"
#include "kernels.h"
#include "mlp_common.h"
#include "workspace.h"
#include "stamps.h"

DH_STAMP_READER(dh_dev_read_stamps_bwd)

namespace dh {

// tile partial-sum slots (workspace.h: tpart [nt][N_TILE_PART][256])
enum : int { TP_SDF_B0 = 0, TP_SDF_B8 = 8, TP_W8ROW0_T = 9, TP_W8ROW0_S = 10, TP_SCAL = 11, TP_COL_B0 = 12,
             TP_COL_W4 = 16, TP_COL_B4 = 19 };

// ------------------------------------------------------------------------------------------------
// K6: RenderingNetwork backward.  d_colors is wrt the post-sigmoid colour.
// ------------------------------------------------------------------------------------------------
template <bool RAYS>      // RAYS (pose refinement): as color_bwd_s_kernel<true> below -- the second arithmetic covers it too
__global__ __launch_bounds__(256, 2) void color_bwd_kernel(ColPtrs C, const float* __restrict__ colors,
                                                            const float* __restrict__ d_colors, int64_t npts,
                                                            const float* __restrict__ cact, float* __restrict__ czbar,
                                                            float* __restrict__ featbar, float* __restrict__ d_normals,
                                                            float* __restrict__ tpart, const float* __restrict__ dirs,
                                                            int n_per_ray, float* __restrict__ d_pts,
                                                            float* __restrict__ d_dirs_pts) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: craw [128][4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
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
        acc_load_native(acc, cact + ((int64_t)3 * ntiles + tile) * TILE_F, wave, lane);
        {
            const int col0 = acc_col(wave, 0, lane), col1 = acc_col(wave, 1, lane);
            float w4[3][2];              // re-read per tile (L1/L2 hits): six registers not held across the GEMMs
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
        acc_store_native(acc, czbar + ((int64_t)3 * ntiles + tile) * TILE_F, wave, lane);
        tile_colsum(acc, tp + (TP_COL_B0 + 3) * 256, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        BFrag pre = gemm_b_prefetch(C.rev_main[3], wave, lane);
        for (int l = 3; l >= 1; --l) {
            acc_zero(acc);
            gemm_rows(acc, smain, LDX, 32, C.rev_main[l], wave, lane, pre);               // hbar_l = zbar_l W_l
            pre = gemm_b_prefetch(C.rev_main[l - 1], wave, lane);
            const f32x4* hp = reinterpret_cast<const f32x4*>(cact + ((int64_t)(l - 1) * ntiles + tile) * TILE_F) + (size_t)wave * MT * 8 * 64 + lane;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = DH_TILE_LD(hp + ((m * 2 + t) * 4 + r4) * 64);
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr)
                            if (!(h[rr] > 0.f)) acc[m][t][4 * r4 + rr] = 0.f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc_store_native(acc, czbar + ((int64_t)(l - 1) * ntiles + tile) * TILE_F, wave, lane);
            tile_colsum(acc, tp + (TP_COL_B0 + l - 1) * 256, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        // lin0: featbar = zbar_0 W0[:,33:] ; extras adjoint = zbar_0 W0[:,:33] (only the normal columns 30..32 matter)
        acc_zero(acc);
        gemm_rows(acc, smain, LDX, 32, C.rev_main[0], wave, lane, pre);
        acc_store_native(acc, featbar + tile * TILE_F, wave, lane);
        f32x16 a2[AUX_NTW];
        aux_zero(a2);
        gemm_auxout(a2, smain, 32, C.rev_aux0, wave, lane);
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
    }
}

// K6 on the split-bf16 core.  RAYS (pose refinement): the first-layer input adjoint zbar_0 W0[:, :33] is also turned into the
// adjoints of the sample point (columns 0..2 -> d_pts, WRITTEN) and of the ray direction through the view embedding
// (columns 3..29 -> d_dirs_pts [npts,3], written; summed per ray by the caller).
template <bool RAYS>
__global__ __launch_bounds__(256, 2) void color_bwd_s_kernel(Col16Ptrs C, const float* __restrict__ colors,
                                                            const float* __restrict__ d_colors, int64_t npts,
                                                            const float* __restrict__ cact, float* __restrict__ czbar,
                                                            float* __restrict__ featbar, float* __restrict__ d_normals,
                                                            float* __restrict__ tpart, const float* __restrict__ dirs,
                                                            int n_per_ray, float* __restrict__ d_pts,
                                                            float* __restrict__ d_dirs_pts) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: craw [128][4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
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
        acc_load_native(acc, cact + ((int64_t)3 * ntiles + tile) * TILE_F, wave, lane);
        {
            const int col0 = acc_col(wave, 0, lane), col1 = acc_col(wave, 1, lane);
            float w4[3][2];              // re-read per tile (L1/L2 hits): six registers not held across the GEMMs
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
        acc_store_native(acc, czbar + ((int64_t)3 * ntiles + tile) * TILE_F, wave, lane);
        tile_colsum(acc, tp + (TP_COL_B0 + 3) * 256, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        for (int l = 3; l >= 1; --l) {
            acc_zero(acc);
            gemm_rows_s(acc, smain, LDX, 16, C.rev16[l], wave, lane);                      // hbar_l = zbar_l W_l
            const f32x4* hp = reinterpret_cast<const f32x4*>(cact + ((int64_t)(l - 1) * ntiles + tile) * TILE_F) + (size_t)wave * MT * 8 * 64 + lane;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = DH_TILE_LD(hp + ((m * 2 + t) * 4 + r4) * 64);
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr)
                            if (!(h[rr] > 0.f)) acc[m][t][4 * r4 + rr] = 0.f;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc_store_native(acc, czbar + ((int64_t)(l - 1) * ntiles + tile) * TILE_F, wave, lane);
            tile_colsum(acc, tp + (TP_COL_B0 + l - 1) * 256, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        // lin0: featbar = zbar_0 W0[:,33:] ; extras adjoint = zbar_0 W0[:,:33] (only the normal columns 30..32 matter)
        acc_zero(acc);
        gemm_rows_s(acc, smain, LDX, 16, C.rev16[0], wave, lane);
        acc_store_native(acc, featbar + tile * TILE_F, wave, lane);
        f32x16 a2[AUX_NTW];
        aux_zero(a2);
        gemm_auxout_s(a2, smain, 16, C.revaux16, wave, lane);
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
    }
}


// ------------------------------------------------------------------------------------------------
// K7a: tangent chain (forward direction) -> t_l, r_l ; colsum(t_8) feeds Wbar_8[0,:]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sdf_tangent_kernel(SdfPtrs P, const float* __restrict__ pts,
                                                              const float* __restrict__ d_normals, int64_t npts,
                                                              const float* __restrict__ act, const float* __restrict__ asave,
                                                              float* __restrict__ t0aux, float* __restrict__ tsave,
                                                              float* __restrict__ rsave, float* __restrict__ tpart) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
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
            if (part == 0) { row[0] = nb[0]; row[1] = nb[1]; row[2] = nb[2]; }
            if (part == 1) { for (int c = 39; c < LDA; ++c) row[c] = 0.f; }
            for (int k = part; k < 6; k += TPP) {
                const float f = (float)(1 << k);
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    float s, co;
                    sincosf(x[c] * f, &s, &co);
                    row[3 + 6 * k + c] = f * co * nb[c];
                    row[3 + 6 * k + 3 + c] = -f * s * nb[c];
                }
            }
        }
        __syncthreads();
        aux_lds_to_native(saux, t0aux + tile * AUXT_F, wave, lane);
        f32x16 acc[MT][2];
        BFrag pre = gemm_b_prefetch(P.fwd_main[1], wave, lane);
        for (int l = 0; l < 8; ++l) {
            acc_zero(acc);
            if (l > 0) gemm_rows(acc, smain, LDX, l == 4 ? 28 : 32, P.fwd_main[l], wave, lane, pre);
            if (l == 0 || l == 4) gemm_rows(acc, saux, LDA, 5, P.fwd_aux[l], wave, lane);     // abar_l
            if (l < 7) pre = gemm_b_prefetch(P.fwd_main[l + 1], wave, lane);
            const size_t woff = (size_t)wave * MT * 8 * 64 + lane;
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            const f32x4* ap = reinterpret_cast<const f32x4*>(asave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            f32x4* rp = reinterpret_cast<f32x4*>(rsave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const int idx = ((m * 2 + t) * 4 + r4) * 64;
                        const f32x4 h = hp[idx], a = ap[idx];
                        f32x4 rv;
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            const float ab = acc[m][t][4 * r4 + rr];
                            rv[rr] = ab * a[rr] * (SOFTPLUS_BETA * em);
                            acc[m][t][4 * r4 + rr] = s * ab;
                        }
                        rp[idx] = rv;
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (l < 7) {
                acc_store_native(acc, tsave + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);    // t_{l+1}
                __syncthreads();
                acc_to_lds(acc, smain, wave, lane);
                __syncthreads();
            } else {
                tile_colsum(acc, tp + TP_W8ROW0_T * 256, wave, lane);                                 // colsum t_8
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K7b: backward chain -> zbar_l (l = 7..0), bias-gradient partials, Wbar_8[0,:] partial
// ------------------------------------------------------------------------------------------------
template <bool RAYS>      // RAYS (pose refinement): as sdf_bwd_s_kernel<true> below
__global__ __launch_bounds__(256, 2) void sdf_bwd_kernel(SdfPtrs P, const float* __restrict__ d_sdf, int64_t npts,
                                                          const float* __restrict__ act, const float* __restrict__ rsave,
                                                          const float* __restrict__ featbar, float* __restrict__ zbar,
                                                          float* __restrict__ tpart, const float* __restrict__ pts,
                                                          const float* __restrict__ d_normals, const float* __restrict__ gesave,
                                                          float* __restrict__ d_pts) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: sdfbar [128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
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
        acc_load_native(acc, featbar + tile * TILE_F, wave, lane);
        tile_colsum(acc, tp + TP_SDF_B8 * 256, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        if (wave == 0) {                                       // sum of sdfbar -> bbar_8[0]
            float s = 0.f;
            for (int i = lane; i < TM; i += 64) s += saux[i];
            DH_UNROLL for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) tp[TP_SCAL * 256] = s;
        }
        // hbar_8 = featbar W8[1:,:] + sdfbar (x) W8[0,:]
        acc_zero(acc);
        gemm_rows(acc, smain, LDX, 32, P.rev_main[8], wave, lane);
        BFrag pre = gemm_b_prefetch(P.rev_main[7], wave, lane);
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int r = 0; r < 16; ++r) {
                const float sb = saux[acc_row(m, r, lane)];
                acc[m][0][r] = fmaf(sb, w0c0, acc[m][0][r]);
                acc[m][1][r] = fmaf(sb, w0c1, acc[m][1][r]);
            }
        for (int l = 7; l >= 0; --l) {
            const size_t woff = (size_t)wave * MT * 8 * 64 + lane;
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            const f32x4* rp = reinterpret_cast<const f32x4*>(rsave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            float ws0 = 0.f, ws1 = 0.f;                         // sum_rows sdfbar * h_8 (l == 7 only)
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const int idx = ((m * 2 + t) * 4 + r4) * 64;
                        const f32x4 h = hp[idx], rv = rp[idx];
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
            acc_store_native(acc, zbar + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);
            tile_colsum(acc, tp + (TP_SDF_B0 + l) * 256, wave, lane);
            if (l > 0) {
                __syncthreads();
                acc_to_lds(acc, smain, wave, lane);
                __syncthreads();
                if (RAYS && l == 4) gemm_auxout(eb, smain, 32, P.rev_aux[4], wave, lane);         // skip path -> ebar
                acc_zero(acc);
                gemm_rows(acc, smain, LDX, 32, P.rev_main[l], wave, lane, pre);     // hbar_l = zbar_l W_l
                if (l > 1) pre = gemm_b_prefetch(P.rev_main[l - 1], wave, lane);
            } else if (RAYS) {
                __syncthreads();
                acc_to_lds(acc, smain, wave, lane);                                 // zbar_0
                __syncthreads();
                gemm_auxout(eb, smain, 32, P.rev_aux[0], wave, lane);               // ebar += zbar_0 W_0
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
            }
        }
        __syncthreads();
    }
}

// K7a, split-on-fetch
__global__ __launch_bounds__(256, 2) void sdf_tangent_s_kernel(Sdf16Ptrs P, const float* __restrict__ pts,
                                                              const float* __restrict__ d_normals, int64_t npts,
                                                              const float* __restrict__ act, const float* __restrict__ asave,
                                                              float* __restrict__ t0aux, float* __restrict__ tsave,
                                                              float* __restrict__ rsave, float* __restrict__ tpart) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    int it = 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        float* tp = tpart + tile * N_TILE_PART * 256;
        {   // tt_0 = J_e(x) nbar
            const int p = tid & (TM - 1), part = tid / TM;
            const int64_t gp = tile * TM + p;
            float x[3] = {0.f, 0.f, 0.f}, nb[3] = {0.f, 0.f, 0.f};
            if (gp < npts) {
                DH_UNROLL for (int c = 0; c < 3; ++c) { x[c] = pts[gp * 3 + c]; nb[c] = d_normals[gp * 3 + c]; }
            }
            float* row = saux + p * LDA;
            if (part == 0) { row[0] = nb[0]; row[1] = nb[1]; row[2] = nb[2]; }
            if (part == 1) { for (int c = 39; c < LDA; ++c) row[c] = 0.f; }
            for (int k = part; k < 6; k += TPP) {
                const float f = (float)(1 << k);
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    float s, co;
                    sincosf(x[c] * f, &s, &co);
                    row[3 + 6 * k + c] = f * co * nb[c];
                    row[3 + 6 * k + 3 + c] = -f * s * nb[c];
                }
            }
        }
        __syncthreads();
        aux_lds_to_native(saux, t0aux + tile * AUXT_F, wave, lane);
        f32x16 acc[MT][2];
        for (int l = 0; l < 8; ++l) {
            DH_STAMP(it, l, 0);
            acc_zero(acc);
            if (l > 0) gemm_rows_s(acc, smain, LDX, l == 4 ? 14 : 16, P.main16[l], wave, lane);
            if (l == 0 || l == 4) gemm_rows_s(acc, saux, LDA, AUX_KC, P.aux16[l], wave, lane);     // abar_l
            DH_STAMP(it, l, 1);
            const size_t woff = (size_t)wave * MT * 8 * 64 + lane;
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            const f32x4* ap = reinterpret_cast<const f32x4*>(asave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            f32x4* rp = reinterpret_cast<f32x4*>(rsave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const int idx = ((m * 2 + t) * 4 + r4) * 64;
                        const f32x4 h = DH_TILE_LD(hp + idx), a = DH_TILE_LD(ap + idx);
                        f32x4 rv;
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            const float ab = acc[m][t][4 * r4 + rr];
                            rv[rr] = ab * a[rr] * (SOFTPLUS_BETA * em);
                            acc[m][t][4 * r4 + rr] = s * ab;
                        }
                        DH_TILE_ST(rp + idx, rv);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            DH_STAMP(it, l, 2);
            if (l < 7) {
                acc_store_native(acc, tsave + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);    // t_{l+1}
                DH_STAMP(it, l, 3);
                __syncthreads();
                DH_STAMP(it, l, 4);
                acc_to_lds(acc, smain, wave, lane);
                DH_STAMP(it, l, 5);
                __syncthreads();
                DH_STAMP(it, l, 6);
            } else {
                tile_colsum(acc, tp + TP_W8ROW0_T * 256, wave, lane);                                 // colsum t_8
            }
        }
        __syncthreads();
    }
}

// K7b, split-on-fetch.  RAYS (pose refinement): additionally the adjoint of the sample points through the SDF network,
//   ebar = zbar_0 W_0 + zbar_4 W_4[:, 217:]/sqrt2 (adjoint of the embedding; the matrices of the normal pass),
//   xbar = J_e(x)^T ebar  +  nbar * d/dx [J_e(x)^T] ge        (second term: the direct x-dependence of n = J_e(x)^T ge),
// ACCUMULATED onto d_pts (the colour backward wrote its part first).  ge was saved by dh_sdf_gradient(save = 2).
template <bool RAYS>
__global__ __launch_bounds__(256, 2) void sdf_bwd_s_kernel(Sdf16Ptrs P, const float* __restrict__ d_sdf, int64_t npts,
                                                          const float* __restrict__ act, const float* __restrict__ rsave,
                                                          const float* __restrict__ featbar, float* __restrict__ zbar,
                                                          float* __restrict__ tpart, const float* __restrict__ pts,
                                                          const float* __restrict__ d_normals, const float* __restrict__ gesave,
                                                          float* __restrict__ d_pts) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];     // scratch: sdfbar [128]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
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
        acc_load_native(acc, featbar + tile * TILE_F, wave, lane);
        tile_colsum(acc, tp + TP_SDF_B8 * 256, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        if (wave == 0) {                                       // sum of sdfbar -> bbar_8[0]
            float s = 0.f;
            for (int i = lane; i < TM; i += 64) s += saux[i];
            DH_UNROLL for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if (lane == 0) tp[TP_SCAL * 256] = s;
        }
        // hbar_8 = featbar W8[1:,:] + sdfbar (x) W8[0,:]
        acc_zero(acc);
        gemm_rows_s(acc, smain, LDX, 16, P.rev16[8], wave, lane);
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int r = 0; r < 16; ++r) {
                const float sb = saux[acc_row(m, r, lane)];
                acc[m][0][r] = fmaf(sb, w0c0, acc[m][0][r]);
                acc[m][1][r] = fmaf(sb, w0c1, acc[m][1][r]);
            }
        for (int l = 7; l >= 0; --l) {
            const size_t woff = (size_t)wave * MT * 8 * 64 + lane;
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            const f32x4* rp = reinterpret_cast<const f32x4*>(rsave + ((int64_t)l * ntiles + tile) * TILE_F) + woff;
            float ws0 = 0.f, ws1 = 0.f;                         // sum_rows sdfbar * h_8 (l == 7 only)
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const int idx = ((m * 2 + t) * 4 + r4) * 64;
                        const f32x4 h = DH_TILE_LD(hp + idx), rv = DH_TILE_LD(rp + idx);
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
            acc_store_native(acc, zbar + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);
            tile_colsum(acc, tp + (TP_SDF_B0 + l) * 256, wave, lane);
            if (l > 0) {
                __syncthreads();
                acc_to_lds(acc, smain, wave, lane);
                __syncthreads();
                if (RAYS && l == 4) gemm_auxout_s(eb, smain, 16, P.revaux16[4], wave, lane);      // skip path -> ebar
                acc_zero(acc);
                gemm_rows_s(acc, smain, LDX, 16, P.rev16[l], wave, lane);           // hbar_l = zbar_l W_l
            } else if (RAYS) {
                __syncthreads();
                acc_to_lds(acc, smain, wave, lane);                                 // zbar_0
                __syncthreads();
                gemm_auxout_s(eb, smain, 16, P.revaux16[0], wave, lane);            // ebar += zbar_0 W_0
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
            }
        }
        __syncthreads();
    }
}

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }
static inline int grid_for(int64_t npts, int grid) {
    const int64_t ntiles = (npts + TM - 1) / TM;
    return (int)(ntiles < grid ? ntiles : grid);
}

int launch_color_bwd(const float* packed, const float* colors, const float* d_colors, int64_t npts, const float* cact,
                     float* czbar, float* featbar, float* d_normals, float* tpart, float* absmax, int grid, int arith, hipStream_t st) {
    const int form = arith >> 8;                              // include/dynhor_hip.h DH_CHAIN_FORM_*: 0 auto, 1 tile, 2 pair
    arith &= 0xff;
    if (arith == ARITH_F16) {
        unsigned* am = reinterpret_cast<unsigned*>(absmax);
        if (use_pair_form(form, npts, false)) return launch_color_bwd_p(packed, colors, d_colors, npts, cact, czbar, featbar, d_normals, tpart, am,
                                                                 am + ABSMAX_FLOATS, st);
        return launch_color_bwd_h(packed, colors, d_colors, nullptr, 1, npts, cact, czbar, featbar, d_normals, tpart, nullptr, nullptr, am,
                                  am + ABSMAX_FLOATS, grid, st);
    }
    if (arith == ARITH_FP32) hipLaunchKernelGGL(color_bwd_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_col_ptrs(packed), colors,
                                                d_colors, npts, cact, czbar, featbar, d_normals, tpart, nullptr, 1, nullptr, nullptr);
    else hipLaunchKernelGGL(color_bwd_s_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_col16_ptrs(packed), colors,
                            d_colors, npts, cact, czbar, featbar, d_normals, tpart, nullptr, 1, nullptr, nullptr);
    return ok();
}
int launch_color_bwd_rays(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                          int64_t npts, const float* cact, float* czbar, float* featbar, float* d_normals, float* tpart,
                          float* d_pts, float* d_dirs_pts, float* absmax, int grid, int arith, hipStream_t st) {
    if (arith == ARITH_F16) return launch_color_bwd_h(packed, colors, d_colors, dirs, n_per_ray, npts, cact, czbar, featbar, d_normals, tpart, d_pts,
                                                     d_dirs_pts, reinterpret_cast<unsigned*>(absmax), reinterpret_cast<unsigned*>(absmax) + ABSMAX_FLOATS, grid, st);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(color_bwd_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_col_ptrs(packed), colors,
                                                d_colors, npts, cact, czbar, featbar, d_normals, tpart, dirs, n_per_ray, d_pts, d_dirs_pts);
    else hipLaunchKernelGGL(color_bwd_s_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_col16_ptrs(packed), colors,
                            d_colors, npts, cact, czbar, featbar, d_normals, tpart, dirs, n_per_ray, d_pts, d_dirs_pts);
    return ok();
}
int launch_sdf_tangent(const float* packed, const float* pts, const float* d_normals, int64_t npts, const float* act,
                       const float* asave, float* t0aux, float* tsave, float* rsave, float* tpart, float* absmax, int grid, int arith,
                       hipStream_t st) {
    if (arith == ARITH_F16) return launch_sdf_tangent_h(packed, pts, d_normals, npts, act, asave, t0aux, tsave, rsave, tpart,
                                                       reinterpret_cast<unsigned*>(absmax), reinterpret_cast<unsigned*>(absmax) + ABSMAX_FLOATS, grid, st);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(sdf_tangent_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf_ptrs(packed), pts,
                                                d_normals, npts, act, asave, t0aux, tsave, rsave, tpart);
    else hipLaunchKernelGGL(sdf_tangent_s_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf16_ptrs(packed), pts, d_normals,
                            npts, act, asave, t0aux, tsave, rsave, tpart);
    return ok();
}
int launch_sdf_bwd(const float* packed, const float* d_sdf, int64_t npts, const float* act, const float* rsave,
                   const float* featbar, float* zbar, float* tpart, float* absmax, int grid, int arith, hipStream_t st) {
    if (arith == ARITH_F16) return launch_sdf_bwd_h(packed, d_sdf, nullptr, nullptr, npts, act, rsave, featbar, nullptr, zbar, tpart, nullptr,
                                                   reinterpret_cast<unsigned*>(absmax), reinterpret_cast<unsigned*>(absmax) + ABSMAX_FLOATS, grid, st);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(sdf_bwd_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf_ptrs(packed), d_sdf, npts,
                                                act, rsave, featbar, zbar, tpart, nullptr, nullptr, nullptr, nullptr);
    else hipLaunchKernelGGL(sdf_bwd_s_kernel<false>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf16_ptrs(packed), d_sdf, npts,
                            act, rsave, featbar, zbar, tpart, nullptr, nullptr, nullptr, nullptr);
    return ok();
}
int launch_sdf_bwd_rays(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                        const float* act, const float* rsave, const float* featbar, const float* gesave, float* zbar,
                        float* tpart, float* d_pts, float* absmax, int grid, int arith, hipStream_t st) {
    if (arith == ARITH_F16) return launch_sdf_bwd_h(packed, d_sdf, pts, d_normals, npts, act, rsave, featbar, gesave, zbar, tpart, d_pts,
                                                   reinterpret_cast<unsigned*>(absmax), reinterpret_cast<unsigned*>(absmax) + ABSMAX_FLOATS, grid, st);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(sdf_bwd_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf_ptrs(packed), d_sdf, npts,
                                                act, rsave, featbar, zbar, tpart, pts, d_normals, gesave, d_pts);
    else hipLaunchKernelGGL(sdf_bwd_s_kernel<true>, dim3(grid_for(npts, grid)), dim3(256), 0, st, make_sdf16_ptrs(packed), d_sdf, npts, act,
                            rsave, featbar, zbar, tpart, pts, d_normals, gesave, d_pts);
    return ok();
}

}  // namespace dh
