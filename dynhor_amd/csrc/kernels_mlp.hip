// Tile-resident MLP chain kernels (SDF no-grad / train forward / input-gradient / colour forward).  Each chain exists in two
// forms: the shipping split-bf16 kernel and `<name>_kernel`, its native-fp32-MFMA twin (tile.h), the second arithmetic the parity
// tests check the first against; dh_set_arithmetic() (include/dynhor_hip.h) selects the set.  The split-bf16 form of the two SDF
// forward chains is the register-resident kernel of chain_t.hip (round 3); the input-gradient and colour chains are
// `<name>_s_kernel` here (A split on fetch from the fp32 LDS image, tile16.h).
#include "tile.h"
#include "kernels.h"
#include "mlp_common.h"
#include "tile16.h"
#include "workspace.h"


namespace dh {

// ------------------------------------------------------------------------------------------------
// K1: SDF forward, no grad, sdf only (hierarchical up-sampling evaluations; SURVEY §8 a5 "no-grad").
// lin8 reduces to its row 0: a 256-long dot per point, done on the VALU.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sdf_nograd_kernel(SdfPtrs P, const float* __restrict__ pts, int64_t npts,
                                                             float* __restrict__ sdf_out) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        embed_tile(pts, tile * TM, npts, saux, tid);
        __syncthreads();
        f32x16 acc[MT][2];
        BFrag pre = gemm_b_prefetch(P.fwd_main[1], wave, lane);
        for (int l = 0; l < 8; ++l) {
            acc_zero(acc);
            if (l > 0) gemm_rows(acc, smain, LDX, l == 4 ? 28 : 32, P.fwd_main[l], wave, lane, pre);
            if (l == 0 || l == 4) gemm_rows(acc, saux, LDA, 5, P.fwd_aux[l], wave, lane);
            pre = gemm_b_prefetch(P.fwd_main[l + 1], wave, lane);      // next layer's first weights, ahead of the epilogue
            const float b0 = P.bias[l][acc_col(wave, 0, lane)], b1 = P.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return softplus100(v + (t ? b1 : b0)); });
            __syncthreads();                 // every wave finished reading smain as the A operand
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        const float s = row_dot256(smain, P.w8row0, tid) + P.b8_0[0];
        const int64_t gp = tile * TM + tid / TPP;
        if (tid % TPP == 0 && gp < npts) sdf_out[gp] = s;
        __syncthreads();                     // smain/saux are rewritten by the next tile
    }
}

// ------------------------------------------------------------------------------------------------
// K2a: SDF forward for training: saves the embedding (aux native) and every layer input act[l] (l=1..8,
// post-softplus, native tiles), writes feat = lin8 rows 1..256 (native) and sdf = lin8 row 0 (VALU dot).
//   act : [8][ntiles][TILE_F]   (act[l-1] <-> input of layer l)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sdf_fwd_train_kernel(SdfPtrs P, const float* __restrict__ pts, int64_t npts,
                                                                float* __restrict__ sdf_out, float* __restrict__ feat,
                                                                float* __restrict__ act, float* __restrict__ eaux) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        embed_tile(pts, tile * TM, npts, saux, tid);
        __syncthreads();
        aux_lds_to_native(saux, eaux + tile * AUXT_F, wave, lane);
        f32x16 acc[MT][2];
        BFrag pre = gemm_b_prefetch(P.fwd_main[1], wave, lane);
        for (int l = 0; l < 8; ++l) {
            acc_zero(acc);
            if (l > 0) gemm_rows(acc, smain, LDX, l == 4 ? 28 : 32, P.fwd_main[l], wave, lane, pre);
            if (l == 0 || l == 4) gemm_rows(acc, saux, LDA, 5, P.fwd_aux[l], wave, lane);
            pre = gemm_b_prefetch(P.fwd_main[l + 1], wave, lane);      // next layer's first weights, ahead of the epilogue
            const float b0 = P.bias[l][acc_col(wave, 0, lane)], b1 = P.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return softplus100(v + (t ? b1 : b0)); });
            acc_store_native(acc, act + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        const float s = row_dot256(smain, P.w8row0, tid) + P.b8_0[0];
        const int64_t gp = tile * TM + tid / TPP;
        if (tid % TPP == 0 && gp < npts) sdf_out[gp] = s;
        acc_zero(acc);
        gemm_rows(acc, smain, LDX, 32, P.fwd_main[8], wave, lane, pre);
        const float b0 = P.bias[8][acc_col(wave, 0, lane)], b1 = P.bias[8][acc_col(wave, 1, lane)];
        acc_map(acc, [&](int, int t, int, float v) { return v + (t ? b1 : b0); });
        acc_store_native(acc, feat + tile * TILE_F, wave, lane);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K2b: n = d sdf / d x  (upstream SDFNetwork.gradient, App. A.2), reverse chain through the saved activations.
//   u_8 = W8[0,:];  a_l = u_{l+1} * sigma'(z_l)  (sigma' from the saved act[l+1]);  u_l = a_l W_l  (l = 7..1)
//   ge = a_0 W_0 + a_4 W_4[:,217:]/sqrt2 ;  n = J_e(x)^T ge.   Saves a_l (l=0..7) for the backward pass.
//   asave : [8][ntiles][TILE_F]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void sdf_grad_kernel(SdfPtrs P, const float* __restrict__ pts, int64_t npts,
                                                           const float* __restrict__ act, float* __restrict__ asave,
                                                           float* __restrict__ normals, int save, float* __restrict__ gesave) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[MT][2];
        f32x16 ge[AUX_NTW];
        aux_zero(ge);
        // a_7 = W8[0,:] * sigma'(z_7)
        {
            const float w0 = P.w8row0[acc_col(wave, 0, lane)], w1 = P.w8row0[acc_col(wave, 1, lane)];
            acc_load_native(acc, act + ((int64_t)7 * ntiles + tile) * TILE_F, wave, lane);
            acc_map(acc, [&](int, int t, int, float h) { float s, em; softplus_deriv_from_h(h, s, em); return (t ? w1 : w0) * s; });
            if (save) acc_store_native(acc, asave + ((int64_t)7 * ntiles + tile) * TILE_F, wave, lane);
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        BFrag pre = gemm_b_prefetch(P.rev_main[7], wave, lane);
        for (int l = 7; l >= 1; --l) {
            acc_zero(acc);
            gemm_rows(acc, smain, LDX, 32, P.rev_main[l], wave, lane, pre);         // u_l = a_l W_l
            if (l > 1) pre = gemm_b_prefetch(P.rev_main[l - 1], wave, lane);
            if (l == 4) gemm_auxout(ge, smain, 32, P.rev_aux[4], wave, lane);       // skip path -> ge
            // a_{l-1} = u_l * sigma'(z_{l-1})   (sigma' from act[l-1] == input of layer l)
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)(l - 1) * ntiles + tile) * TILE_F) + (size_t)wave * MT * 8 * 64 + lane;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = DH_TILE_LD(hp + ((m * 2 + t) * 4 + r4) * 64);
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            acc[m][t][4 * r4 + rr] *= s;
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);     // bound live registers: one m-slab (8 float4) in flight
            }
            if (save) acc_store_native(acc, asave + ((int64_t)(l - 1) * ntiles + tile) * TILE_F, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        gemm_auxout(ge, smain, 32, P.rev_aux[0], wave, lane);                       // ge += a_0 W_0
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
}

// K2b on the split-bf16 core (same saved tiles and outputs as sdf_grad_kernel; the small ge / saux image stays fp32)
__global__ __launch_bounds__(256, 2) void sdf_grad_s_kernel(Sdf16Ptrs P, const float* __restrict__ pts, int64_t npts,
                                                           const float* __restrict__ act, float* __restrict__ asave,
                                                           float* __restrict__ normals, int save, float* __restrict__ gesave) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        f32x16 acc[MT][2];
        f32x16 ge[AUX_NTW];
        aux_zero(ge);
        // a_7 = W8[0,:] * sigma'(z_7)
        {
            const float w0 = P.w8row0[acc_col(wave, 0, lane)], w1 = P.w8row0[acc_col(wave, 1, lane)];
            acc_load_native(acc, act + ((int64_t)7 * ntiles + tile) * TILE_F, wave, lane);
            acc_map(acc, [&](int, int t, int, float h) { float s, em; softplus_deriv_from_h(h, s, em); return (t ? w1 : w0) * s; });
            if (save) acc_store_native(acc, asave + ((int64_t)7 * ntiles + tile) * TILE_F, wave, lane);
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        for (int l = 7; l >= 1; --l) {
            acc_zero(acc);
            gemm_rows_s(acc, smain, LDX, 16, P.rev16[l], wave, lane);                 // u_l = a_l W_l
            if (l == 4) gemm_auxout_s(ge, smain, 16, P.revaux16[4], wave, lane);      // skip path -> ge
            // a_{l-1} = u_l * sigma'(z_{l-1})   (sigma' from act[l-1] == input of layer l)
            const f32x4* hp = reinterpret_cast<const f32x4*>(act + ((int64_t)(l - 1) * ntiles + tile) * TILE_F) + (size_t)wave * MT * 8 * 64 + lane;
            DH_UNROLL for (int m = 0; m < MT; ++m) {
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 h = DH_TILE_LD(hp + ((m * 2 + t) * 4 + r4) * 64);
                        DH_UNROLL for (int rr = 0; rr < 4; ++rr) {
                            float s, em; softplus_deriv_from_h(h[rr], s, em);
                            acc[m][t][4 * r4 + rr] *= s;
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);     // bound live registers: one m-slab (8 float4) in flight
            }
            if (save) acc_store_native(acc, asave + ((int64_t)(l - 1) * ntiles + tile) * TILE_F, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        gemm_auxout_s(ge, smain, 16, P.revaux16[0], wave, lane);                     // ge += a_0 W_0
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
}



// ------------------------------------------------------------------------------------------------
// K2c: RenderingNetwork forward (mode idr, App. A.3).  input = [p(3), embed_4(view)(27), n(3) | feat(256)]:
// the 33 extras live in the aux image, feat in the main image.  Saves caux (aux native), the post-ReLU
// activations cact[l] (l=1..4 -> slot l-1) and writes colour = sigmoid(lin4).
//   dirs: [nrays,3], point gp belongs to ray gp / n_per_ray.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void color_fwd_kernel(ColPtrs C, const float* __restrict__ pts, const float* __restrict__ dirs,
                                                            int n_per_ray, const float* __restrict__ normals,
                                                            const float* __restrict__ feat, int64_t npts,
                                                            float* __restrict__ color, float* __restrict__ cact,
                                                            float* __restrict__ caux, int save) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid < TM) {
            const int64_t gp = tile * TM + tid;
            float* row = saux + tid * LDA;
            if (gp < npts) {
                const int64_t ray = gp / n_per_ray;
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = dirs[ray * 3 + c];
                    row[c] = pts[gp * 3 + c];
                    row[3 + c] = d;
                    DH_UNROLL for (int k = 0; k < 4; ++k) {
                        float s, co; sincosf(d * (float)(1 << k), &s, &co);
                        row[6 + 6 * k + c] = s;
                        row[6 + 6 * k + 3 + c] = co;
                    }
                    row[30 + c] = normals[gp * 3 + c];
                }
            } else {
                DH_UNROLL for (int c = 0; c < CAUX; ++c) row[c] = 0.f;
            }
            DH_UNROLL for (int c = CAUX; c < LDA; ++c) row[c] = 0.f;
        }
        f32x16 acc[MT][2];
        acc_load_native(acc, feat + tile * TILE_F, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        if (save) aux_lds_to_native(saux, caux + tile * AUXT_F, wave, lane);
        BFrag pre = gemm_b_prefetch(C.fwd_main[0], wave, lane);
        for (int l = 0; l < 4; ++l) {
            acc_zero(acc);
            gemm_rows(acc, smain, LDX, 32, C.fwd_main[l], wave, lane, pre);
            if (l == 0) gemm_rows(acc, saux, LDA, 5, C.fwd_aux0, wave, lane);
            if (l < 3) pre = gemm_b_prefetch(C.fwd_main[l + 1], wave, lane);
            const float b0 = C.bias[l][acc_col(wave, 0, lane)], b1 = C.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return fmaxf(v + (t ? b1 : b0), 0.f); });
            if (save) acc_store_native(acc, cact + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        const int64_t gp = tile * TM + tid / TPP;
        DH_UNROLL for (int j = 0; j < 3; ++j) {
            const float raw = row_dot256(smain, C.w4 + j * 256, tid) + C.b4[j];
            if (tid % TPP == 0 && gp < npts) color[gp * 3 + j] = 1.f / (1.f + __expf(-raw));
        }
        __syncthreads();
    }
}

// K1 in the split-bf16 arithmetic is chain_t.hip's register-resident kernel (sdf_nograd_t_kernel)

// K2a in the split-bf16 arithmetic is chain_t.hip's register-resident kernel (sdf_fwd_train_t_kernel)

// K2c, split-on-fetch
__global__ __launch_bounds__(256, 2) void color_fwd_s_kernel(Col16Ptrs C, const float* __restrict__ pts, const float* __restrict__ dirs,
                                                            int n_per_ray, const float* __restrict__ normals,
                                                            const float* __restrict__ feat, int64_t npts,
                                                            float* __restrict__ color, float* __restrict__ cact,
                                                            float* __restrict__ caux, int save) {
    __shared__ __attribute__((aligned(16))) float smain[TM * LDX];
    __shared__ __attribute__((aligned(16))) float saux[TM * LDA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t ntiles = (npts + TM - 1) / TM;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (tid < TM) {
            const int64_t gp = tile * TM + tid;
            float* row = saux + tid * LDA;
            if (gp < npts) {
                const int64_t ray = gp / n_per_ray;
                DH_UNROLL for (int c = 0; c < 3; ++c) {
                    const float d = dirs[ray * 3 + c];
                    row[c] = pts[gp * 3 + c];
                    row[3 + c] = d;
                    DH_UNROLL for (int k = 0; k < 4; ++k) {
                        float s, co; sincosf(d * (float)(1 << k), &s, &co);
                        row[6 + 6 * k + c] = s;
                        row[6 + 6 * k + 3 + c] = co;
                    }
                    row[30 + c] = normals[gp * 3 + c];
                }
            } else {
                DH_UNROLL for (int c = 0; c < CAUX; ++c) row[c] = 0.f;
            }
            DH_UNROLL for (int c = CAUX; c < LDA; ++c) row[c] = 0.f;
        }
        f32x16 acc[MT][2];
        acc_load_native(acc, feat + tile * TILE_F, wave, lane);
        acc_to_lds(acc, smain, wave, lane);
        __syncthreads();
        if (save) aux_lds_to_native(saux, caux + tile * AUXT_F, wave, lane);
        for (int l = 0; l < 4; ++l) {
            acc_zero(acc);
            gemm_rows_s(acc, smain, LDX, 16, C.main16[l], wave, lane);
            if (l == 0) gemm_rows_s(acc, saux, LDA, AUX_KC, C.aux16, wave, lane);
            const float b0 = C.bias[l][acc_col(wave, 0, lane)], b1 = C.bias[l][acc_col(wave, 1, lane)];
            acc_map(acc, [&](int, int t, int, float v) { return fmaxf(v + (t ? b1 : b0), 0.f); });
            if (save) acc_store_native(acc, cact + ((int64_t)l * ntiles + tile) * TILE_F, wave, lane);
            __syncthreads();
            acc_to_lds(acc, smain, wave, lane);
            __syncthreads();
        }
        const int64_t gp = tile * TM + tid / TPP;
        DH_UNROLL for (int j = 0; j < 3; ++j) {
            const float raw = row_dot256(smain, C.w4 + j * 256, tid) + C.b4[j];
            if (tid % TPP == 0 && gp < npts) color[gp * 3 + j] = 1.f / (1.f + __expf(-raw));
        }
        __syncthreads();
    }
}

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }
static inline int grid_for(int64_t npts, int grid) {
    const int64_t ntiles = (npts + TM - 1) / TM;
    return (int)(ntiles < grid ? ntiles : grid);
}

int launch_sdf_fwd_train(const float* packed, const float* pts, int64_t npts, float* sdf, float* feat, float* act,
                         float* eaux, float* absmax, int grid, int arith, hipStream_t stream) {
    // workspace.h: the step's class maxima and the arithmetic tag -- cleared by EVERY arithmetic's training forward, so that a SPLIT_F16
    // weight-gradient launch behind another arithmetic's forward finds no tag (and poisons its result) instead of stale scales
    if (absmax) (void)hipMemsetAsync(absmax, 0, ABSMAX_FLOATS * sizeof(float), stream);
    if (arith != ARITH_FP32) return launch_sdf_fwd_train_t(packed, pts, npts, sdf, feat, act, eaux, reinterpret_cast<unsigned*>(absmax),
                                                           arith == ARITH_F16, stream);
    hipLaunchKernelGGL(sdf_fwd_train_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_sdf_ptrs(packed), pts, npts, sdf, feat, act, eaux);
    return ok();
}
int launch_sdf_grad(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals,
                    int save, float* gesave, float* absmax, int grid, int arith, hipStream_t stream) {
    const int form = arith >> 8;                              // include/dynhor_hip.h DH_CHAIN_FORM_*: 0 auto, 1 tile, 2 pair
    arith &= 0xff;
    if (arith == ARITH_F16) {
        if (use_pair_form(form, npts, false)) return launch_sdf_grad_p(packed, pts, npts, act, asave, normals, save, gesave, reinterpret_cast<unsigned*>(absmax), stream);
        return launch_sdf_grad_h(packed, pts, npts, act, asave, normals, save, gesave, reinterpret_cast<unsigned*>(absmax), grid, stream);
    }
    if (arith == ARITH_FP32) hipLaunchKernelGGL(sdf_grad_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_sdf_ptrs(packed), pts, npts, act, asave, normals, save, gesave);
    else hipLaunchKernelGGL(sdf_grad_s_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_sdf16_ptrs(packed), pts, npts, act, asave, normals, save, gesave);
    return ok();
}
int launch_color_fwd(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                     const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, float* absmax, int grid,
                     int arith, hipStream_t stream) {
    const int form = arith >> 8;                              // include/dynhor_hip.h DH_CHAIN_FORM_*: 0 auto, 1 tile, 2 pair
    arith &= 0xff;
    if (arith == ARITH_F16) {
        if (use_pair_form(form, npts, true)) return launch_color_fwd_p(packed, pts, dirs, n_per_ray, normals, feat, npts, color, cact, caux, save,
                                                                 reinterpret_cast<unsigned*>(absmax), stream);
        return launch_color_fwd_h(packed, pts, dirs, n_per_ray, normals, feat, npts, color, cact, caux, save,
                                  reinterpret_cast<unsigned*>(absmax), grid, stream);
    }
    const int g = grid_for(npts, grid);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(color_fwd_kernel, dim3(g), dim3(256), 0, stream, make_col_ptrs(packed), pts, dirs, n_per_ray, normals,
                                                feat, npts, color, cact, caux, save);
    else hipLaunchKernelGGL(color_fwd_s_kernel, dim3(g), dim3(256), 0, stream, make_col16_ptrs(packed), pts, dirs, n_per_ray, normals,
                            feat, npts, color, cact, caux, save);
    return ok();
}

int launch_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, int grid, int arith, hipStream_t stream) {
    if (npts <= 0) return 0;
    if (arith != ARITH_FP32) return launch_sdf_nograd_t(packed, pts, npts, sdf, arith == ARITH_F16, stream);
    hipLaunchKernelGGL(sdf_nograd_kernel, dim3(grid_for(npts, grid)), dim3(256), 0, stream, make_sdf_ptrs(packed), pts, npts, sdf);
    return ok();
}

}  // namespace dh
