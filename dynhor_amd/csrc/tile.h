// Device primitives for the tile-resident MLP chains (gfx950 / CDNA4 only).
//
// One workgroup = 256 threads = 4 waves owns a tile of TM points (layout.h: 64 -> two workgroups per CU so one's
// epilogue overlaps the other's MFMAs; 128 -> one).  The tile's current
// activation lives in LDS as a row-major [128 x 256] fp32 image (row stride LDX=260 floats -> ds_read_b128 of
// 32 consecutive rows is bank-conflict free) plus a [128 x 40] "aux" image (embedding / colour extras).
// Every layer is  OUT[128 x 256] = X[128 x K] * M[K x 256]  on v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain
// numerics):  wave w owns output columns [64w, 64w+64) = 2 n-tiles x 4 m-tiles = 8 accumulators (128 VGPRs).
//   A operand (X) : ds_read_b128 from LDS  -> lane (i=l&31, h=l>>5) gets X[m*32+i][kg*8+4h+s], s=0..3
//   B operand (M) : global_load_dwordx4 from the packed, L2-resident weight image (layout.h)
// MFMA step s of k-group kg therefore contracts k = kg*8 + s (lanes 0-31) and kg*8 + 4 + s (lanes 32-63).
//
// "Native" HBM layout of a saved [128 x 256] tile == the accumulator layout, so stores/loads are 1 KiB coalesced
// float4 wave-instructions and the dW kernel can consume saved tiles as MFMA operands straight from global:
//   float4 index = (((w*4 + m)*2 + t)*4 + r4)*64 + lane ; element rr  <->  row m*32 + 8*r4 + 4*(lane>>5) + rr,
//                                                                           col 64*w + 32*t + (lane&31).
#pragma once
#include <hip/hip_runtime.h>
#include "layout.h"

namespace dh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DH_UNROLL _Pragma("unroll")

__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT][2]) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
}

// acc[m][t] += X[128 x nkg*8] * Mpacked   (P1).  xs: LDS base (row stride ldx floats); wp: packed, NT=8.
// Two k-groups per trip with ping-pong operand registers (no register-rotation copies -> the next group's
// LDS/L2 loads stay in flight under the current group's 32 MFMAs).
__device__ __forceinline__ void mfma_block(f32x16 (&acc)[MT][2], const f32x4 (&a)[MT], const f32x4 (&b)[2]) {
    DH_UNROLL for (int s = 0; s < 4; ++s)
        DH_UNROLL for (int m = 0; m < MT; ++m)
            DH_UNROLL for (int t = 0; t < 2; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][s], b[t][s], acc[m][t], 0, 0, 0);
}
// First B fragments (k-group 0) of a layer: they depend only on the weights, so a chain issues them BEFORE the previous
// layer's epilogue + LDS hand-off barrier and their L2 latency is off the critical path of the next GEMM's first MFMAs.
struct BFrag { f32x4 b[2]; };
__device__ __forceinline__ BFrag gemm_b_prefetch(const f32x4* __restrict__ wp, const int wave, const int lane) {
    BFrag f;
    const f32x4* wl = wp + (2 * wave) * 64 + lane;
    DH_UNROLL for (int t = 0; t < 2; ++t) f.b[t] = wl[t * 64];
    return f;
}
__device__ __forceinline__ void gemm_rows(f32x16 (&acc)[MT][2], const float* xs, const int ldx, const int nkg,
                                          const f32x4* __restrict__ wp, const int wave, const int lane, const BFrag& pre) {
    const float* xrow = xs + (lane & 31) * ldx + 4 * (lane >> 5);
    const f32x4* wl = wp + (2 * wave) * 64 + lane;
    f32x4 a0[MT], b0[2], a1[MT], b1[2];
    DH_UNROLL for (int t = 0; t < 2; ++t) b0[t] = pre.b[t];
    DH_UNROLL for (int m = 0; m < MT; ++m) a0[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx);
    // sched_barrier(0) pins "issue next operands, THEN the 8*MT MFMAs of the current ones": left alone, hipcc sinks
    // the loads to just ahead of their first use and every k-group eats the L2 latency.
    _Pragma("unroll 1") for (int kg = 0; kg < nkg; kg += 2) {
        const int k1 = (kg + 1 < nkg) ? kg + 1 : kg;
        DH_UNROLL for (int t = 0; t < 2; ++t) b1[t] = wl[(k1 * 8 + t) * 64];
        DH_UNROLL for (int m = 0; m < MT; ++m) a1[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + k1 * 8);
        __builtin_amdgcn_sched_barrier(0);
        mfma_block(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        if (kg + 1 < nkg) {
            const int k2 = (kg + 2 < nkg) ? kg + 2 : kg + 1;
            DH_UNROLL for (int t = 0; t < 2; ++t) b0[t] = wl[(k2 * 8 + t) * 64];
            DH_UNROLL for (int m = 0; m < MT; ++m) a0[m] = *reinterpret_cast<const f32x4*>(xrow + m * 32 * ldx + k2 * 8);
            __builtin_amdgcn_sched_barrier(0);
            mfma_block(acc, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
__device__ __forceinline__ void gemm_rows(f32x16 (&acc)[MT][2], const float* xs, const int ldx, const int nkg,
                                          const f32x4* __restrict__ wp, const int wave, const int lane) {
    gemm_rows(acc, xs, ldx, nkg, wp, wave, lane, gemm_b_prefetch(wp, wave, lane));
}

// acc2[.] += X[rows of one m-tile][256] * Mpacked(NT=2)   (P2: 64-wide "aux" output).
// The 4 waves split the MT m-tiles x 2 n-tiles of the [TM x 64] output: WPM = 4/MT waves per m-tile, each owning
// NTW = 2/WPM n-tiles (TM=128: wave = m-tile, both n-tiles; TM=64: wave>>1 = m-tile, wave&1 = n-tile).
constexpr int AUX_WPM = 4 / MT;
constexpr int AUX_NTW = 2 / AUX_WPM;
__device__ __forceinline__ int aux_mtile(int wave) { return wave / AUX_WPM; }
__device__ __forceinline__ int aux_ntile(int wave, int tt) { return (wave % AUX_WPM) * AUX_NTW + tt; }
__device__ __forceinline__ void gemm_auxout(f32x16 (&acc2)[AUX_NTW], const float* xs, const int nkg,
                                            const f32x4* __restrict__ wp, const int wave, const int lane) {
    const float* xrow = xs + (32 * aux_mtile(wave) + (lane & 31)) * LDX + 4 * (lane >> 5);
    const f32x4* wl = wp + aux_ntile(wave, 0) * 64 + lane;
    f32x4 a = *reinterpret_cast<const f32x4*>(xrow);
    f32x4 b[AUX_NTW];
    DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) b[t] = wl[t * 64];
    _Pragma("unroll 2") for (int kg = 0; kg < nkg; ++kg) {
        const int kn = (kg + 1 < nkg) ? kg + 1 : kg;
        f32x4 an = *reinterpret_cast<const f32x4*>(xrow + kn * 8);
        f32x4 bn[AUX_NTW];
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) bn[t] = wl[(kn * 2 + t) * 64];
        __builtin_amdgcn_sched_barrier(0);
        DH_UNROLL for (int s = 0; s < 4; ++s)
            DH_UNROLL for (int t = 0; t < AUX_NTW; ++t)
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[t][s], acc2[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        a = an;
        DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) b[t] = bn[t];
    }
}
__device__ __forceinline__ void aux_zero(f32x16 (&a2)[AUX_NTW]) {
    DH_UNROLL for (int t = 0; t < AUX_NTW; ++t) DH_UNROLL for (int r = 0; r < 16; ++r) a2[t][r] = 0.f;
}
// row (within the tile) / column of aux accumulator element (tt, r)
__device__ __forceinline__ int aux_row(int wave, int r, int lane) { return 32 * aux_mtile(wave) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int aux_col(int wave, int tt, int lane) { return 32 * aux_ntile(wave, tt) + (lane & 31); }

// Row / column of accumulator element (m,t,r) for this lane.
__device__ __forceinline__ int acc_row(int m, int r, int lane) { return m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int acc_col(int wave, int t, int lane) { return 64 * wave + 32 * t + (lane & 31); }

// accumulators -> LDS main tile (row-major, stride LDX)
__device__ __forceinline__ void acc_to_lds(const f32x16 (&acc)[MT][2], float* xs, int wave, int lane) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t) {
            float* base = xs + (m * 32 + 4 * (lane >> 5)) * LDX + acc_col(wave, t, lane);
            DH_UNROLL for (int r = 0; r < 16; ++r) base[((r & 3) + 8 * (r >> 2)) * LDX] = acc[m][t][r];
        }
}

// native tile <-> accumulators
// Saved tiles are written once and read once or twice much later (the per-step working set is 26 GB): non-temporal loads and
// stores keep them from displacing the weight pieces every workgroup re-reads from L2 (same-box A/B, round 2: the seven MLP
// stages together 11.86 -> 11.55 ms; loads alone 11.65, stores alone 11.65).
#define DH_TILE_LD(ptr) __builtin_nontemporal_load(ptr)
#define DH_TILE_ST(ptr, v) __builtin_nontemporal_store(v, ptr)
__device__ __forceinline__ void acc_store_native(const f32x16 (&acc)[MT][2], float* __restrict__ tile, int wave, int lane) {
    f32x4* p = reinterpret_cast<f32x4*>(tile) + (size_t)wave * MT * 8 * 64 + lane;
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v;
                v[0] = acc[m][t][4 * r4 + 0]; v[1] = acc[m][t][4 * r4 + 1];
                v[2] = acc[m][t][4 * r4 + 2]; v[3] = acc[m][t][4 * r4 + 3];
                DH_TILE_ST(&p[((m * 2 + t) * 4 + r4) * 64], v);
            }
}
__device__ __forceinline__ void acc_load_native(f32x16 (&acc)[MT][2], const float* __restrict__ tile, int wave, int lane) {
    const f32x4* p = reinterpret_cast<const f32x4*>(tile) + (size_t)wave * MT * 8 * 64 + lane;
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int t = 0; t < 2; ++t)
            DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 v = DH_TILE_LD(&p[((m * 2 + t) * 4 + r4) * 64]);
                acc[m][t][4 * r4 + 0] = v[0]; acc[m][t][4 * r4 + 1] = v[1];
                acc[m][t][4 * r4 + 2] = v[2]; acc[m][t][4 * r4 + 3] = v[3];
            }
}

// aux native tile ([TM x 64]): float4 index = ((m*2 + t)*4 + r4)*64 + lane,
// element rr <-> row 32m + 8*r4 + 4*(lane>>5) + rr, col 32t + (lane&31).
__device__ __forceinline__ void aux_store_native(const f32x16 (&a2)[AUX_NTW], float* __restrict__ tile, int wave, int lane) {
    f32x4* p = reinterpret_cast<f32x4*>(tile) + lane;
    DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt)
        DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
            f32x4 v;
            v[0] = a2[tt][4 * r4 + 0]; v[1] = a2[tt][4 * r4 + 1]; v[2] = a2[tt][4 * r4 + 2]; v[3] = a2[tt][4 * r4 + 3];
            p[((aux_mtile(wave) * 2 + aux_ntile(wave, tt)) * 4 + r4) * 64] = v;
        }
}
// LDS aux image (row stride LDA, cols < 40 valid, others 0) -> aux native tile in HBM
__device__ __forceinline__ void aux_lds_to_native(const float* aux, float* __restrict__ tile, int wave, int lane) {
    f32x16 a2[AUX_NTW];
    DH_UNROLL for (int tt = 0; tt < AUX_NTW; ++tt) {
        const int col = aux_col(wave, tt, lane);
        DH_UNROLL for (int r = 0; r < 16; ++r)
            a2[tt][r] = (col < AUXW) ? aux[aux_row(wave, r, lane) * LDA + col] : 0.f;
    }
    aux_store_native(a2, tile, wave, lane);
}

// ---------------------------------------------------------------- activation math
// fp32 MFMA issues at the VALU rate and does not overlap VALU work of either wave on the SIMD (measured:
// scripts/micro/gemm_micro.hip), so every epilogue instruction is paid in full -- keep these minimal.
// softplus(beta=100) = max(z,0) + log1p(exp(-|beta z|))/beta : 1 mul, v_exp, 1 add, v_log, v_max, 1 fma.
// Equals torch's softplus (incl. its threshold=20 branch: the log term is < 2.1e-11 there) to <= 1e-9 absolute.
__device__ __forceinline__ float softplus100(float z) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(z) * (SOFTPLUS_BETA * 1.44269504088896f));
    const float l = __builtin_amdgcn_logf(1.f + e);
    return fmaf(l, 0.69314718055995f / SOFTPLUS_BETA, fmaxf(z, 0.f));
}
// From the saved post-activation h = softplus(z): em = exp(-beta h) = 1 - sigma'(z), s = sigma'(z) = 1 - em
// (absolute error <= 6e-8 on s in [0,1]).
__device__ __forceinline__ void softplus_deriv_from_h(float h, float& s, float& em) {
    em = __builtin_amdgcn_exp2f(h * (-SOFTPLUS_BETA * 1.44269504088896f));
    s = 1.f - em;
}

}  // namespace dh
