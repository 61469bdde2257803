// Tile-PAIR chain core (gfx950; round 6): the tile-resident chains of kernels_mlp_h.hip re-cut so that a layer's weight pieces enter
// the CU ONCE per 128 points and the image hand-off of one 64-point tile runs under the MFMAs of the other.
//
// What round 5 measured (DESIGN.md section 3 "What the phase stamps say"): with two 64-point workgroups per CU every workgroup pulls a
// layer's 256 KB of weight pieces from L2 for every tile, the CU's L2 path delivers 20-30 B/clk, and the GEMM phase takes 2.4 x its
// matrix time.  Halving the bytes per point (128-point workgroups) made the GEMM matrix-bound, but with ONE workgroup per CU nothing
// ran under the hand-off any more.  Sharing a weight stream between two tiles needs them IN phase, overlapping a hand-off needs them
// OUT of phase -- unless the weights stay where both phases find them.  This core keeps them in REGISTERS:
//
//   * one workgroup of 4 waves per CU, one wave per SIMD, up to 512 registers per lane; the workgroup owns a PAIR of 64-point tiles
//     A, B with one piece-plane LDS image each (tile16h.h: 2 x 67,584 B);
//   * wave w owns output columns [64 w, 64 w + 64) of BOTH tiles and holds its slice of the layer's weights -- 16 k-chunks x 2 n-tiles
//     x 2 pieces = 64 x 16 B per lane = 256 registers (PW) -- for the two phases of the layer:
//       phase 1 (layer l):  G_A(l): accA += imgA x W_l     under it: epilogue of B's layer l-1 and B's hand-off into imgB
//       phase 2 (layer l):  G_B(l): accB += imgB x W_l     under it: epilogue of A's layer l and A's hand-off into imgA, and, chunk by
//                                                           chunk behind the chunk's last MFMA, the reload of W with layer l+1;
//   * a phase is 16 k-chunks x 12 MFMAs = 192 MFMA slots; the epilogue + hand-off of the other tile is written as 192 micro-steps
//     (a few vector / LDS / memory instructions each) and dealt one per slot behind a scheduling barrier -- the way chain_t.hip deals
//     its epilogue -- so nothing depends on how the compiler would interleave two independent instruction streams; the hand-off's
//     tile-maximum barrier sits in front of chunk E::MID (`mid`: the epilogue type says where its elementwise part ends), the image
//     barrier at the end of the phase;
//   * per 128 points a layer costs 256 KB of L2 -> CU weight traffic (was 512), 4 LDS fragment reads per 12 MFMAs and wave (was 4 +
//     4 weight loads), 4 barriers (was 8).
// The arithmetic is tile16h.h's, value for value: same scales, same piece planes, same MFMA order inside a k-chunk, same epilogue
// expressions -- a pair kernel writes bit-identical tiles to its kernels_mlp_h.hip twin (tests/test_gpu_pair_chains.py).
#pragma once
#include "tile16h.h"

namespace dh {

constexpr int P_NKC = 16;                          // k-chunks of a 256-deep layer
constexpr int PAIR_LDS_BYTES = 2 * IMG_H * 2 + 2 * TM * LDA * 4 + 512;      // two piece-plane images, two fp32 aux images, scratch
static_assert(PAIR_LDS_BYTES <= 160 * 1024, "a pair workgroup must fit the CU's LDS");
constexpr int P_SLOTS = P_NKC * 12;                // MFMA slots of a phase
// A wave's 64-column slice of one linear: k-chunks 0 .. NREG-1 stay in registers across the layer's two phases ([k-chunk][n-tile] x 2
// pieces: 16 registers per chunk); chunks NREG .. 15 are streamed from L2 in each phase through three rotating buffers, two chunks
// ahead.  NREG = 16: the whole slice (256 registers), the L2 -> CU weight bytes of a layer halved against the tile form; NREG = 8: 128
// registers, three quarters.  What a kernel can afford is decided by the ARCHITECTURAL half of the register file: the epilogue's tile,
// the A fragments and every vector-ALU operand must sit in v0..v255, only MFMA operands and load targets can live in a0..a255.
template <int NREG>
struct PW { H2 w[NREG][2]; H2 wb[3][2]; };

__device__ __forceinline__ int pw_lane_off(int wave, int lane) { return ((2 * wave) * 2 * 64 + lane) * 16; }
// chunk KC of the slice: four 1-KB loads (PACKH layout: f16x8 index ((kc * 8 + nt) * 2 + piece) * 64 + lane)
template <int KC>
__device__ __forceinline__ void pw_load_chunk(H2 (&dst)[2], rsrc_t wr, int woff) {
    DH_UNROLL for (int t = 0; t < 2; ++t)
        DH_UNROLL for (int p = 0; p < 2; ++p)
            dst[t].p[p] = __builtin_amdgcn_raw_buffer_load_b128(wr, woff + (t * 2 + p) * 1024, KC * (8 * 2 * 64 * 16), 0);
}
template <int KC, int NREG>
__device__ __forceinline__ void pw_load_from(PW<NREG>& W, rsrc_t wr, int woff) {
    if constexpr (KC < NREG) { pw_load_chunk<KC>(W.w[KC], wr, woff); pw_load_from<KC + 1>(W, wr, woff); }
}
template <int NREG>
__device__ __forceinline__ void pw_load_all(PW<NREG>& W, const u32x4* wp, int wave, int lane) {
    pw_load_from<0>(W, weight_rsrc(wp), pw_lane_off(wave, lane));
}

// a descriptor whose stores the hardware drops (0 records): an odd last tile's replayed partner writes nothing, without a branch in the
// dealt instruction stream
__device__ __forceinline__ rsrc_t tile_rsrc_if(const float* tile, bool on) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tile), 0, on ? TILE_F * 4 : 0, 0x00020000);
}

// barrier of the hand-offs inside a phase: the LDS operations of this wave have landed, nothing is said about vector memory (the
// weight reload and the tile stores stay in flight: __syncthreads() would drain them)
__device__ __forceinline__ void pair_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// an epilogue with nothing to do (phase 1 of a pair's first layer)
struct PNoEpi {
    static constexpr int MID = P_NKC / 2;
    template <int S> __device__ __forceinline__ void step() {}
    __device__ __forceinline__ void mid() {}
};

template <int KC, int I, class E>
__device__ __forceinline__ void pair_slots(f32x16 (&acc)[MT][2], const H2 (&a)[MT], const H2 (&b)[2], E& e) {
    if constexpr (I < 12) {
        mfma_step_h<I>(acc, a, b);
        __builtin_amdgcn_sched_barrier(0);
        e.template step<KC * 12 + I>();
        __builtin_amdgcn_sched_barrier(0);
        pair_slots<KC, I + 1>(acc, a, b, e);
    }
}
__device__ __forceinline__ void pair_loada(H2 (&a)[MT], const _Float16* xrow, int kc) {
    DH_UNROLL for (int m = 0; m < MT; ++m)
        DH_UNROLL for (int p = 0; p < 2; ++p)
            a[m].p[p] = *reinterpret_cast<const u32x4*>(xrow + p * PLANE_H + m * 32 * LDH + kc * 16);
}
template <int KC, bool RELOAD, int NREG, class E>
__device__ __forceinline__ void pair_chunks(f32x16 (&acc)[MT][2], const _Float16* xrow, PW<NREG>& W, rsrc_t wcur, rsrc_t wnext, int woff,
                                            H2 (&a0)[MT], H2 (&a1)[MT], E& e) {
    if constexpr (KC < P_NKC) {
        if constexpr (KC == E::MID) { e.mid(); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (KC + 1 < P_NKC) pair_loada((KC & 1) ? a0 : a1, xrow, KC + 1);
        if constexpr (KC + 2 >= NREG && KC + 2 < P_NKC) pw_load_chunk<KC + 2>(W.wb[(KC + 2) % 3], wcur, woff);      // a streamed chunk, two ahead
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KC < NREG) pair_slots<KC, 0>(acc, (KC & 1) ? a1 : a0, W.w[KC], e);
        else pair_slots<KC, 0>(acc, (KC & 1) ? a1 : a0, W.wb[KC % 3], e);
        if constexpr (RELOAD && KC < NREG) { pw_load_chunk<KC>(W.w[KC], wnext, woff); __builtin_amdgcn_sched_barrier(0); }
        pair_chunks<KC + 1, RELOAD>(acc, xrow, W, wcur, wnext, woff, a0, a1, e);
    }
}
// One phase.  acc += img[TM x 256] x (the linear at wcur, its first NREG chunks from W); e's 192 micro-steps dealt under the MFMAs;
// RELOAD: W takes the linear at `wnext` chunk by chunk.  The caller places the phase-ending pair_barrier().
template <bool RELOAD, int NREG, class E>
__device__ __forceinline__ void pair_phase(f32x16 (&acc)[MT][2], const _Float16* img, PW<NREG>& W, const u32x4* wcur, const u32x4* wnext,
                                           int wave, int lane, E& e) {
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const rsrc_t rc = weight_rsrc(wcur);
    const int woff = pw_lane_off(wave, lane);
    H2 a0[MT], a1[MT];
    pair_loada(a0, xrow, 0);
    if constexpr (NREG < 1) pw_load_chunk<0>(W.wb[0], rc, woff);
    if constexpr (NREG < 2) pw_load_chunk<1>(W.wb[1], rc, woff);
    pair_chunks<0, RELOAD>(acc, xrow, W, rc, weight_rsrc(wnext), woff, a0, a1, e);
}
// the same micro-steps with no GEMM above them (a pair's last epilogue)
template <int S, class E>
__device__ __forceinline__ void pair_epi_alone_from(E& e) {
    if constexpr (S < P_SLOTS) {
        if constexpr (S == E::MID * 12) e.mid();
        e.template step<S>();
        pair_epi_alone_from<S + 1>(e);
    }
}

// ---------------------------------------------------------------- the hand-off as micro-steps
// Part 1 (slots 0 .. 95: the caller's elementwise epilogue, 16 groups (m, t, r4) of four values, 6 slots each) leaves the tile's
// final values in acc and the wave's running maxima in (m0, m1); mid(): wave maximum -> sred[wave] -> barrier -> tile scale;
// part 2 (slots 96 .. 191): 32 value pairs x 3 slots: scale + convert | residuals + convert | four 16-bit LDS writes
// (acc_to_lds_split's expressions, pair by pair).
struct PHandoff {
    float S;
    template <int J, int SUB>                        // pair J = 0 .. 31 of the wave's 64 values: (m, t, r = 2 (J % 8)), sub-step 0 .. 2
    __device__ __forceinline__ void split_step(const f32x16 (&acc)[MT][2], _Float16* img, int wave, int lane, unsigned (&hl)[2]) {
        constexpr int m = J / 16, t = (J / 8) % 2, r = 2 * (J % 8);
        if constexpr (SUB == 0) {
            f32x2 x;
            x[0] = acc[m][t][r] * S;
            x[1] = acc[m][t][r + 1] * S;
            hl[0] = pack_f16x2(x);
            // (the residual needs x again: recomputed in sub-step 1 from the same product -- bit-identical, one multiply pair more,
            // two registers less across the MFMA in between)
        } else if constexpr (SUB == 1) {
            f32x2 x;
            x[0] = acc[m][t][r] * S;
            x[1] = acc[m][t][r + 1] * S;
            hl[1] = pack_f16x2(resid_f16x2(x, hl[0]));
        } else {
            _Float16* base = img + (m * 32 + 4 * (lane >> 5)) * LDH + acc_col(wave, t, lane);
            _Float16* q = base + ((r & 3) + 8 * (r >> 2)) * LDH;
            const f16x2v hv = __builtin_bit_cast(f16x2v, hl[0]), lv = __builtin_bit_cast(f16x2v, hl[1]);
            q[0] = hv[0]; q[LDH] = hv[1];
            q[PLANE_H] = lv[0]; q[PLANE_H + LDH] = lv[1];
        }
    }
};

}  // namespace dh
