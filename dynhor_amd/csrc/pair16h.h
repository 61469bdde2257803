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
//     wave maximum, its tile-maximum barrier and the scale derivation are micro-steps of that stream too (PEpi below), the image
//     barrier ends the phase;
//   * per 128 points a layer costs 256 KB of L2 -> CU weight traffic (was 512), 4 LDS fragment reads per 12 MFMAs and wave (was 4 +
//     4 weight loads), 4 barriers (was 8).
// The arithmetic is tile16h.h's, value for value: same scales, same piece planes, same MFMA order inside a k-chunk, same epilogue
// expressions -- a pair kernel writes bit-identical tiles to its kernels_mlp_h.hip twin (tests/test_gpu_pair_chains.py).
#pragma once
#include "tile16h.h"

#ifdef DH_STAMPS                 // diagnostic build (scripts/stamps.sh): s_memtime stamps of the pair kernels' phases and of one phase's k-chunks
namespace dh {
constexpr int PSTAMP_SLOTS = 96;
static __device__ unsigned long long dh_pstamps[256 * 4 * PSTAMP_SLOTS];
}
#define PSTAMP(on, slot)                                                                                               \
    do {                                                                                                               \
        if ((on) && (threadIdx.x & 63) == 0 && blockIdx.x < 256)                                                        \
            dh::dh_pstamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * dh::PSTAMP_SLOTS + (slot)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define PSTAMP(on, slot) do { } while (0)
#endif

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
    static constexpr int stamp = -1;
    template <int S> __device__ __forceinline__ void step() {}
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
        PSTAMP(e.stamp >= 0, e.stamp + 2 * KC);
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
// reload_on = false: the reload's loads go through a 0-record descriptor -- no memory traffic, zeros into registers nobody reads (the
// pair's last layer: one code form of phase 2 for every layer, see chain_pair.hip)
template <bool RELOAD, int NREG, class E>
__device__ __forceinline__ void pair_phase(f32x16 (&acc)[MT][2], const _Float16* img, PW<NREG>& W, const u32x4* wcur, const u32x4* wnext,
                                           int wave, int lane, E& e, bool reload_on = true) {
    const _Float16* xrow = img + (lane & 31) * LDH + 8 * (lane >> 5);
    const rsrc_t rc = weight_rsrc(wcur);
    const int woff = pw_lane_off(wave, lane);
    H2 a0[MT], a1[MT];
    pair_loada(a0, xrow, 0);
    if constexpr (NREG < 1) pw_load_chunk<0>(W.wb[0], rc, woff);
    if constexpr (NREG < 2) pw_load_chunk<1>(W.wb[1], rc, woff);
    const rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(wnext), 0, reload_on ? 0x40000000 : 0, 0x00020000);
    pair_chunks<0, RELOAD>(acc, xrow, W, rc, rn, woff, a0, a1, e);
}
// the same micro-steps with no GEMM above them (a pair's last epilogue)
template <int S, class E>
__device__ __forceinline__ void pair_epi_alone_from(E& e) {
    if constexpr (S < P_SLOTS) {
        e.template step<S>();
        pair_epi_alone_from<S + 1>(e);
    }
}

// ---------------------------------------------------------------- an epilogue + hand-off as 192 micro-steps
// The slot program of a tile's epilogue (CRTP: D supplies the elementwise part and what happens to the tile's maximum):
//   slots [0, 16 SPG)            : D::elem<G, SUB>() -- 16 groups G = (m, t, r4) of four accumulator values, SPG slots each; leaves the
//                                  tile's final values in acc and updates the running maxima m0 / m1;
//   the next 12 slots (MIDSEQ)   : 0-5 the six DPP steps of the wave maximum, 6 readlane + D::publish(wave max) (-> sred[wave]),
//                                  7 the tile-maximum barrier, 8 the four maxima read back, 9 D::on_tile_max(max) + the tile scale;
//   then 32 value pairs x SP2    : scale, convert, residuals, convert (acc_to_lds_split's expressions, the pair = rows R, R+1 of this
//                                  lane's column packed into one dword per plane) | lanes 2i / 2i+1 (columns n, n+1) trade halves --
//                                  one DPP quad swap + one v_perm_b32 per plane -- so that the even lane holds row R's two columns and
//                                  the odd lane row R+1's | each writes ONE 32-bit word per plane: 2 ds_write_b32 instead of 4
//                                  ds_write_b16, the same halves in the same places.
// One wave per SIMD hides about five single-issue instructions per MFMA gap (MI355X_MICROARCH.md "one wave per SIMD"): SPG / SP2 spread
// an epilogue's instructions accordingly.  (Stamps of the first version -- 4 ds_write_b16 in every third slot, the maximum and its
// barrier between two chunks -- read 600 cycles per hand-off chunk against 384 of MFMA time and ~1,000 exposed at the barrier:
// profiles/r06_pair_stamps_color_fwd_v1.json.)
constexpr int P_MIDSEQ = 12;
template <class D, int SPG_, int SP2_, bool HANDOFF = true>
struct PEpi {
    static constexpr int SPG = SPG_, SP2 = SP2_, P1 = 16 * SPG_;
    static_assert(P1 + P_MIDSEQ + (HANDOFF ? 32 * SP2_ : 0) <= P_SLOTS, "the epilogue's slot program must fit a phase");
    static_assert(SP2_ == 3 || SP2_ == 4, "");
    float S = 0.f, inv_S = 0.f, m0 = 0.f, m1 = 0.f, wm = 0.f;
    unsigned xsel;                 // v_perm_b32 selector of this lane's parity
    _Float16* wbase;               // this lane's word of image row 4 (lane >> 5) [+ 1 for odd lanes], columns (n & ~1, n | 1) of n-tile 0
    unsigned hl[2], wd[2];
    f32x2 xs;
    int stamp = -1;                // (-DDH_STAMPS: first slot of this phase's k-chunk stamps; -1: none)
    __device__ __forceinline__ D& d() { return *static_cast<D*>(this); }
    __device__ __forceinline__ void handoff_init(_Float16* img, int wave, int lane) {
        xsel = (lane & 1) ? 0x03020706u : 0x05040100u;
        wbase = img + (4 * (lane >> 5) + (lane & 1)) * LDH + (acc_col(wave, 0, lane) & ~1);
    }
    template <int K>
    __device__ __forceinline__ void mid_stage() {
        if constexpr (K == 0) wm = wave_max_step<0x111, 0xf>(fmaxf(m0, m1));
        else if constexpr (K == 1) wm = wave_max_step<0x112, 0xf>(wm);
        else if constexpr (K == 2) wm = wave_max_step<0x114, 0xf>(wm);
        else if constexpr (K == 3) wm = wave_max_step<0x118, 0xf>(wm);
        else if constexpr (K == 4) wm = wave_max_step<0x142, 0xa>(wm);
        else if constexpr (K == 5) wm = wave_max_step<0x143, 0xc>(wm);
        else if constexpr (K == 6) { wm = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wm), 63)); d().publish(wm); }
        else if constexpr (K == 7) { if constexpr (HANDOFF) pair_barrier(); }
        else if constexpr (K == 8) { if constexpr (HANDOFF) wm = d().read_tile_max(); }
        else if constexpr (K == 9) {
            if constexpr (HANDOFF) {
                d().on_tile_max(wm);
                const TileScale ts = scale_for_max(wm);
                S = ts.S;
                inv_S = ts.inv;
            }
        }
    }
    // the 13 operations of pair J = (m, t, r = 2 (J % 8)) over SP2 slots
    template <int J, int OP>
    __device__ __forceinline__ void split_op(const f32x16 (&acc)[MT][2]) {
        constexpr int m = J / 16, t = (J / 8) % 2, r = 2 * (J % 8);
        if constexpr (OP == 0) { xs[0] = acc[m][t][r] * S; xs[1] = acc[m][t][r + 1] * S; hl[0] = pack_f16x2(xs); }
        else if constexpr (OP == 1) hl[1] = pack_f16x2(resid_f16x2(xs, hl[0]));
        else if constexpr (OP == 2) {
            DH_UNROLL for (int p = 0; p < 2; ++p) {
                const unsigned nb = (unsigned)__builtin_amdgcn_update_dpp(0, (int)hl[p], 0xB1, 0xf, 0xf, false);      // quad_perm [1,0,3,2]
                wd[p] = __builtin_amdgcn_perm(nb, hl[p], xsel);
            }
        } else {
            _Float16* q = wbase + (m * 32 + (r & 3) + 8 * (r >> 2)) * LDH + 32 * t;
            *reinterpret_cast<unsigned*>(q) = wd[0];
            *reinterpret_cast<unsigned*>(q + PLANE_H) = wd[1];
        }
    }
    template <int Q>
    __device__ __forceinline__ void split_slot(const f32x16 (&acc)[MT][2]) {
        constexpr int J = Q / SP2, sub = Q % SP2;
        if constexpr (J < 32) {
            if constexpr (SP2 == 4) split_op<J, sub>(acc);
            else {                                   // three slots: convert | residual + exchange | write
                if constexpr (sub == 0) split_op<J, 0>(acc);
                else if constexpr (sub == 1) { split_op<J, 1>(acc); split_op<J, 2>(acc); }
                else split_op<J, 3>(acc);
            }
        }
    }
    template <int SL>
    __device__ __forceinline__ void step() {
        if constexpr (SL < P1) d().template elem<SL / SPG, SL % SPG>();
        else if constexpr (SL < P1 + P_MIDSEQ) mid_stage<SL - P1>();
        else if constexpr (HANDOFF) split_slot<SL - P1 - P_MIDSEQ>(d().acc);
    }
};

}  // namespace dh
