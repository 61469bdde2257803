// Weight-gradient GEMMs, cross-tile reduction and weight-norm fold.
//
// Wbar[256 x NB*32] = sum over points of A[p,:]^T B[p,:]  with the POINT axis as the MFMA k dimension.  Both operands
// are saved "native" tiles (tile.h): a float4 at ((..m..t..)*4 + r4)*64 + lane IS four consecutive A (or B)
// fragments of v_mfma_f32_32x32x2_f32 for the k-pairs {row, row+4} (and two consecutive ones the 8 k-values of one
// v_mfma_f32_32x32x16_bf16 operand), so operands leave HBM as 1 KiB coalesced wave loads.  8 waves per workgroup,
// split-K over tiles across gridDim.x persistent workgroups, slabs reduced in slab_reduce_kernel / fold_kernel
// (deterministic: no float atomics).
#include "tile16h.h"
#include "kernels.h"
#include "workspace.h"

namespace dh {

struct DwJob {
    const float* A1; const float* B1;
    const float* A2; const float* B2;      // optional second (A,B) pair accumulated into the same output
    int64_t off;                           // float offset of this job's [256 x nb*32] slab inside a split's slab block
    int nb;                                // 8: B is a main native tile; 2: B is an aux native tile
    // two-piece fp16 kernel, per operand pair: the operands' workspace.h absmax class (launch-wide maximum; -1: the constant
    // H2_XS) and, for the ONE heavy-tailed operand of the pair, its tmax class (per-tile maxima; -1: this operand is not the one)
    int ca[2], cb[2], ha[2], hb[2];
};
struct DwJobs { DwJob j[16]; int n; };
// Job groups (round 3): the persistent workgroups are dealt to DW_GROUPS groups of consecutive jobs, group k getting a share of
// the workgroups proportional to its MFMA cost; a workgroup of group k runs ONLY that group's jobs, over ntiles / (its group's
// size) tiles.  Work per workgroup is unchanged (fewer jobs x more tiles) and still equal across workgroups, but a job now has
// as many split-K slabs as its group has workgroups (~G / DW_GROUPS) instead of G: the slab traffic (0.84 GB written by this
// kernel + read back by slab_reduce_kernel with one group) falls by that factor.  DW_GROUPS = 1 is the round-2 kernel.
#ifndef DW_GROUPS
#define DW_GROUPS 8
#endif
#ifndef DW_AUX_COST
#define DW_AUX_COST 10.0           // (round 5, two-piece aux body, two same-box sweeps: 8: 2.27, 10: 2.19 / 2.22, 12: 2.22 / 2.26, 16: 2.27 ms)
#endif
struct DwGroups { int n; int job0[DW_GROUPS + 1]; int wg0[DW_GROUPS + 1]; int64_t off0[DW_GROUPS + 1]; };
__device__ __forceinline__ int dw_group_of(const DwGroups& Gp, int g) {
    int k = 0;
    while (k + 1 < Gp.n && g >= Gp.wg0[k + 1]) ++k;
    return k;
}

// NB == 8: wave (wo = wave>>1, wn = wave&1) owns output rows [64wo, 64wo+64) x cols [128wn, 128wn+128): 2 x 4 tiles,
//          6 operand float4 per 32 MFMAs.   NB == 2: wave owns rows [32wave, 32wave+32) x 64 cols: 1 x 2 tiles.
template <int NB> struct DwShape;
template <> struct DwShape<8> { static constexpr int NA = 2, NBW = 4; };
template <> struct DwShape<2> { static constexpr int NA = 1, NBW = 2; };

template <int NB>
struct DwOperands { f32x4 a[DwShape<NB>::NA]; f32x4 b[DwShape<NB>::NBW]; };

template <int NB>
__device__ __forceinline__ void dw_mfma(f32x16 (&acc)[DwShape<NB>::NA][DwShape<NB>::NBW], const DwOperands<NB>& o) {
    DH_UNROLL for (int rr = 0; rr < 4; ++rr)
        DH_UNROLL for (int i = 0; i < DwShape<NB>::NA; ++i)
            DH_UNROLL for (int j = 0; j < DwShape<NB>::NBW; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[i][rr], o.b[j][rr], acc[i][j], 0, 0, 0);
}

// ---------------------------------------------------------------- native-fp32-MFMA variant (dh_set_arithmetic(DH_ARITH_FP32_MFMA))
// One persistent workgroup per CU: split g owns tiles [nt*g/G, nt*(g+1)/G) and runs EVERY job over them, so all
// workgroups do identical work (no tail) and write one slab block each.
// A register-streamed kernel would fetch every operand byte 2x (A) / 4x (B) per workgroup and is limited by the
// CU's L1-miss throughput (scripts/micro/dw_micro2.hip: 63 % streamed vs 84 % cache-resident; measured 7.3 vs 5.5 ms,
// removed in round 2).  Here each k-quad's
// 16 KiB (8 A pieces + 8 B pieces of 1 KiB = one wave-wide 16-B LDS-DMA each: the native tile layout is lane-linear,
// exactly what global_load_lds needs) enters the CU once into a DW_STAGES-deep LDS ring; the 8 waves then read their
// 2 A + 4 B fragments with conflict-free ds_read_b128.  Every wave issues exactly 2 DMAs per k-quad, so one counted
// s_waitcnt vmcnt(2*(S-2)) + one raw s_barrier per k-quad orders RAW (all pieces landed) and WAR (stage i-1 fully
// read before it is refilled).  ds_reads are inline asm: hipcc would otherwise drain vmcnt(0) before LDS reads
// while a DMA is in flight.
constexpr int DW_STAGES = 6;
constexpr int DW_STAGE_BYTES = 16384;

__device__ __forceinline__ f32x4 lds_read_b128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

template <int NB>
__device__ __forceinline__ void dw_body_lds(const DwJob& J, int64_t t0, int64_t t1, float* __restrict__ out, int wave, int lane,
                                            char* lds) {
    constexpr int KQ = MT * 4;
    constexpr int BT = (NB == 8) ? TILE_F : AUXT_F;
    constexpr int NA = DwShape<NB>::NA, NBW = DwShape<NB>::NBW;
    constexpr int S = DW_STAGES;
    f32x16 acc[NA][NBW];
    DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int T = (int)(t1 - t0);
    const int npairs = J.A2 ? 2 : 1;
    const int total = npairs * T * KQ;
    (void)total;
    const unsigned lds_base = (unsigned)(uintptr_t)lds;     // LDS byte address of the ring
    // this wave's DMA pieces: A piece ot = wave ; B piece nt = wave (NB == 8) or wave & 1 (NB == 2; slots >= 2 are dummies)
    const int a_ot = wave, b_nt = (NB == 8) ? wave : (wave & 1);

    // issue-side cursor (pair, tile, kq) advanced incrementally: no division / modulo in the loop (the recomputing
    // form cost ~100 SALU instructions per k-quad, 3 per MFMA)
    int is_pair = 0, is_kq = 0, is_stage = 0;
    int64_t is_tile = t0;
    auto issue = [&](int /*u*/) {
        const int m = is_kq >> 2, r4 = is_kq & 3;
        const float* A = is_pair ? J.A2 : J.A1;
        const float* Bm = is_pair ? J.B2 : J.B1;
        const f32x4* ga = reinterpret_cast<const f32x4*>(A + is_tile * TILE_F) + ((((a_ot >> 1) * MT + m) * 2 + (a_ot & 1)) * 4 + r4) * 64 + lane;
        const int bi = (NB == 8) ? ((((b_nt >> 1) * MT + m) * 2 + (b_nt & 1)) * 4 + r4) : ((m * 2 + b_nt) * 4 + r4);
        const f32x4* gb = reinterpret_cast<const f32x4*>(Bm + is_tile * BT) + bi * 64 + lane;
        char* la = lds + is_stage * DW_STAGE_BYTES + wave * 1024;
        char* lb = lds + is_stage * DW_STAGE_BYTES + 8192 + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                         (__attribute__((address_space(3))) void*)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb,
                                         (__attribute__((address_space(3))) void*)lb, 16, 0, 0);
        if (++is_stage == S) is_stage = 0;
        if (++is_kq == KQ) {
            is_kq = 0;
            if (++is_tile == t1) { is_tile = t0; ++is_pair; }
        }
    };

    // Operand registers are double-buffered: unit i+1 is read from LDS (latency hidden) while unit i's MFMAs issue.
    // total is a multiple of KQ (even), so the two register sets alternate statically.
    int rd_stage = 0;
    auto read_ops = [&](DwOperands<NB>& o, int /*u*/) {
        const unsigned sb = lds_base + (unsigned)rd_stage * DW_STAGE_BYTES + (unsigned)lane * 16;
        if (++rd_stage == S) rd_stage = 0;
        DH_UNROLL for (int ii = 0; ii < NA; ++ii) {
            const int ot = (NB == 8) ? ((wave >> 1) * 2 + ii) : wave;
            o.a[ii] = lds_read_b128(sb + ot * 1024);
        }
        DH_UNROLL for (int j = 0; j < NBW; ++j) {
            const int nt = (NB == 8) ? ((wave & 1) * 4 + j) : j;
            o.b[j] = lds_read_b128(sb + 8192 + nt * 1024);
        }
    };
    // wait until unit u's two DMAs (this wave's) have landed: later-issued units may stay in flight
    auto wait_unit = [&](int u, int issued_upto) {          // issued_upto = index of the youngest issued unit
        const int later = issued_upto - u;
        if (later >= S - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (S - 2)) : "memory");
        else if (later == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if (total > 0) {
        int issued = -1;
        for (int u = 0; u < S - 1 && u < total; ++u) { issue(u); issued = u; }
        DwOperands<NB> o0, o1;
        wait_unit(0, issued);
        asm volatile("s_barrier" ::: "memory");
        read_ops(o0, 0);
        for (int i = 0; i < total; i += 2) {
            // ---- even unit i (operands o0); look-ahead read of unit i+1 into o1
            wait_unit(i + 1, issued);
            asm volatile("s_barrier" ::: "memory");
            if (issued + 1 < total) { issue(issued + 1); ++issued; }
            read_ops(o1, i + 1);
            __builtin_amdgcn_sched_barrier(0);
            dw_mfma<NB>(acc, o0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // ---- odd unit i+1 (operands o1); look-ahead read of unit i+2 into o0
            if (i + 2 < total) {
                wait_unit(i + 2, issued);
                asm volatile("s_barrier" ::: "memory");
                if (issued + 1 < total) { issue(issued + 1); ++issued; }
                read_ops(o0, i + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            dw_mfma<NB>(acc, o1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_barrier" ::: "memory");      // ring is reused by the next job
    DH_UNROLL for (int i = 0; i < NA; ++i)
        DH_UNROLL for (int j = 0; j < NBW; ++j) {
            const int ot = (NB == 8) ? ((wave >> 1) * 2 + i) : wave;
            const int nt = (NB == 8) ? ((wave & 1) * 4 + j) : j;
            float* o = out + ((int64_t)ot * NB + nt) * 1024 + lane;
            DH_UNROLL for (int r = 0; r < 16; ++r) o[r * 64] = acc[i][j][r];
        }
}

// ---------------------------------------------------------------- split-bf16 variant (shipping)
// Same native-tile operands and slab layout as the kernels above; only the multiply changes.  Each operand value is split
// into three bf16 pieces x = x1 + x2 + x3 (24 mantissa bits, exact residuals) and the six products a1b1, a1b2, a2b1, a1b3,
// a2b2, a3b1 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the sum reproduces the fp32 product to 2^-24 (the
// dropped terms are <= 2^-24 relative; scripts/micro/bf16_split_accuracy.py: 1.1e-7 vs fp64 against 2.9e-7 for the fp32
// GEMM) while a 16-point k-step costs 6 x 32 cycles instead of 8 x 64 (v_mfma_f32_32x32x2_f32).  Two consecutive k-quads
// of the native tiles give a lane its 8 k-values (4 + 4 points of its kk half); A and B use the same point <-> k-slot
// map and the contraction index is a dummy, so no shuffling is needed.
// Piece staging: every operand tile of a k-pair (8 A tiles + NB B tiles of 16 points) is split ONCE per workgroup -- wave w
// loads A tile w and B tile w (two f32x4 per tile and lane, straight from the native tiles: 1 KiB coalesced), splits them
// and writes the three bf16x8 pieces lane-linear into a double-buffered 48 KiB LDS image; after one barrier every wave
// reads the pieces of its 2 A + 4 B tiles (conflict-free ds_read_b128) and issues its 48 MFMAs.  (Splitting inside each
// consumer wave instead costs 3x the VALU work -- 2 + 4 tiles per wave against 16 unique per workgroup: measured 4.43 ms
// against 4.02 ms for this form and 5.65 ms for the native-fp32 kernel.)
constexpr int DWP_TILE = 3 * 1024;                 // bytes of one tile's three pieces (64 lanes x 16 B each)
constexpr int DWP_BUF = 16 * DWP_TILE;             // one k-pair: A tiles 0..7, B tiles 8..15

// One half of a k-pair for the NB == 8 shape, program order written out (the chain kernels' scheme, tile16.h gemm_rows_s):
// 24 MFMAs product-major over the 2 x 2 accumulators of this half, each issued at raised wave priority, and after every
// second one a third of the split of one pair of the tile this wave publishes for the NEXT k-pair; the three piece writes
// follow the last MFMA.
template <int I, int JB>
__device__ __forceinline__ void dw_half_steps(f32x16 (&acc)[2][4], const Bf3 (&a)[2], const Bf3 (&b)[2], U3 (&pc)[MT],
                                              const RawA& raw, SplitState& st) {
    if constexpr (I < 24) {
        constexpr int pa[6] = {2, 1, 0, 1, 0, 0}, pb[6] = {0, 1, 2, 0, 1, 0};
        constexpr int p = I / 4, ii = (I % 4) / 2, j = I % 2;
        __builtin_amdgcn_s_setprio(1);
        acc[ii][JB + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ii].p[pa[p]], b[j].p[pb[p]], acc[ii][JB + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I % 2 == 1) {
            split_step<I / 2>(pc, raw, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        dw_half_steps<I + 1, JB>(acc, a, b, pc, raw, st);
    }
}

template <int NB>
__device__ __forceinline__ void dw_body_pieces(const DwJob& J, int64_t t0, int64_t t1, float* __restrict__ out, int wave,
                                               int lane, char* lds) {
    constexpr int KQ = MT * 4;
    constexpr int BT = (NB == 8) ? TILE_F : AUXT_F;
    constexpr int NA = DwShape<NB>::NA, NBW = DwShape<NB>::NBW;
    f32x16 acc[NA][NBW];
    DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int T = (int)(t1 - t0);
    const int npairs = J.A2 ? 2 : 1;
    const int NP = npairs * T * (KQ / 2);                 // k-pairs (16 points each)
    const bool has_b = NB == 8 || wave < 2;               // NB == 2: only two B tiles exist
    struct Raw { f32x4 a0, a1, b0, b1; };
    // load-side cursor over (operand pair, tile, k-pair)
    int ld_pair = 0, ld_kp = 0;
    int64_t ld_tile = t0;
    auto load = [&](Raw& r) {
        const float* A = ld_pair ? J.A2 : J.A1;
        const float* Bm = ld_pair ? J.B2 : J.B1;
        const int kq = 2 * ld_kp, m = kq >> 2, r4 = kq & 3;                       // r4 is 0 or 2: the pair is (r4, r4 + 1)
        const f32x4* ga = reinterpret_cast<const f32x4*>(A + ld_tile * TILE_F) + ((((wave >> 1) * MT + m) * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
        // non-temporal loads: every operand byte is read exactly once by exactly one workgroup (same-box A/B, two rounds each:
        // 3.34 vs 3.39 ms)
        r.a0 = __builtin_nontemporal_load(ga); r.a1 = __builtin_nontemporal_load(ga + 64);
        if (has_b) {
            const int bn = (NB == 8) ? wave : (wave & 1);
            const int bi = (NB == 8) ? ((((bn >> 1) * MT + m) * 2 + (bn & 1)) * 4 + r4) : ((m * 2 + bn) * 4 + r4);
            const f32x4* gb = reinterpret_cast<const f32x4*>(Bm + ld_tile * BT) + bi * 64 + lane;
            r.b0 = __builtin_nontemporal_load(gb); r.b1 = __builtin_nontemporal_load(gb + 64);
        }
        if (++ld_kp == KQ / 2) {
            ld_kp = 0;
            if (++ld_tile == t1) { ld_tile = t0; if (++ld_pair == npairs) ld_pair = 0; }    // past the end: wrap (harmless re-read)
        }
    };
    auto publish_a = [&](const Raw& r, int par) {         // split this wave's A tile and write its three pieces
        char* base = lds + par * DWP_BUF + lane * 16;
        const Bf3 pa = split3(r.a0, r.a1);
        DH_UNROLL for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(base + wave * DWP_TILE + p * 1024) = pa.p[p];
    };
    auto publish_b = [&](const Raw& r, int par) {
        char* base = lds + par * DWP_BUF + lane * 16;
        if (has_b) {
            const Bf3 pb = split3(r.b0, r.b1);
            const int bn = (NB == 8) ? wave : (wave & 1);
            DH_UNROLL for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(base + (8 + bn) * DWP_TILE + p * 1024) = pb.p[p];
        }
    };
    auto piece = [&](int par, int tile) {
        Bf3 f;
        const char* base = lds + par * DWP_BUF + tile * DWP_TILE + lane * 16;
        DH_UNROLL for (int p = 0; p < 3; ++p) f.p[p] = *reinterpret_cast<const bf16x8*>(base + p * 1024);
        return f;
    };
    if (NP > 0) {
        Raw r0, r1;
        load(r0);
        load(r1);
        __builtin_amdgcn_sched_barrier(0);
        publish_a(r0, 0);
        publish_b(r0, 0);
        load(r0);                                          // pair 2 in flight
        __syncthreads();
        constexpr int H = NBW / 2;
        // one k-pair; `nxt` holds the raw operands of pair p+1 (published here) and is refilled with pair p+3.  The two
        // register sets alternate STATICALLY (the loop is unrolled by two): selecting the set with a run-time index makes the
        // compiler load into temporaries and wait for them on the spot, which exposes the full HBM latency every pair.
        // Buffer par^1 was last read before the barrier that ended the previous step, so the next pair's split + piece
        // writes may sit ANYWHERE in this step: the A tile's go into one scheduling region with the first half of the MFMAs,
        // the B tile's with the second half, and the compiler interleaves the VALU / ds_write work with the matrix
        // instructions (round 2, same-box A/B: before the MFMAs, pinned 3.55 ms; between the halves 3.58; all with the first
        // half 3.44; this form 3.44; explicit 1 MFMA : 4 VALU sched_group_barrier pipeline 3.48).  UNCONDITIONAL -- a branch
        // would end the scheduling region: after the last pair this writes wrapped-around data into the idle buffer, which
        // the next job's prologue overwrites behind a barrier.
        auto step = [&](int par, Raw& nxt) {
            if constexpr (NB == 8) {
                Bf3 a[2], b[2];
                U3 pc[MT];
                RawA raw;
                SplitState st;
                char* wbase = lds + (par ^ 1) * DWP_BUF + lane * 16;
                DH_UNROLL for (int ii = 0; ii < 2; ++ii) a[ii] = piece(par, (wave >> 1) * 2 + ii);
                DH_UNROLL for (int j = 0; j < 2; ++j) b[j] = piece(par, 8 + (wave & 1) * 4 + j);
                raw.lo[0] = nxt.a0; raw.hi[0] = nxt.a1;
                __builtin_amdgcn_sched_barrier(0);
                dw_half_steps<0, 0>(acc, a, b, pc, raw, st);
                DH_UNROLL for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(wbase + wave * DWP_TILE + p * 1024) = pc[0].p[p];
                DH_UNROLL for (int j = 0; j < 2; ++j) b[j] = piece(par, 8 + (wave & 1) * 4 + 2 + j);
                raw.lo[0] = nxt.b0; raw.hi[0] = nxt.b1;
                __builtin_amdgcn_sched_barrier(0);
                dw_half_steps<0, 2>(acc, a, b, pc, raw, st);
                DH_UNROLL for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(wbase + (8 + wave) * DWP_TILE + p * 1024) = pc[0].p[p];
                __builtin_amdgcn_sched_barrier(0);
                load(nxt);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                return;
            }
            Bf3 a[NA], b[H];
            DH_UNROLL for (int ii = 0; ii < NA; ++ii) a[ii] = piece(par, (NB == 8) ? ((wave >> 1) * 2 + ii) : wave);
            DH_UNROLL for (int j = 0; j < H; ++j) b[j] = piece(par, 8 + ((NB == 8) ? ((wave & 1) * 4 + j) : j));
            __builtin_amdgcn_sched_barrier(0);
            publish_a(nxt, par ^ 1);
            DH_UNROLL for (int ii = 0; ii < NA; ++ii)
                DH_UNROLL for (int j = 0; j < H; ++j) acc[ii][j] = mfma6(a[ii], b[j], acc[ii][j]);
            __builtin_amdgcn_sched_barrier(0);
            if (H < NBW) {
                DH_UNROLL for (int j = 0; j < H; ++j) b[j] = piece(par, 8 + ((NB == 8) ? ((wave & 1) * 4 + H + j) : (H + j)));
            }
            publish_b(nxt, par ^ 1);
            DH_UNROLL for (int ii = 0; ii < NA; ++ii)
                DH_UNROLL for (int j = 0; j < H; ++j) acc[ii][H + j] = mfma6(a[ii], b[j], acc[ii][H + j]);
            __builtin_amdgcn_sched_barrier(0);
            load(nxt);      // pair p+3; UNCONDITIONAL so the compiler can count on it being in flight (vmcnt(4..7) at the next
                            // publish instead of draining everything); past the end the cursor wraps onto valid memory.  A third
                            // register set (pair p+4 in flight) measured the same 3.47 ms: the depth is not what limits it
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        int p = 0;
        for (; p + 1 < NP; p += 2) {
            step(0, r1);
            step(1, r0);
        }
        if (p < NP) step(0, r1);
    }
    DH_UNROLL for (int i = 0; i < NA; ++i)
        DH_UNROLL for (int j = 0; j < NBW; ++j) {
            const int ot = (NB == 8) ? ((wave >> 1) * 2 + i) : wave;
            const int nt = (NB == 8) ? ((wave & 1) * 4 + j) : j;
            float* o = out + ((int64_t)ot * NB + nt) * 1024 + lane;
            DH_UNROLL for (int r = 0; r < 16; ++r) o[r * 64] = acc[i][j][r];
        }
}

// ---------------------------------------------------------------- two-piece fp16 variant (DH_ARITH_SPLIT_F16; tile16h.h)
// dw_body_pieces with two fp16 pieces per value and three products: 24 MFMAs per wave and k-pair instead of 48, a 32 KiB piece
// image per k-pair instead of 48.  Operand scales (powers of two, applied as the raw values are split).  Every operand pair has
// one heavy-tailed operand X (an adjoint or a tangent: a few sample points near the surface carry almost everything) and one
// tame operand Y (activations, embeddings, reverse-chain values).  X is scaled tile by tile from the tile's own maximum (workspace.h
// tmax: scaled maximum in [256, 512)) -- one launch-wide scale, set by one outlier, would push the typical tile into fp16's
// subnormals (measured: 1.7e-4 on the gradient).  Y is scaled by its class scale (launch-wide maximum, workspace.h absmax, or the
// constant H2_XS) TIMES S_X(launch) / S_X(tile) <= 1, so every product carries the same S_X(launch) S_Y; what Y loses below fp16's
// range in a tile whose X is small is bounded by 2^-33 of the dominant tiles' products, well under the fp32 accumulator's own
// rounding.  A job with two operand pairs runs pair 0 over all its tiles first; at the change-over the accumulators are
// multiplied by the ratio of the two pairs' scale products (exact), and the slab is written divided by the last pair's.
constexpr int DWH_TILE = 2 * 1024;
constexpr int DWH_BUF = 16 * DWH_TILE;
// per pair (plain scalars, selected with ?: -- an array indexed by the run-time pair number goes to scratch and turns the tile
// maximum's load into a flat load): prod = S_X(launch) S_Y, also the tame operand's scale before the per-tile division; xt = the
// heavy operand's per-tile maxima; heavy_a: X is A
// poison: 1, or NaN when the workspace's scale tables were not written by this step's SPLIT_F16 stages (workspace.h ABSMAX_TAG)
struct DwScales { float prod0, prod1; const unsigned* xt0; const unsigned* xt1; bool heavy_a0, heavy_a1; float poison; };
__device__ __forceinline__ float dw_class_scale(const unsigned* __restrict__ absmax, int cls) {
    return cls < 0 ? H2_XS : __builtin_bit_cast(float, pow2_scale_bits(absmax[cls * ABSMAX_STRIDE], H2_AT));
}
__device__ __forceinline__ DwScales dw_job_scales(const DwJob& J, const unsigned* __restrict__ absmax, const unsigned* __restrict__ tmax,
                                                  int64_t ntiles) {
    DwScales s;
    auto one = [&](int i, float& prod, const unsigned*& xt, bool& heavy_a) {
        heavy_a = J.ha[i] >= 0;
        const int hx = heavy_a ? J.ha[i] : J.hb[i];
        xt = tmax + (int64_t)(hx < 0 ? 0 : hx) * ntiles;
        prod = dw_class_scale(absmax, J.ca[i]) * dw_class_scale(absmax, J.cb[i]);
    };
    one(0, s.prod0, s.xt0, s.heavy_a0);
    one(1, s.prod1, s.xt1, s.heavy_a1);
    s.poison = absmax[ABSMAX_TAG * ABSMAX_STRIDE] == ABSMAX_TAG_F16 ? 1.f : __builtin_nanf("");
    return s;
}
// development switches of the main body (scripts/build_variant.sh; same-box A/Bs in profiles/r05_ab_dw_variants.json)
#ifdef DW_NO_SETPRIO
#define DW_PRIO(x)
#else
#define DW_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifdef DW_PLAIN_LOADS
#define DW_LOAD(p) (*(p))
#else
#define DW_LOAD(p) __builtin_nontemporal_load(p)
#endif
template <int I, int JB>
__device__ __forceinline__ void dw_half_steps_h(f32x16 (&acc)[2][4], const H2 (&a)[2], const H2 (&b)[2], H2 (&pc)[MT],
                                                const RawA& raw, SplitStateH& st) {
    if constexpr (I < 12) {
        constexpr int pa[3] = {0, 1, 0}, pb[3] = {1, 0, 0};
        constexpr int p = I / 4, ii = (I % 4) / 2, j = I % 2;
        DW_PRIO(1);
        acc[ii][JB + j] = mfma_h(a[ii].p[pa[p]], b[j].p[pb[p]], acc[ii][JB + j]);
        DW_PRIO(0);
        __builtin_amdgcn_sched_barrier(0);
        split_step_h<I, true>(pc, raw, st);
        __builtin_amdgcn_sched_barrier(0);
        dw_half_steps_h<I + 1, JB>(acc, a, b, pc, raw, st);
    }
}
template <int NB>
__device__ __forceinline__ void dw_body_pieces_h(const DwJob& J, const DwScales& sc, int64_t t0, int64_t t1, float* __restrict__ out,
                                                 int wave, int lane, char* lds) {
    static_assert(NB == 8, "main jobs only: the aux jobs run dw_body_pieces<2> (dw_f16x2_kernel)");
    constexpr int KQ = MT * 4;
    constexpr int BT = (NB == 8) ? TILE_F : AUXT_F;
    constexpr int NA = DwShape<NB>::NA, NBW = DwShape<NB>::NBW;
    f32x16 acc[NA][NBW];
    DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int T = (int)(t1 - t0);
    const int npairs = J.A2 ? 2 : 1;
    const int NP1 = T * (KQ / 2);                         // k-pairs of one operand pair
    const int NP = npairs * NP1;
    const bool has_b = NB == 8 || wave < 2;
    struct Raw { f32x4 a0, a1, b0, b1; float sa, sb; };
    int ld_pair = 0, ld_kp = 0;
    int64_t ld_tile = t0;
    auto load = [&](Raw& r) {
        const float* A = ld_pair ? J.A2 : J.A1;
        const float* Bm = ld_pair ? J.B2 : J.B1;
        {   // the tile's scales: X by its own maximum, Y by prod / S_X(tile)   (wave-uniform: scalar load + a few scalar ops)
            const unsigned sxb = pow2_scale_bits((ld_pair ? sc.xt1 : sc.xt0)[ld_tile], H2_AT);
            const float sx = __builtin_bit_cast(float, sxb), sy = (ld_pair ? sc.prod1 : sc.prod0) * __builtin_bit_cast(float, pow2_inv_bits(sxb));
            const bool ha = ld_pair ? sc.heavy_a1 : sc.heavy_a0;
            r.sa = ha ? sx : sy;
            r.sb = ha ? sy : sx;
        }
        const int kq = 2 * ld_kp, m = kq >> 2, r4 = kq & 3;
        const f32x4* ga = reinterpret_cast<const f32x4*>(A + ld_tile * TILE_F) + ((((wave >> 1) * MT + m) * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
        r.a0 = DW_LOAD(ga); r.a1 = DW_LOAD(ga + 64);
        if (has_b) {
            const int bn = (NB == 8) ? wave : (wave & 1);
            const int bi = (NB == 8) ? ((((bn >> 1) * MT + m) * 2 + (bn & 1)) * 4 + r4) : ((m * 2 + bn) * 4 + r4);
            const f32x4* gb = reinterpret_cast<const f32x4*>(Bm + ld_tile * BT) + bi * 64 + lane;
            r.b0 = DW_LOAD(gb); r.b1 = DW_LOAD(gb + 64);
        }
        if (++ld_kp == KQ / 2) {
            ld_kp = 0;
            if (++ld_tile == t1) { ld_tile = t0; if (++ld_pair == npairs) ld_pair = 0; }    // past the end: wrap (harmless re-read)
        }
    };
    auto publish_a = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        const H2 pa = split2(r.a0 * r.sa, r.a1 * r.sa);
        DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + wave * DWH_TILE + p * 1024) = pa.p[p];
    };
    auto publish_b = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        if (has_b) {
            const H2 pb = split2(r.b0 * r.sb, r.b1 * r.sb);
            const int bn = (NB == 8) ? wave : (wave & 1);
            DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + (8 + bn) * DWH_TILE + p * 1024) = pb.p[p];
        }
    };
    auto piece = [&](int par, int tile) {
        H2 f;
        const char* base = lds + par * DWH_BUF + tile * DWH_TILE + lane * 16;
        DH_UNROLL for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const u32x4*>(base + p * 1024);
        return f;
    };
    if (NP > 0) {
        // THREE raw register sets: with the matrix work halved this kernel is bound by the HBM stream (32 KB per k-pair and CU
        // against 768 MFMA cycles), so three k-pairs of loads stay in flight (96 KB per CU)
        Raw r0, r1, r2;
        load(r0);
        load(r1);
        load(r2);
        __builtin_amdgcn_sched_barrier(0);
        publish_a(r0, 0);
        publish_b(r0, 0);
        load(r0);                                          // pair 3 in flight
        __syncthreads();
        constexpr int H = NBW / 2;
        auto step = [&](int par, Raw& nxt) {
            if constexpr (NB == 8) {
                H2 a[2], b[2];
                H2 pc[MT];
                RawA raw;
                SplitStateH st;
                char* wbase = lds + (par ^ 1) * DWH_BUF + lane * 16;
                DH_UNROLL for (int ii = 0; ii < 2; ++ii) a[ii] = piece(par, (wave >> 1) * 2 + ii);
                DH_UNROLL for (int j = 0; j < 2; ++j) b[j] = piece(par, 8 + (wave & 1) * 4 + j);
                raw.lo[0] = nxt.a0; raw.hi[0] = nxt.a1; st.sc = nxt.sa;
                __builtin_amdgcn_sched_barrier(0);
                dw_half_steps_h<0, 0>(acc, a, b, pc, raw, st);
                DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(wbase + wave * DWH_TILE + p * 1024) = pc[0].p[p];
                DH_UNROLL for (int j = 0; j < 2; ++j) b[j] = piece(par, 8 + (wave & 1) * 4 + 2 + j);
                raw.lo[0] = nxt.b0; raw.hi[0] = nxt.b1; st.sc = nxt.sb;
                __builtin_amdgcn_sched_barrier(0);
                dw_half_steps_h<0, 2>(acc, a, b, pc, raw, st);
                DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(wbase + (8 + wave) * DWH_TILE + p * 1024) = pc[0].p[p];
                __builtin_amdgcn_sched_barrier(0);
                load(nxt);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                return;
            }
        };
        const float ratio = npairs == 2 ? sc.prod1 / sc.prod0 : 1.f;       // powers of two: exact
        // step p computes piece buffer p & 1 and publishes raw set (p + 1) % 3 (k-pair p + 1), which it refills with k-pair p + 4:
        // the buffers alternate with period 2, the register sets with period 3 -- both STATICALLY over six steps
        auto rescale = [&](int p) {
            if (npairs == 2 && p == NP1) {
                DH_UNROLL for (int i = 0; i < NA; ++i) DH_UNROLL for (int j = 0; j < NBW; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[i][j][r] *= ratio;
            }
        };
        int p = 0;
        for (; p + 5 < NP; p += 6) {
            rescale(p); step(0, r1);
            rescale(p + 1); step(1, r2);
            rescale(p + 2); step(0, r0);
            rescale(p + 3); step(1, r1);
            rescale(p + 4); step(0, r2);
            rescale(p + 5); step(1, r0);
        }
        if (p < NP) { rescale(p); step(0, r1); ++p; }
        if (p < NP) { rescale(p); step(1, r2); ++p; }
        if (p < NP) { rescale(p); step(0, r0); ++p; }
        if (p < NP) { rescale(p); step(1, r1); ++p; }
        if (p < NP) { rescale(p); step(0, r2); ++p; }
    }
    const float inv = sc.poison / (npairs == 2 ? sc.prod1 : sc.prod0);
    DH_UNROLL for (int i = 0; i < NA; ++i)
        DH_UNROLL for (int j = 0; j < NBW; ++j) {
            const int ot = (NB == 8) ? ((wave >> 1) * 2 + i) : wave;
            const int nt = (NB == 8) ? ((wave & 1) * 4 + j) : j;
            float* o = out + ((int64_t)ot * NB + nt) * 1024 + lane;
            DH_UNROLL for (int r = 0; r < 16; ++r) o[r * 64] = acc[i][j][r] * inv;
        }
}

// The aux jobs (64-wide B operand: jobs 0, 8, 11) in the two-piece arithmetic: wave w owns output rows [32w, 32w+32) x 64 columns
// (two accumulators), waves 0 / 1 publish the two B tiles.  Two raw register sets alternate with the two piece buffers.
// (The hazard hunt's variants of this body -- round 5, DESIGN.md section 4 -- live in scripts/micro/dw_aux_variants.h, not here.)
__device__ __forceinline__ void dw_body_aux_h(const DwJob& J, const DwScales& sc, int64_t t0, int64_t t1, float* __restrict__ out,
                                              int wave, int lane, char* lds) {
    constexpr int KQ = MT * 4;
    f32x16 acc[2];
    DH_UNROLL for (int j = 0; j < 2; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int T = (int)(t1 - t0);
    const int npairs = J.A2 ? 2 : 1;
    const int NP1 = T * (KQ / 2);
    const int NP = npairs * NP1;
    const bool has_b = wave < 2;
    struct Raw { f32x4 a0, a1, b0, b1; float sa, sb; };
    int ld_pair = 0, ld_kp = 0;
    int64_t ld_tile = t0;
    auto load = [&](Raw& r) {
        const float* A = ld_pair ? J.A2 : J.A1;
        const float* Bm = ld_pair ? J.B2 : J.B1;
        {
            const unsigned sxb = pow2_scale_bits((ld_pair ? sc.xt1 : sc.xt0)[ld_tile], H2_AT);
            const float sx = __builtin_bit_cast(float, sxb), sy = (ld_pair ? sc.prod1 : sc.prod0) * __builtin_bit_cast(float, pow2_inv_bits(sxb));
            const bool ha = ld_pair ? sc.heavy_a1 : sc.heavy_a0;
            r.sa = ha ? sx : sy;
            r.sb = ha ? sy : sx;
        }
        const int kq = 2 * ld_kp, m = kq >> 2, r4 = kq & 3;
        const f32x4* ga = reinterpret_cast<const f32x4*>(A + ld_tile * TILE_F) + ((((wave >> 1) * MT + m) * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
        r.a0 = __builtin_nontemporal_load(ga); r.a1 = __builtin_nontemporal_load(ga + 64);
        if (has_b) {
            const f32x4* gb = reinterpret_cast<const f32x4*>(Bm + ld_tile * AUXT_F) + ((m * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
            r.b0 = __builtin_nontemporal_load(gb); r.b1 = __builtin_nontemporal_load(gb + 64);
        }
        if (++ld_kp == KQ / 2) {
            ld_kp = 0;
            if (++ld_tile == t1) { ld_tile = t0; if (++ld_pair == npairs) ld_pair = 0; }
        }
    };
    auto piece = [&](int par, int tile) {
        H2 f;
        const char* base = lds + par * DWH_BUF + tile * DWH_TILE + lane * 16;
        DH_UNROLL for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const u32x4*>(base + p * 1024);
        return f;
    };
    auto scaled_split = [&](const f32x4& x0, const f32x4& x1, float s) {
        return split2(x0 * s, x1 * s);
    };
    auto publish_a = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        const H2 pa = scaled_split(r.a0, r.a1, r.sa);
        DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + wave * DWH_TILE + p * 1024) = pa.p[p];
    };
    auto publish_b = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        if (has_b) {
            const H2 pb = scaled_split(r.b0, r.b1, r.sb);
            DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + (8 + (wave & 1)) * DWH_TILE + p * 1024) = pb.p[p];
        }
    };
    if (NP > 0) {
        Raw r0, r1;
        load(r0);
        load(r1);
        __builtin_amdgcn_sched_barrier(0);
        publish_a(r0, 0);
        publish_b(r0, 0);
        load(r0);
        __syncthreads();
        auto step = [&](int par, Raw& nxt) {
            const H2 a = piece(par, wave), b0 = piece(par, 8), b1 = piece(par, 9);
            __builtin_amdgcn_sched_barrier(0);
            publish_a(nxt, par ^ 1);
            acc[0] = mfma3(a, b0, acc[0]);
            __builtin_amdgcn_sched_barrier(0);
            publish_b(nxt, par ^ 1);
            acc[1] = mfma3(a, b1, acc[1]);
            __builtin_amdgcn_sched_barrier(0);
            load(nxt);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        const float ratio = npairs == 2 ? sc.prod1 / sc.prod0 : 1.f;
        auto rescale = [&](int p) {
            if (npairs == 2 && p == NP1) {
                DH_UNROLL for (int j = 0; j < 2; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[j][r] *= ratio;
            }
        };
        int p = 0;
        for (; p + 1 < NP; p += 2) {
            rescale(p); step(0, r1);
            rescale(p + 1); step(1, r0);
        }
        if (p < NP) { rescale(p); step(0, r1); }
    }
    const float inv = sc.poison / (npairs == 2 ? sc.prod1 : sc.prod0);
    DH_UNROLL for (int j = 0; j < 2; ++j) {
        float* o = out + ((int64_t)wave * 2 + j) * 1024 + lane;
        DH_UNROLL for (int r = 0; r < 16; ++r) o[r * 64] = acc[j][r] * inv;
    }
}

__global__ __launch_bounds__(512, 1) void dw_f16x2_kernel(DwJobs jobs, DwGroups groups, int64_t ntiles, float* __restrict__ slabs,
                                                          int64_t gstride, const unsigned* __restrict__ absmax,
                                                          const unsigned* __restrict__ tmax) {
    // (sized for the aux jobs' three-piece image, below)
    __shared__ __attribute__((aligned(16))) char pieces[2 * (DWP_BUF > DWH_BUF ? DWP_BUF : DWH_BUF)];
    const int g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = dw_group_of(groups, g), gl = g - groups.wg0[k], gc = groups.wg0[k + 1] - groups.wg0[k];
    const int64_t t0 = ntiles * gl / gc, t1 = ntiles * (gl + 1) / gc;
    float* base = slabs + (int64_t)g * gstride;
    for (int job = groups.job0[k]; job < groups.job0[k + 1]; ++job) {
        const DwJob J = jobs.j[job];
        const DwScales sc = dw_job_scales(J, absmax, tmax, ntiles);
        // The three aux jobs (64-wide B: 6 % of the kernel's bytes, a quarter of a main job's MFMAs) run their own two-piece body.  (Round 4
        // ran them on the three-piece bf16 body because the two-piece form differed in one launch out of 5,000; round 5 traced that to
        // packed-fp32 multiplies under MFMAs -- layout.h -- and the library is built without them: 0 of 299,999 launches differ.)
        if (J.nb == 8) dw_body_pieces_h<8>(J, sc, t0, t1, base + J.off, wave, lane, pieces);
        else dw_body_aux_h(J, sc, t0, t1, base + J.off, wave, lane, pieces);
    }
}

__global__ __launch_bounds__(512, 1) void dw_bf16x3_kernel(DwJobs jobs, DwGroups groups, int64_t ntiles, float* __restrict__ slabs,
                                                           int64_t gstride) {
    __shared__ __attribute__((aligned(16))) char pieces[2 * DWP_BUF];
    const int g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = dw_group_of(groups, g), gl = g - groups.wg0[k], gc = groups.wg0[k + 1] - groups.wg0[k];
    const int64_t t0 = ntiles * gl / gc, t1 = ntiles * (gl + 1) / gc;
    float* base = slabs + (int64_t)g * gstride;
    for (int job = groups.job0[k]; job < groups.job0[k + 1]; ++job) {
        const DwJob J = jobs.j[job];
        if (J.nb == 8) dw_body_pieces<8>(J, t0, t1, base + J.off, wave, lane, pieces);
        else dw_body_pieces<2>(J, t0, t1, base + J.off, wave, lane, pieces);
    }
}

__global__ __launch_bounds__(512, 2) void dw_lds_kernel(DwJobs jobs, DwGroups groups, int64_t ntiles, float* __restrict__ slabs,
                                                        int64_t gstride) {
    __shared__ __attribute__((aligned(16))) char ring[DW_STAGES * DW_STAGE_BYTES];
    const int g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = dw_group_of(groups, g), gl = g - groups.wg0[k], gc = groups.wg0[k + 1] - groups.wg0[k];
    const int64_t t0 = ntiles * gl / gc, t1 = ntiles * (gl + 1) / gc;
    float* base = slabs + (int64_t)g * gstride;
    for (int job = groups.job0[k]; job < groups.job0[k + 1]; ++job) {
        const DwJob J = jobs.j[job];
        if (J.nb == 8) dw_body_lds<8>(J, t0, t1, base + J.off, wave, lane, ring);
        else dw_body_lds<2>(J, t0, t1, base + J.off, wave, lane, ring);
    }
}

// red[e] = sum over the workgroups of e's job group of slabs[g*gstride + e].  Eight independent partial sums (slab g goes to
// partial (g - first) % 8, combined pairwise in a fixed order: deterministic) keep eight 16-B loads in flight per lane; the slabs
// are read once: non-temporal.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t gstride, DwGroups groups,
                                                          float* __restrict__ red) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= gstride) return;
    int kg = 0;
    while (kg + 1 < groups.n && e >= groups.off0[kg + 1]) ++kg;
    const int G = groups.wg0[kg + 1] - groups.wg0[kg];
    f32x4 s[8];
    DH_UNROLL for (int k = 0; k < 8; ++k) s[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* p = slabs + e + (int64_t)groups.wg0[kg] * gstride;
    int g = 0;
    for (; g + 8 <= G; g += 8) {
        f32x4 v[8];
        DH_UNROLL for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + (int64_t)(g + k) * gstride));
        DH_UNROLL for (int k = 0; k < 8; ++k) s[k] += v[k];
    }
    for (; g < G; ++g) s[g & 7] += *reinterpret_cast<const f32x4*>(p + (int64_t)g * gstride);
    *reinterpret_cast<f32x4*>(red + e) = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// ---------------------------------------------------------------- per-tile partial sums -> [S][N_TILE_PART][256]
__global__ __launch_bounds__(256) void tpart_reduce_kernel(const float* __restrict__ tpart, int64_t ntiles, float* __restrict__ tred) {
    const int slot = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int64_t t0 = ntiles * s / S, t1 = ntiles * (s + 1) / S;
    float acc = 0.f;
    for (int64_t t = t0; t < t1; ++t) acc += tpart[(t * N_TILE_PART + slot) * 256 + threadIdx.x];
    tred[((int64_t)s * N_TILE_PART + slot) * 256 + threadIdx.x] = acc;
}

// ---------------------------------------------------------------- slab reduction + weight-norm fold -> flat gradient
struct FoldLin {
    int64_t boff, goff, voff, rsoff;      // parameter offsets; rowscale offset in packed
    int out, in;
    int jobA, a_c0, a_c1; float a_scale;  // parameter columns [a_c0,a_c1) come from slab jobA at slab col (c - a_c0)
    int jobB, b_c0, b_c1; float b_scale;
    int row_shift;                        // slab row = param row - row_shift
    int bias_slot;                        // tile-partial slot of the bias gradient (slab indexing)
    int special;                          // 1: sdf lin8 (row 0 from TP 9+10, bias 0 from TP 11) ; 2: colour lin4 (TP 16..18, 19)
    int row_base;                         // first global row id of this linear
};
struct FoldTable { FoldLin lin[N_SDF + N_COL]; int total_rows; };
struct SlabPtrs { const float* out[16]; int nb[16]; };

__device__ __forceinline__ float slab_elem(const float* __restrict__ slab, int nb, int o, int i) {
    const int w = o >> 5, ro = o & 31, j = i >> 5;
    const int r = (ro & 3) + 4 * (ro >> 3);
    const int lane = (i & 31) + 32 * ((ro >> 2) & 1);
    return slab[((int64_t)w * nb + j) * 1024 + r * 64 + lane];
}

__global__ __launch_bounds__(256) void fold_kernel(FoldTable T, SlabPtrs S, const float* __restrict__ tred, int nS,
                                                   const float* __restrict__ params, const float* __restrict__ packed,
                                                   float* __restrict__ grad) {
    __shared__ float s_red[4];
    const int rowid = blockIdx.x, tid = threadIdx.x;
    int li = 0;
    for (int k = 1; k < N_SDF + N_COL; ++k) if (rowid >= T.lin[k].row_base) li = k;
    const FoldLin Ln = T.lin[li];
    const int o = rowid - Ln.row_base;
    auto tsum = [&](int slot, int c) {
        float s = 0.f;
        for (int k = 0; k < nS; ++k) s += tred[((int64_t)k * N_TILE_PART + slot) * 256 + c];
        return s;
    };
    float dw[2] = {0.f, 0.f}, vv[2] = {0.f, 0.f};
    float dot = 0.f;
    DH_UNROLL for (int q = 0; q < 2; ++q) {
        const int i = tid + 256 * q;
        if (i < Ln.in) {
            float d = 0.f;
            if (Ln.special == 2) d = tsum(16 + o, i);                                     // colour lin4 rows
            else if (Ln.special == 1 && o == 0) d = tsum(9, i) + tsum(10, i);             // sdf lin8 row 0
            else {
                const int so = o - Ln.row_shift;
                if (i >= Ln.a_c0 && i < Ln.a_c1) d = Ln.a_scale * slab_elem(S.out[Ln.jobA], S.nb[Ln.jobA], so, i - Ln.a_c0);
                else if (Ln.jobB >= 0 && i >= Ln.b_c0 && i < Ln.b_c1)
                    d = Ln.b_scale * slab_elem(S.out[Ln.jobB], S.nb[Ln.jobB], so, i - Ln.b_c0);
            }
            dw[q] = d;
            vv[q] = params[Ln.voff + (int64_t)o * Ln.in + i];
            dot = fmaf(d, vv[q], dot);
        }
    }
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if ((tid & 63) == 0) s_red[tid >> 6] = dot;
    __syncthreads();
    dot = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    const float inv = packed[Ln.rsoff + (PACK.invnorm - PACK.rowscale) + o];
    const float gval = params[Ln.goff + o];
    DH_UNROLL for (int q = 0; q < 2; ++q) {
        const int i = tid + 256 * q;
        if (i < Ln.in) grad[Ln.voff + (int64_t)o * Ln.in + i] = gval * inv * (dw[q] - dot * inv * inv * vv[q]);
    }
    if (tid < 64) {
        // the row's bias gradient: nS tile-partial sums, one per lane of the first wave and a fixed shuffle tree (ONE thread adding
        // them in a dependent chain of nS L2 loads was the critical path of this kernel: 52 -> 2x us)
        int slot, c;
        if (Ln.special == 2) { slot = 19; c = o; }
        else if (Ln.special == 1 && o == 0) { slot = 11; c = 0; }
        else { slot = Ln.bias_slot; c = o - Ln.row_shift; }
        float b = 0.f;
        for (int k = tid; k < nS; k += 64) b += tred[((int64_t)k * N_TILE_PART + slot) * 256 + c];
        DH_UNROLL for (int off = 32; off > 0; off >>= 1) b += __shfl_xor(b, off);
        if (tid == 0) {
            grad[Ln.goff + o] = dot * inv;
            grad[Ln.boff + o] = b;
        }
    }
}

static FoldTable build_fold_table() {
    FoldTable T{};
    int row = 0;
    for (int l = 0; l < N_SDF; ++l) {
        FoldLin& F = T.lin[l];
        const LinOff o = sdf_off(l);
        F.boff = o.bias; F.goff = o.g; F.voff = o.v; F.rsoff = PACK.rowscale + (int64_t)l * 260;
        F.out = SDF_DIMS[l].out; F.in = SDF_DIMS[l].in;
        F.jobA = l; F.a_c0 = 0; F.a_c1 = F.in; F.a_scale = 1.f;
        F.jobB = -1; F.b_c0 = F.b_c1 = 0; F.b_scale = 1.f;
        F.row_shift = 0; F.bias_slot = l; F.special = 0;
        if (l == 4) { F.a_c1 = SKIP_OUT; F.a_scale = INV_SQRT2; F.jobB = 8; F.b_c0 = SKIP_OUT; F.b_c1 = 256; F.b_scale = INV_SQRT2; }
        if (l == 8) { F.jobA = 9; F.row_shift = 1; F.special = 1; F.bias_slot = 8; }
        F.row_base = row; row += F.out;
    }
    for (int l = 0; l < N_COL; ++l) {
        FoldLin& F = T.lin[N_SDF + l];
        const LinOff o = col_off(l);
        F.boff = o.bias; F.goff = o.g; F.voff = o.v; F.rsoff = PACK.rowscale + (int64_t)N_SDF * 260 + (int64_t)l * 256;
        F.out = COL_DIMS[l].out; F.in = COL_DIMS[l].in;
        F.jobA = 11 + l; F.a_c0 = 0; F.a_c1 = F.in; F.a_scale = 1.f;
        F.jobB = -1; F.b_c0 = F.b_c1 = 0; F.b_scale = 1.f;
        F.row_shift = 0; F.bias_slot = 12 + l; F.special = 0;
        if (l == 0) { F.jobA = 11; F.a_c0 = 0; F.a_c1 = CAUX; F.jobB = 10; F.b_c0 = CAUX; F.b_c1 = 289; }
        if (l == 4) { F.special = 2; F.jobA = -1; }
        F.row_base = row; row += F.out;
    }
    T.total_rows = row;
    return T;
}

// dW slab workspace: G split blocks of (sum over 15 jobs of 256 x nb*32) floats, plus one reduced block
static const int DW_NBS[15] = {2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8};
static int64_t dw_gstride() {
    int64_t n = 0;
    for (int j = 0; j < 15; ++j) n += (int64_t)8 * DW_NBS[j] * 1024;
    return n;
}
int64_t dw_slab_floats(int G) { return (int64_t)(G + 1) * dw_gstride(); }

static void build_dw_jobs(const Workspace& w, float* red, DwJobs& J, SlabPtrs& S) {
    const int64_t nt = w.ntiles;
    int64_t off = 0;
    for (int j = 0; j < 15; ++j) { J.j[j].nb = DW_NBS[j]; J.j[j].off = off; S.out[j] = red + off; S.nb[j] = DW_NBS[j]; off += (int64_t)8 * DW_NBS[j] * 1024; }
    J.n = 15;
    auto T_ = [&](float* base, int idx) { return base + (int64_t)idx * nt * TILE_F; };
    // operand classes of the two-piece fp16 kernel: (absmax class, tmax class) of A and B for each pair
    auto C_ = [&](int j, int i, int ca, int ha, int cb, int hb) { J.j[j].ca[i] = ca; J.j[j].ha[i] = ha; J.j[j].cb[i] = cb; J.j[j].hb[i] = hb; };
    for (int j = 0; j < 15; ++j) { C_(j, 0, -1, -1, -1, -1); C_(j, 1, -1, -1, -1, -1); }
    J.j[0].A1 = T_(w.zbar, 0); J.j[0].B1 = w.eaux; J.j[0].A2 = T_(w.asave, 0); J.j[0].B2 = w.t0aux;
    C_(0, 0, ABSMAX_ZBAR + 0, TMAX_ZBAR + 0, -1, -1); C_(0, 1, ABSMAX_ASAVE + 0, -1, ABSMAX_T0AUX, TMAX_T0AUX);
    for (int l = 1; l <= 7; ++l) {
        J.j[l].A1 = T_(w.zbar, l); J.j[l].B1 = T_(w.act, l - 1);
        J.j[l].A2 = T_(w.asave, l); J.j[l].B2 = T_(w.tsave, l - 1);
        C_(l, 0, ABSMAX_ZBAR + l, TMAX_ZBAR + l, ABSMAX_ACT, -1); C_(l, 1, ABSMAX_ASAVE + l, -1, ABSMAX_TSAVE + l - 1, TMAX_TSAVE + l - 1);
    }
    J.j[8].A1 = T_(w.zbar, 4); J.j[8].B1 = w.eaux; J.j[8].A2 = T_(w.asave, 4); J.j[8].B2 = w.t0aux;
    C_(8, 0, ABSMAX_ZBAR + 4, TMAX_ZBAR + 4, -1, -1); C_(8, 1, ABSMAX_ASAVE + 4, -1, ABSMAX_T0AUX, TMAX_T0AUX);
    J.j[9].A1 = w.featbar; J.j[9].B1 = T_(w.act, 7); J.j[9].A2 = nullptr; J.j[9].B2 = nullptr;
    C_(9, 0, ABSMAX_FEATBAR, TMAX_FEATBAR, ABSMAX_ACT, -1);
    J.j[10].A1 = T_(w.czbar, 0); J.j[10].B1 = w.feat; J.j[10].A2 = nullptr; J.j[10].B2 = nullptr;
    C_(10, 0, ABSMAX_CZBAR + 0, TMAX_CZBAR + 0, ABSMAX_FEAT, -1);
    J.j[11].A1 = T_(w.czbar, 0); J.j[11].B1 = w.caux; J.j[11].A2 = nullptr; J.j[11].B2 = nullptr;
    C_(11, 0, ABSMAX_CZBAR + 0, TMAX_CZBAR + 0, ABSMAX_CAUX, -1);
    for (int l = 1; l <= 3; ++l) {
        J.j[11 + l].A1 = T_(w.czbar, l); J.j[11 + l].B1 = T_(w.cact, l - 1); J.j[11 + l].A2 = nullptr; J.j[11 + l].B2 = nullptr;
        C_(11 + l, 0, ABSMAX_CZBAR + l, TMAX_CZBAR + l, ABSMAX_CACT + l - 1, -1);
    }
}

// job groups: consecutive jobs, workgroup shares proportional to the jobs' cost, every group >= 1 workgroup
static DwGroups build_dw_groups(const DwJobs& J, int G) {
    DwGroups Gp{};
    const int ng = DW_GROUPS < G ? DW_GROUPS : (G < 1 ? 1 : G);
    double cost[16], total = 0.0;
    // cost of a job per tile: a main job (64 KB + 64 KB of operands, 48 MFMAs per wave and 16 points) = 16; an aux job (64 KB +
    // 16 KB, a quarter of the MFMAs) is bound by its bytes: DW_AUX_COST, measured (profiles/r03_ab_dw_job_groups.json; round 4: profiles/r04_dw_aux_reproducibility.json);
    // two operand pairs = twice
    for (int j = 0; j < J.n; ++j) { cost[j] = (J.j[j].nb == 8 ? 16.0 : (double)DW_AUX_COST) * (J.j[j].A2 ? 2 : 1); total += cost[j]; }
    // greedy contiguous partition: close a group when its cost reaches the running target
    Gp.n = ng; Gp.job0[0] = 0;
    double acc = 0.0; int k = 1;
    for (int j = 0; j < J.n && k < ng; ++j) {
        acc += cost[j];
        const int jobs_left = J.n - (j + 1), groups_left = ng - k;
        if (acc >= total * k / ng - 1e-9 || jobs_left <= groups_left) { if (jobs_left >= groups_left) { Gp.job0[k++] = j + 1; } }
    }
    while (k < ng) { Gp.job0[k] = Gp.job0[k - 1] + 1; ++k; }
    Gp.job0[ng] = J.n;
    // workgroup shares (largest-remainder rounding, at least one each)
    double gc[DW_GROUPS + 1]; int cnt[DW_GROUPS + 1]; int used = 0;
    for (int g2 = 0; g2 < ng; ++g2) {
        gc[g2] = 0.0;
        for (int j = Gp.job0[g2]; j < Gp.job0[g2 + 1]; ++j) gc[g2] += cost[j];
        cnt[g2] = (int)(G * gc[g2] / total); if (cnt[g2] < 1) cnt[g2] = 1; used += cnt[g2];
    }
    while (used < G) { int best = 0; double bv = -1.0; for (int g2 = 0; g2 < ng; ++g2) { const double v = gc[g2] / cnt[g2]; if (v > bv) { bv = v; best = g2; } } ++cnt[best]; ++used; }
    while (used > G) { int best = -1; double bv = 1e300; for (int g2 = 0; g2 < ng; ++g2) { if (cnt[g2] > 1) { const double v = gc[g2] / cnt[g2]; if (v < bv) { bv = v; best = g2; } } } --cnt[best]; --used; }
    Gp.wg0[0] = 0;
    for (int g2 = 0; g2 < ng; ++g2) { Gp.wg0[g2 + 1] = Gp.wg0[g2] + cnt[g2]; Gp.off0[g2] = J.j[Gp.job0[g2]].off; }
    Gp.off0[ng] = dw_gstride();
    return Gp;
}

// stage 1: the split-K weight-gradient GEMMs (one kernel)
int launch_weight_grads_gemm(const Workspace& w, float* slabs, int G, int arith, hipStream_t st) {
    const int64_t gstride = dw_gstride();
    DwJobs J{};
    SlabPtrs S{};
    build_dw_jobs(w, slabs + (int64_t)G * gstride, J, S);
    const DwGroups Gp = build_dw_groups(J, G);
    if (arith == ARITH_FP32) hipLaunchKernelGGL(dw_lds_kernel, dim3(G), dim3(512), 0, st, J, Gp, w.ntiles, slabs, gstride);
    else if (arith == ARITH_F16) hipLaunchKernelGGL(dw_f16x2_kernel, dim3(G), dim3(512), 0, st, J, Gp, w.ntiles, slabs, gstride,
                                                    reinterpret_cast<const unsigned*>(w.absmax), reinterpret_cast<const unsigned*>(w.tmax));
    else hipLaunchKernelGGL(dw_bf16x3_kernel, dim3(G), dim3(512), 0, st, J, Gp, w.ntiles, slabs, gstride);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// stage 2: slab + tile-partial reductions, weight-norm fold into the flat gradient
int launch_weight_grads_fold(const Workspace& w, float* slabs, float* tred, int G, int nS, const float* params,
                             const float* packed, float* grad, hipStream_t st) {
    const int64_t gstride = dw_gstride();
    float* red = slabs + (int64_t)G * gstride;
    DwJobs J{};
    SlabPtrs S{};
    build_dw_jobs(w, red, J, S);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((gstride / 4 + 255) / 256)), dim3(256), 0, st, slabs, gstride,
                       build_dw_groups(J, G), red);
    hipLaunchKernelGGL(tpart_reduce_kernel, dim3(N_TILE_PART, nS), dim3(256), 0, st, w.tpart, w.ntiles, tred);
    static const FoldTable T = build_fold_table();
    hipLaunchKernelGGL(fold_kernel, dim3(T.total_rows), dim3(256), 0, st, T, S, tred, nS, params, packed, grad);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh

