// Register-resident ("transposed") layer chain: does it beat the product's tile16.h core?   (round 3)
//
// The shipping chain kernels keep a [64 points x 256] fp32 activation image in LDS; every one of the 4 waves fetches ALL of it per
// layer and splits it into bf16 pieces itself (4 x redundant split = 21 % of a kernel), and each wave pulls its own weight
// fragments from L2 (12.9 GB per launch).  This micro measures the alternative:
//   * out^T[feature][point] = W[feature][k] * act^T[k][point]: the WEIGHTS are the MFMA A operand, the activations the B operand;
//     a wave owns 32 points (the MFMA's column axis) and ALL 256 output features (8 accumulators of 32 x 32 = 128 registers).
//   * the accumulator layout (feature rows in registers, point on the lane) IS the B-operand layout of the next layer with a
//     permuted k order (the packer permutes the weights to match), so activations never leave registers: bias + softplus + ONE
//     3-way bf16 split per value, no LDS image, no hand-off barrier, no redundant split.
//   * the epilogue of m-tile m+1 of layer l is dealt out, one micro-step per MFMA, under the MFMAs of k-steps 2m, 2m+1 of layer
//     l+1 (which need only m-tile m's output): one wave per SIMD (512 registers: two accumulator sets) still hides it.
//   * weight pieces enter the CU ONCE per 128 points (4 waves share them) through an LDS ring filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds, 24 KB per 16-deep k-step, dealt singly between MFMAs), one raw s_barrier per k-step.
// Arms: this chain, its bare MFMA + ring stream (no epilogue), and the tile16.h loop (same 8 x [256 x 256] layers, softplus, data
// kept O(1) with random signs in every arm -- collapsed activations raise the clock by 20 %).  Results and the earlier forms of this
// file: profiles/r03_ab_register_resident_chain.json.  The product kernels built from the -DTC_BUILTIN form (compiler-allocated
// accumulators, LDS-DMA dealt singly): dynhor_amd/csrc/chain_t.hip; this file stays as the timing study (its values are NOT
// checked against a reference -- the product kernels' are, and bring-up found two inline-asm hazards this micro shares in
// principle: scripts/isa_inflight_check.py).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tchain_micro.hip -o tchain_micro
#include "../../dynhor_amd/csrc/tile16.h"
#include <cstdio>
#include <vector>
#include <algorithm>
using namespace dh;

constexpr int NM = 8;                        // 32-feature m-tiles of a 256-wide layer
constexpr int STAGE_BYTES = NM * 3 * 1024;   // one k-step of weight pieces: 8 m-tiles x 3 pieces x 1 KiB fragments
constexpr int NSTAGE = 5, DEPTH = 3;         // ring slots / k-steps in flight (NSTAGE >= DEPTH + 2: the barrier sits mid-step)
constexpr int DMA_PER_WAVE = NM * 3 / 4;     // 6 LDS-DMA instructions per wave and k-step

struct Pieces { u32x4 p[3]; };               // one k-step of the activation (B) operand: 3 bf16x8 pieces
struct AFrag { u32x4 p[3]; };                // one m-tile's weight (A) fragments of a k-step
struct EpiSt { f32x2 x, t, e, u; float lane_rnd; };
// activations must stay O(1) with random signs in BOTH arms (values that collapse to ~0 raise the clock by ~20 %: rule 25): the
// pre-activation is acc * 1e-3 + a per-lane offset in (-1, 1) + a per-register offset -> about half the outputs are ~0 (as after
// softplus(100 z) in the real MLP), the rest spread over (0, 1.5)
__host__ __device__ constexpr float reg_rnd(int r) { return (float)((r * 37 + 11) % 64) * (1.f / 64.f) - 0.5f; }

template <int OFF>
__device__ __forceinline__ u32x4 lds_b128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// one micro-step of the epilogue of an m-tile whose 16 finished values sit in x (plain registers): pair j = STEP / 12 = values
// 2j, 2j+1 -> u32 j % 4 of the pieces of k-step half j / 4.  12 steps of 2-3 vector ops: bias / scale, -|x| c, exp2, 1 + e, log2,
// max, fma, then the 3-way split.  M only seeds the synthetic per-register offset of this micro.
#ifndef TC_VARIANT
#define TC_VARIANT 1
#endif
template <int M, int STEP>
__device__ __forceinline__ void epi_step(const f32x16& x, Pieces (&out)[2], EpiSt& st) {
    constexpr int j = STEP / 12, s = STEP % 12, half = j / 4, q = j % 4, r0 = 2 * j;
    if constexpr (s == 0) { st.x[0] = fmaf(x[r0], 1e-3f, st.lane_rnd + reg_rnd(M * 16 + r0)); }
    else if constexpr (s == 1) { st.x[1] = fmaf(x[r0 + 1], 1e-3f, st.lane_rnd + reg_rnd(M * 16 + r0 + 1)); }
    else if constexpr (s == 2) { st.t[0] = -fabsf(st.x[0]) * (SOFTPLUS_BETA * 1.44269504088896f); st.t[1] = -fabsf(st.x[1]) * (SOFTPLUS_BETA * 1.44269504088896f); }
    else if constexpr (s == 3) { st.e[0] = __builtin_amdgcn_exp2f(st.t[0]); st.e[1] = __builtin_amdgcn_exp2f(st.t[1]); }
    else if constexpr (s == 4) { st.e[0] = 1.f + st.e[0]; st.e[1] = 1.f + st.e[1]; }
    else if constexpr (s == 5) { st.e[0] = __builtin_amdgcn_logf(st.e[0]); st.e[1] = __builtin_amdgcn_logf(st.e[1]); }
    else if constexpr (s == 6) { st.t[0] = fmaxf(st.x[0], 0.f); st.t[1] = fmaxf(st.x[1], 0.f); }
    else if constexpr (s == 7) { st.x[0] = fmaf(st.e[0], 0.69314718055995f / SOFTPLUS_BETA, st.t[0]); st.x[1] = fmaf(st.e[1], 0.69314718055995f / SOFTPLUS_BETA, st.t[1]); }
    else if constexpr (s == 8) { const unsigned h = pack_bf16x2(st.x); out[half].p[0][q] = h; st.u = unpack_bf16x2(h); }
    else if constexpr (s == 9) { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; }
    else if constexpr (s == 10) { const unsigned h = pack_bf16x2(st.x); out[half].p[1][q] = h; st.u = unpack_bf16x2(h); }
    else { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; out[half].p[2][q] = pack_bf16x2(st.x); }
}
template <int M, int I, int N>
__device__ __forceinline__ void epi_only(const f32x16& x, Pieces (&out)[2], EpiSt& st) {
    if constexpr (I < N) { epi_step<M, I>(x, out, st); epi_only<M, I + 1, N>(x, out, st); }
}

// The two accumulator sets are HAND-ALLOCATED accumulator-file registers (set 0 = a[0:127], set 1 = a[128:255]; m-tile m of a set
// = 16 registers from base + 16 m): hipcc cannot be told to keep a value in the accumulator file and moves tuples between the
// files with v_accvgpr_read / _write.  (An early reading of this micro priced those at ~60 cycles each; that was the clumped LDS-DMA
// issue cost -- profiles/r03_ab_register_resident_chain.json -- and this hand-allocated arm measured SLOWER than the
// compiler-allocated one, 225 vs 234-240 TFLOP/s.)  MFMAs and the LDS stash writes name the registers literally; the compiler
// never sees them (clobber list below).
#define TC_ALL_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
// -DTC_BUILTIN: the compiler allocates both accumulator sets (f32x16 arrays, __builtin_amdgcn_mfma) and reads finished values
// with v_accvgpr_read pairs inside the epilogue steps -- the form that measured 237-240 TFLOP/s with the DMA in a clump
struct Acc { f32x16 s0[NM], s1[NM]; };
template <int BASE, bool ZERO>
__device__ __forceinline__ void mfma_lit(const u32x4& a, const u32x4& b) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %2, %3, 0" ::"n"(BASE), "n"(BASE + 15), "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], %2, %3, a[%c0:%c1]" ::"n"(BASE), "n"(BASE + 15), "v"(a), "v"(b));
}
// MFMA I (0..11) of group G (m-tiles 2G, 2G+1) into set NB: product-major, consecutive MFMAs hit different accumulators;
// FIRST: the layer's first k-step starts the accumulators from 0 (inline constant C operand) instead of clearing them
template <int NB, int G, int I, bool FIRST>
__device__ __forceinline__ void mfma_step(Acc& A, const AFrag (&a)[2], const Pieces& b) {
    constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first (tile16.h mfma6)
    constexpr int p = I / 2, t = I % 2, mt = 2 * G + t;
#ifdef TC_BUILTIN
    f32x16& acc = NB ? A.s1[mt] : A.s0[mt];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t].p[pw[p]]), __builtin_bit_cast(bf16x8, b.p[px[p]]), acc, 0, 0, 0);
#else
    mfma_lit<NB + 16 * mt, FIRST && p == 0>(a[t].p[pw[p]], b.p[px[p]]);
#endif
}
// the 12 MFMAs of a group, each followed by one epilogue micro-step EBASE + I of m-tile EM (EM < 0: bare MFMAs)
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct Ring {
    unsigned rd_addr;          // LDS byte address of the slot being read + lane * 16
    unsigned rd_slot;
    unsigned is_slot;          // slot the stage being issued goes to
    unsigned is_goff;          // byte offset of that stage in the packed weights (wraps over the 8 layers)
    unsigned gbytes;           // size of the packed weights
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lds_base;
    char* lds;
    int wave, lane;
    unsigned stash;            // LDS byte address of this wave's accumulator stash + lane * 16
};
// One LDS-DMA of the stage being issued: fragment wave + 4 i (1 KiB: 64 lanes x 16 B).  An LDS-DMA costs the issuing wave
// 60-180 cycles (MI355X guide, cycle constants), 6 in one clump idled the matrix pipe for ~460 cycles per k-step in the first
// version of this micro; dealt singly between MFMAs, each hides most of its cost behind the MFMA in flight.
template <int I>
__device__ __forceinline__ void ring_issue_one(Ring& R) {
    const unsigned frag = R.wave + 4 * I;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.lds + R.is_slot * STAGE_BYTES + frag * 1024),
                                             16, R.lane * 16, R.is_goff + frag * 1024, 0, 0);
    if constexpr (I == DMA_PER_WAVE - 1) {
        R.is_slot = (R.is_slot + 1 == NSTAGE) ? 0 : R.is_slot + 1;
        R.is_goff = (R.is_goff + STAGE_BYTES == R.gbytes) ? 0 : R.is_goff + STAGE_BYTES;
    }
}
__device__ __forceinline__ void ring_issue(Ring& R) {
    ring_issue_one<0>(R); ring_issue_one<1>(R); ring_issue_one<2>(R); ring_issue_one<3>(R); ring_issue_one<4>(R); ring_issue_one<5>(R);
}
__device__ __forceinline__ void ring_advance_read(Ring& R) {
    R.rd_slot = (R.rd_slot + 1 == NSTAGE) ? 0 : R.rd_slot + 1;
    R.rd_addr = R.lds_base + R.rd_slot * STAGE_BYTES + R.lane * 16;
}
// DA / DB: the LDS-DMA piece issued after MFMA 3 / MFMA 9 of the group (-1: none)
template <int NB, int G, int EM, int EBASE, bool FIRST, int DA, int DB, int I>
__device__ __forceinline__ void group_steps(Acc& A, const AFrag (&a)[2], const Pieces& b, const f32x16& xE, Pieces (&bn)[2], EpiSt& st, Ring& R) {
    if constexpr (I < 12) {
        mfma_step<NB, G, I, FIRST>(A, a, b);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == 3 && DA >= 0) { ring_issue_one<DA>(R); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (I == 9 && DB >= 0) { ring_issue_one<DB>(R); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (EM >= 0 && EM < NM) {
            epi_step<EM, EBASE + I>(xE, bn, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        group_steps<NB, G, EM, EBASE, FIRST, DA, DB, I + 1>(A, a, b, xE, bn, st, R);
    }
}

// Finished accumulators reach the plain registers through LDS: DS instructions take their data straight from the accumulator
// file.  Per m-tile 4 ds_write_b128 (a[..] source) + 4 ds_read_b128 into v[..]; a wave-private 4 KB stash; LDS executes one
// wave's accesses in order, so the reads may be issued right behind the writes.
template <int BASE>
__device__ __forceinline__ void stash_write(unsigned addr) {
    asm volatile("ds_write_b128 %0, a[%c1:%c2]\n\tds_write_b128 %0, a[%c3:%c4] offset:1024\n\t"
                 "ds_write_b128 %0, a[%c5:%c6] offset:2048\n\tds_write_b128 %0, a[%c7:%c8] offset:3072"
                 ::"v"(addr), "n"(BASE), "n"(BASE + 3), "n"(BASE + 4), "n"(BASE + 7), "n"(BASE + 8), "n"(BASE + 11), "n"(BASE + 12), "n"(BASE + 15)
                 : "memory");
}
__device__ __forceinline__ void stash_read(f32x16& x, unsigned addr) {
    f32x4 v0, v1, v2, v3;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v0) : "v"(addr));
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(v1) : "v"(addr));
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(v2) : "v"(addr));
    asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(v3) : "v"(addr));
    DH_UNROLL for (int i = 0; i < 4; ++i) { x[i] = v0[i]; x[4 + i] = v1[i]; x[8 + i] = v2[i]; x[12 + i] = v3[i]; }
}

template <int I>
__device__ __forceinline__ void acc_write_one(float v) { asm volatile("v_accvgpr_write_b32 a[%c0], %1" ::"n"(I), "v"(v)); }
template <int I = 0>
__device__ __forceinline__ void acc_write_all(const float (&v)[128]) {
    if constexpr (I < 128) { acc_write_one<I>(v[I]); acc_write_all<I + 1>(v); }
}

template <int G>
__device__ __forceinline__ void read_group(AFrag (&a)[2], unsigned addr) {
    DH_UNROLL for (int t = 0; t < 2; ++t) {
        a[t].p[0] = lds_b128<((2 * G + 0) * 3 + 0) * 1024>(addr + t * 3072);
        a[t].p[1] = lds_b128<((2 * G + 0) * 3 + 1) * 1024>(addr + t * 3072);
        a[t].p[2] = lds_b128<((2 * G + 0) * 3 + 2) * 1024>(addr + t * 3072);
    }
}

// one k-step into accumulator set NB: entering, a0 holds group 0's fragments; leaving, a0 holds group 0 of the NEXT k-step.
// EM / EH: the epilogue m-tile dealt under this k-step's MFMAs (its values in xE) and which half (48 micro-steps) of it.
// PRE >= 0: in the last group the finished accumulators of m-tile PRE of the OTHER set go accumulator file -> stash -> xPre.
template <int NB, int EM, int EH, int PRE, bool FIRST>
__device__ __forceinline__ void kstep(Acc& A, const Pieces& b, const f32x16& xE, Pieces (&bn)[2], EpiSt& st, AFrag (&a0)[2], AFrag (&a1)[2], Ring& R,
                                      f32x16& xPre) {
    read_group<1>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<NB, 0, EM, EH * 48 + 0, FIRST, 3, 4, 0>(A, a0, b, xE, bn, st, R);       // pieces 3, 4 of the stage begun last k-step
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_group<2>(a0, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<NB, 1, EM, EH * 48 + 12, FIRST, 5, -1, 0>(A, a1, b, xE, bn, st, R);     // piece 5: the stage is fully issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // the NEXT k-step's pieces: this wave's DMAs for it have landed once at most DEPTH-1 younger groups are outstanding;
    // after the barrier everyone's have, and everyone has left the previous k-step (its slot may be refilled)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (DEPTH - 1)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    read_group<3>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<NB, 2, EM, EH * 48 + 24, FIRST, 0, 1, 0>(A, a0, b, xE, bn, st, R);      // a new stage: its slot was freed by the barrier
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ring_advance_read(R);
    read_group<0>(a0, R.rd_addr);
#ifndef TC_BUILTIN
    if constexpr (PRE >= 0 && PRE < NM) { stash_write<(128 - NB) + 16 * PRE>(R.stash); stash_read(xPre, R.stash); }
#endif
    __builtin_amdgcn_sched_barrier(0);
    group_steps<NB, 3, EM, EH * 48 + 36, FIRST, 2, -1, 0>(A, a1, b, xE, bn, st, R);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// k-steps 2M, 2M+1 of a layer (input pieces = the epilogue of m-tile M of the previous layer) with the epilogue of m-tile M+1
// dealt under them (its values were brought into xA (M+1 even) / xB (odd) during the previous pair) and m-tile M+2 fetched
template <int NB, int M, int EPI>
__device__ __forceinline__ void mpair(Acc& A, f32x16& xA, f32x16& xB, Pieces (&bA)[2], Pieces (&bB)[2], EpiSt& st, AFrag (&a0)[2], AFrag (&a1)[2],
                                      Ring& R) {
    if constexpr (M < NM) {
        constexpr int EM = EPI ? M + 1 : -1, PRE = EPI ? M + 2 : -1;
#ifdef TC_BUILTIN
        // the epilogue reads the finished set directly (m-tile M + 1 of the set this layer does NOT write)
        const f32x16& xO = (NB ? A.s0 : A.s1)[M + 1 < NM ? M + 1 : 0];
        const f32x16& xE0 = xO; const f32x16& xE1 = xO;
#else
        const f32x16& xE0 = xB; const f32x16& xE1 = xA;
#endif
        // even M: current pieces in bA, next into bB, epilogue values (m-tile M+1, odd) in xB, m-tile M+2 fetched into xA
        if constexpr (M % 2 == 0 || !EPI) {
            kstep<NB, EM, 0, -1, M == 0>(A, bA[0], xE0, bB, st, a0, a1, R, xA);
            kstep<NB, EM, 1, PRE, false>(A, bA[1], xE0, bB, st, a0, a1, R, xA);
        } else {
            kstep<NB, EM, 0, -1, false>(A, bB[0], xE1, bA, st, a0, a1, R, xB);
            kstep<NB, EM, 1, PRE, false>(A, bB[1], xE1, bA, st, a0, a1, R, xB);
        }
        mpair<NB, M + 1, EPI>(A, xA, xB, bA, bB, st, a0, a1, R);
    }
}
// one layer: accumulates into set NB from the finished set 128 - NB
template <int NB, int EPI>
__device__ __forceinline__ void layer(Acc& A, Pieces (&bA)[2], Pieces (&bB)[2], EpiSt& st, AFrag (&a0)[2], AFrag (&a1)[2], Ring& R) {
    f32x16 xA, xB;
#ifdef TC_BUILTIN
    DH_UNROLL for (int m = 0; m < NM; ++m) DH_UNROLL for (int r = 0; r < 16; ++r) (NB ? A.s1 : A.s0)[m][r] = 0.f;
    epi_only<0, 0, 96>((NB ? A.s0 : A.s1)[0], bA, st);
#else
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the previous layer's last MFMAs must have written their results
    stash_write<(128 - NB) + 0>(R.stash); stash_read(xA, R.stash);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    stash_write<(128 - NB) + 16>(R.stash); stash_read(xB, R.stash);
    epi_only<0, 0, 96>(xA, bA, st);           // the one exposed epilogue of a layer: m-tile 0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    if constexpr (!EPI) { bB[0] = bA[0]; bB[1] = bA[1]; }
    mpair<NB, 0, EPI>(A, xA, xB, bA, bB, st, a0, a1, R);
}

// EPI = 0: timing-only arm -- no epilogue at all (the pieces of m-tile 0 feed every k-step): the ceiling of MFMA + weight ring
template <int EPI>
__global__ __launch_bounds__(256, 1) void tchain(const bf16x8* __restrict__ wp, float* out, int layer_pairs, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * STAGE_BYTES + 4 * 4096];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Ring R;
    R.lds = lds; R.lds_base = (unsigned)(uintptr_t)lds; R.wave = wave; R.lane = lane;
    R.gbytes = 8u * 16u * STAGE_BYTES;
    R.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wp, 0, R.gbytes, 0x00020000);
    R.is_goff = 0;
    R.is_slot = 0; R.rd_slot = 0; R.rd_addr = R.lds_base + lane * 16;
    R.stash = R.lds_base + NSTAGE * STAGE_BYTES + wave * 4096 + lane * 16;
    Acc A;
#ifndef TC_BUILTIN
    asm volatile("" ::: TC_ALL_AGPRS);            // the accumulator file is ours: the descriptor must allocate all of it
#endif
    {                                             // set 0 = "the previous layer's" results: random start values
        float v0[128];
        DH_UNROLL for (int i = 0; i < 128; ++i) {
            unsigned h = (tid * 131u + i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            v0[i] = (float)(h & 0xffffff) * (100.f / 16777216.f);
        }
#ifdef TC_BUILTIN
        DH_UNROLL for (int i = 0; i < 128; ++i) { A.s0[i / 16][i % 16] = v0[i]; A.s1[i / 16][i % 16] = 0.f; }
#else
        acc_write_all(v0);
#endif
    }
    Pieces bA[2], bB[2];
    EpiSt st;
    { unsigned h = (tid * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; st.lane_rnd = (float)(h & 0xffff) * (2.f / 65536.f) - 1.f; }
    AFrag a0[2], a1[2];
    for (int d = 0; d < DEPTH; ++d) ring_issue(R);
    ring_issue_one<0>(R); ring_issue_one<1>(R); ring_issue_one<2>(R);      // the steady state enters a k-step with 3 of 6 issued
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE * (DEPTH - 1) + 3) : "memory");
    asm volatile("s_barrier" ::: "memory");
    read_group<0>(a0, R.rd_addr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    _Pragma("unroll 1") for (int lp = 0; lp < layer_pairs; ++lp) {
        layer<128, EPI>(A, bA, bB, st, a0, a1, R);
        layer<0, EPI>(A, bA, bB, st, a0, a1, R);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    float sum = 0.f;
#ifdef TC_BUILTIN
    DH_UNROLL for (int m = 0; m < NM; ++m) DH_UNROLL for (int r = 0; r < 16; ++r) sum += A.s0[m][r];
#else
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    f32x16 x;
    stash_write<0>(R.stash); stash_read(x, R.stash);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    DH_UNROLL for (int r = 0; r < 16; ++r) sum += x[r];
#endif
    out[blockIdx.x * 256 + tid] = sum;
}

// ---------------------------------------------------------------- reference arm: the product's loop (as mfma_shape_micro.hip)
__global__ __launch_bounds__(256, 2) void kref(const bf16x8* __restrict__ wp, float* out, int layers, int tiles) {
    __shared__ __attribute__((aligned(16))) float X[TM * LDX];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < TM * LDX; i += 256) {
        unsigned h = (i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        X[i] = (float)(h & 0xffffff) * (1.f / 16777216.f);
    }
    __syncthreads();
    float sum = 0.f, lane_rnd;
    { unsigned h = (tid * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; lane_rnd = (float)(h & 0xffff) * (2.f / 65536.f) - 1.f; }
    for (int tl = 0; tl < tiles; ++tl)
        for (int l = 0; l < layers; ++l) {
            const bf16x8* wlay = wp + (size_t)(l % 8) * 3 * 16 * 8 * 64;
            f32x16 acc[MT][2];
            acc_zero(acc);
            gemm_rows_s(acc, X, LDX, 16, wlay, wave, lane);
            __syncthreads();
            DH_UNROLL for (int m = 0; m < MT; ++m)
                DH_UNROLL for (int t = 0; t < 2; ++t)
                    DH_UNROLL for (int q = 0; q < 16; ++q) {
                        const float v = softplus100(fmaf(acc[m][t][q], 1e-3f, lane_rnd + reg_rnd((m * 2 + t) * 16 + q)));
                        X[acc_row(m, q, lane) * LDX + acc_col(wave, t, lane)] = v;
                        sum += v;
                    }
            __syncthreads();
        }
    out[blockIdx.x * 256 + tid] = sum;
}

template <int EPI>
void run_t(const char* name, const bf16x8* wp, float* out, unsigned long long* clk, hipEvent_t a, hipEvent_t b, int reps) {
    const int lp = 32;          // 256 workgroups x 128 points, 64 layers each (= 8 tiles x 8 layers)
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(tchain<EPI>, dim3(256), dim3(256), 0, 0, wp, out, lp, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(tchain<EPI>, dim3(256), dim3(256), 0, 0, wp, out, lp, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = 2.0 * 128 * 256 * 256 * (2.0 * lp) * 256;
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < 256; ++i) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i] / (2.0 * lp)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    printf("%-44s %8.3f ms  %7.1f TFLOP/s fp32-equivalent   clock %.2f GHz, %.0f cycles per layer (768 MFMAs = 24576)\n", name, ms,
           flop / ms / 1e9, ghz[128], cyc[128]);
}

int main() {
    bf16x8* wp; float* out;
    const size_t nw = (size_t)8 * 3 * 16 * 8 * 64;
    hipMalloc(&wp, nw * sizeof(bf16x8));
    std::vector<unsigned short> h(nw * 8);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {          // random bf16 in +-[2^-7, 1)
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        h[i] = (unsigned short)(((r & 1) << 15) | ((0x78 + ((r >> 1) & 7)) << 7) | ((r >> 4) & 0x7f));
    }
    hipMemcpy(wp, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, (size_t)512 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 512 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 40;
    for (int round = 0; round < 3; ++round) {
        run_t<1>("register-resident chain (1 wave/SIMD)", wp, out, clk, a, b, reps);
        run_t<0>("  same, no epilogue (timing only)", wp, out, clk, a, b, reps);
        {   // product core: 512 workgroups x 64 points, 16 tiles x 8 layers
            const int layers = 8, tiles = 8;
            for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kref, dim3(512), dim3(256), 0, 0, wp, out, layers, tiles);
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kref, dim3(512), dim3(256), 0, 0, wp, out, layers, tiles);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
            const double flop = 2.0 * TM * 256 * 256 * layers * tiles * 512;
            printf("%-44s %8.3f ms  %7.1f TFLOP/s fp32-equivalent\n", "tile16.h core (LDS image, 2 WG/CU)", ms, flop / ms / 1e9);
        }
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
