// Round 4: does a TWO-piece fp16 split with THREE products per fp32 product beat the shipping three-piece bf16 split with six?
// (VERDICT r3 next #1.)  Both arms are the register-resident chain of dynhor_amd/csrc/chain_t.hip in its compiler-allocated form
// (tchain_micro.hip -DTC_BUILTIN), one binary, arms interleaved in one process, activations O(1) with random signs.
//
//   bf16x3 (shipping): x = x1 + x2 + x3 (bf16, exact residuals), products x1w1 x1w2 x2w1 x1w3 x2w2 x3w1          -> 6 MFMAs, 24 KB / stage
//   f16x2  (new)     : X = S x (S a power of two), X = Xh + Xl with Xh = f16(X), Xl = f16(X - Xh) (the UNSCALED residual: its
//                      absolute error is <= 2^-25 in units of X, i.e. fp32-class once S puts the operand's typical magnitude at
//                      >= 1); products Xh Wl, Xl Wh, Xh Wh                                                       -> 3 MFMAs, 16 KB / stage
//                      split per pair of values: v_cvt_pk_f16_f32, 2 x v_fma_mix_f32 (residual straight from the f16 halves),
//                      v_cvt_pk_f16_f32 = 2 vector ops per value (bf16x3: 4.5)
// Also here: gemm_check -- one 32 x 32 x 256 product per wave through the f16x2 pieces and v_mfma_f32_32x32x16_f16 against fp64 on
// the host, on operand classes whose low pieces are fp16 SUBNORMALS (does the matrix core flush them?), beside the bf16x3 pieces
// and the plain fp32 MFMA on the same data.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tchain2_micro.hip -o tchain2_micro
#include "../../dynhor_amd/csrc/tile16.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace dh;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int NM = 8;                        // 32-feature m-tiles of a 256-wide layer
constexpr int NSTAGE = 5, DEPTH = 3;         // ring slots / k-steps in flight

struct ArB3 {                                // three bf16 pieces, six products
    static constexpr int NPW = 3, NPX = 3, NPROD = 6;
    static constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct ArH2 {                                // two fp16 pieces, three products
    static constexpr int NPW = 2, NPX = 2, NPROD = 3;
    static constexpr int pw[3] = {1, 0, 0}, px[3] = {0, 1, 0};
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    }
};
template <class AR> constexpr int stage_bytes() { return NM * AR::NPW * 1024; }
template <class AR> constexpr int dma_per_wave() { return NM * AR::NPW / 4; }

template <class AR> struct Pieces { u32x4 p[AR::NPX]; };     // one k-step of the activation (B) operand
template <class AR> struct AFrag { u32x4 p[AR::NPW]; };      // one m-tile's weight (A) fragments of a k-step
struct EpiSt { f32x2 x, t, e, u; float lane_rnd; };
__host__ __device__ constexpr float reg_rnd(int r) { return (float)((r * 37 + 11) % 64) * (1.f / 64.f) - 0.5f; }

template <int OFF>
__device__ __forceinline__ u32x4 lds_b128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
__device__ __forceinline__ unsigned pack_f16x2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2)); }
// v - float(h.lo / h.hi): one v_fma_mix_f32 each, the f16 half read in place
__device__ __forceinline__ f32x2 resid_f16x2(f32x2 v, unsigned h) {
    f32x2 r;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(h), "v"(v[0]));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(h), "v"(v[1]));
    return r;
}
constexpr float XS = 16.f;                   // the f16x2 arm carries activations scaled by 16 (softplus folded: same op count)

// one micro-step (12 per pair of values) of the epilogue of an m-tile whose 16 finished values sit in x
template <class AR, int M, int STEP>
__device__ __forceinline__ void epi_step(const f32x16& x, Pieces<AR> (&out)[2], EpiSt& st) {
    constexpr int j = STEP / 12, s = STEP % 12, half = j / 4, q = j % 4, r0 = 2 * j;
    constexpr bool H = AR::NPROD == 3;
    constexpr float S = H ? XS : 1.f;
    // zs = S z;  S softplus(z) = max(zs, 0) + (S ln2 / beta) log2(1 + exp2(-|zs| beta log2e / S))
    if constexpr (s == 0) { st.x[0] = fmaf(x[r0], S * 1e-3f / (H ? XS : 1.f), S * (st.lane_rnd + reg_rnd(M * 16 + r0))); }
    else if constexpr (s == 1) { st.x[1] = fmaf(x[r0 + 1], S * 1e-3f / (H ? XS : 1.f), S * (st.lane_rnd + reg_rnd(M * 16 + r0 + 1))); }
    else if constexpr (s == 2) { st.t[0] = -fabsf(st.x[0]) * (SOFTPLUS_BETA * 1.44269504088896f / S); st.t[1] = -fabsf(st.x[1]) * (SOFTPLUS_BETA * 1.44269504088896f / S); }
    else if constexpr (s == 3) { st.e[0] = __builtin_amdgcn_exp2f(st.t[0]); st.e[1] = __builtin_amdgcn_exp2f(st.t[1]); }
    else if constexpr (s == 4) { st.e[0] = 1.f + st.e[0]; st.e[1] = 1.f + st.e[1]; }
    else if constexpr (s == 5) { st.e[0] = __builtin_amdgcn_logf(st.e[0]); st.e[1] = __builtin_amdgcn_logf(st.e[1]); }
    else if constexpr (s == 6) { st.t[0] = fmaxf(st.x[0], 0.f); st.t[1] = fmaxf(st.x[1], 0.f); }
    else if constexpr (s == 7) { st.x[0] = fmaf(st.e[0], S * 0.69314718055995f / SOFTPLUS_BETA, st.t[0]); st.x[1] = fmaf(st.e[1], S * 0.69314718055995f / SOFTPLUS_BETA, st.t[1]); }
    else if constexpr (!H) {
        if constexpr (s == 8) { const unsigned h = pack_bf16x2(st.x); out[half].p[0][q] = h; st.u = unpack_bf16x2(h); }
        else if constexpr (s == 9) { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; }
        else if constexpr (s == 10) { const unsigned h = pack_bf16x2(st.x); out[half].p[1][q] = h; st.u = unpack_bf16x2(h); }
        else { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; out[half].p[2][q] = pack_bf16x2(st.x); }
    } else {
        if constexpr (s == 8) { const unsigned h = pack_f16x2(st.x); out[half].p[0][q] = h; st.u = resid_f16x2(st.x, h); }
        else if constexpr (s == 9) { out[half].p[1][q] = pack_f16x2(st.u); }
        else if constexpr (s == 10) { st.t[0] = st.x[0] * (1.f / XS); st.t[1] = st.x[1] * (1.f / XS); asm volatile("" ::"v"(st.t[0]), "v"(st.t[1])); }   // what a saved tile costs
    }
}
template <class AR, int M, int I, int N>
__device__ __forceinline__ void epi_only(const f32x16& x, Pieces<AR> (&out)[2], EpiSt& st) {
    if constexpr (I < N) { epi_step<AR, M, I>(x, out, st); epi_only<AR, M, I + 1, N>(x, out, st); }
}

struct Acc { f32x16 s0[NM], s1[NM]; };
// MFMA I of group G (m-tiles 2G, 2G+1) into set NB: product-major, consecutive MFMAs hit different accumulators
template <class AR, int NB, int G, int I>
__device__ __forceinline__ void mfma_step(Acc& A, const AFrag<AR> (&a)[2], const Pieces<AR>& b) {
    constexpr int p = I / 2, t = I % 2, mt = 2 * G + t;
    f32x16& acc = NB ? A.s1[mt] : A.s0[mt];
    acc = AR::mfma(a[t].p[AR::pw[p]], b.p[AR::px[p]], acc);
}
struct Ring {
    unsigned rd_addr, rd_slot, is_slot, is_goff, gbytes;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lds_base;
    char* lds;
    int wave, lane;
};
template <class AR, int I>
__device__ __forceinline__ void ring_issue_one(Ring& R) {
    const unsigned frag = R.wave + 4 * I;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.lds + R.is_slot * stage_bytes<AR>() + frag * 1024),
                                             16, R.lane * 16, R.is_goff + frag * 1024, 0, 0);
    if constexpr (I == dma_per_wave<AR>() - 1) {
        R.is_slot = (R.is_slot + 1 == NSTAGE) ? 0 : R.is_slot + 1;
        R.is_goff = (R.is_goff + stage_bytes<AR>() == R.gbytes) ? 0 : R.is_goff + stage_bytes<AR>();
    }
}
template <class AR, int I = 0>
__device__ __forceinline__ void ring_issue(Ring& R) {
    if constexpr (I < dma_per_wave<AR>()) { ring_issue_one<AR, I>(R); ring_issue<AR, I + 1>(R); }
}
template <class AR, int I = 0>
__device__ __forceinline__ void ring_issue_half(Ring& R) {
    if constexpr (I < dma_per_wave<AR>() / 2) { ring_issue_one<AR, I>(R); ring_issue_half<AR, I + 1>(R); }
}
template <class AR>
__device__ __forceinline__ void ring_advance_read(Ring& R) {
    R.rd_slot = (R.rd_slot + 1 == NSTAGE) ? 0 : R.rd_slot + 1;
    R.rd_addr = R.lds_base + R.rd_slot * stage_bytes<AR>() + R.lane * 16;
}
// the MFMAs of a group, each followed by its share of the dealt epilogue (EPS micro-steps per MFMA) and, after MFMA DAI / DBI,
// the LDS-DMA piece DA / DB (-1: none)
// timing-only switches (wrong results, same instruction shape otherwise): what the LDS-DMA issue and the barrier cost
#ifndef TC_NODMA
#define TC_NODMA 0
#endif
#ifndef TC_NOBAR
#define TC_NOBAR 0
#endif
template <class AR, int NB, int G, int EM, int EBASE, int DA, int DB, int I>
__device__ __forceinline__ void group_steps(Acc& A, const AFrag<AR> (&a)[2], const Pieces<AR>& b, const f32x16& xE, Pieces<AR> (&bn)[2], EpiSt& st, Ring& R) {
    constexpr int NMF = 2 * AR::NPROD, EPS = 12 / NMF;
    constexpr int DAI = AR::NPROD == 6 ? 3 : 2, DBI = 9;
    if constexpr (I < NMF) {
        mfma_step<AR, NB, G, I>(A, a, b);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!TC_NODMA && I == DAI && DA >= 0) { ring_issue_one<AR, DA>(R); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (!TC_NODMA && I == DBI && DB >= 0) { ring_issue_one<AR, DB>(R); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (EM >= 0 && EM < NM) {
            epi_step<AR, EM, EBASE + EPS * I>(xE, bn, st);
            if constexpr (EPS == 2) epi_step<AR, EM, EBASE + EPS * I + 1>(xE, bn, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        group_steps<AR, NB, G, EM, EBASE, DA, DB, I + 1>(A, a, b, xE, bn, st, R);
    }
}
template <class AR, int G>
__device__ __forceinline__ void read_group(AFrag<AR> (&a)[2], unsigned addr) {
    DH_UNROLL for (int t = 0; t < 2; ++t) {
        a[t].p[0] = lds_b128<(2 * G * AR::NPW + 0) * 1024>(addr + t * AR::NPW * 1024);
        a[t].p[1] = lds_b128<(2 * G * AR::NPW + 1) * 1024>(addr + t * AR::NPW * 1024);
        if constexpr (AR::NPW == 3) a[t].p[2] = lds_b128<(2 * G * AR::NPW + 2) * 1024>(addr + t * AR::NPW * 1024);
    }
}

// one k-step into accumulator set NB: entering, a0 holds group 0's fragments; leaving, a0 holds group 0 of the NEXT k-step
template <class AR, int NB, int EM, int EH>
__device__ __forceinline__ void kstep(Acc& A, const Pieces<AR>& b, const f32x16& xE, Pieces<AR> (&bn)[2], EpiSt& st, AFrag<AR> (&a0)[2], AFrag<AR> (&a1)[2], Ring& R) {
    constexpr bool B3 = AR::NPROD == 6;
    // LDS-DMA pieces of the stage being issued: bf16x3 6 per wave and k-step (3, 4 | 5 | barrier | 0, 1 | 2), f16x2 4 (2 | 3 | barrier | 0 | 1)
    read_group<AR, 1>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<AR, NB, 0, EM, EH * 48 + 0, B3 ? 3 : 2, B3 ? 4 : -1, 0>(A, a0, b, xE, bn, st, R);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_group<AR, 2>(a0, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<AR, NB, 1, EM, EH * 48 + 12, B3 ? 5 : 3, -1, 0>(A, a1, b, xE, bn, st, R);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!TC_NODMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dma_per_wave<AR>() * (DEPTH - 1)) : "memory");
    if constexpr (!TC_NOBAR) asm volatile("s_barrier" ::: "memory");
    read_group<AR, 3>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<AR, NB, 2, EM, EH * 48 + 24, 0, B3 ? 1 : -1, 0>(A, a0, b, xE, bn, st, R);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ring_advance_read<AR>(R);
    read_group<AR, 0>(a0, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    group_steps<AR, NB, 3, EM, EH * 48 + 36, B3 ? 2 : 1, -1, 0>(A, a1, b, xE, bn, st, R);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <class AR, int NB, int M, int EPI>
__device__ __forceinline__ void mpair(Acc& A, Pieces<AR> (&bA)[2], Pieces<AR> (&bB)[2], EpiSt& st, AFrag<AR> (&a0)[2], AFrag<AR> (&a1)[2], Ring& R) {
    if constexpr (M < NM) {
        constexpr int EM = EPI ? M + 1 : -1;
        const f32x16& xO = (NB ? A.s0 : A.s1)[M + 1 < NM ? M + 1 : 0];
        if constexpr (M % 2 == 0 || !EPI) {
            kstep<AR, NB, EM, 0>(A, bA[0], xO, bB, st, a0, a1, R);
            kstep<AR, NB, EM, 1>(A, bA[1], xO, bB, st, a0, a1, R);
        } else {
            kstep<AR, NB, EM, 0>(A, bB[0], xO, bA, st, a0, a1, R);
            kstep<AR, NB, EM, 1>(A, bB[1], xO, bA, st, a0, a1, R);
        }
        mpair<AR, NB, M + 1, EPI>(A, bA, bB, st, a0, a1, R);
    }
}
template <class AR, int NB, int EPI>
__device__ __forceinline__ void layer(Acc& A, Pieces<AR> (&bA)[2], Pieces<AR> (&bB)[2], EpiSt& st, AFrag<AR> (&a0)[2], AFrag<AR> (&a1)[2], Ring& R) {
    DH_UNROLL for (int m = 0; m < NM; ++m) DH_UNROLL for (int r = 0; r < 16; ++r) (NB ? A.s1 : A.s0)[m][r] = 0.f;
    epi_only<AR, 0, 0, 96>((NB ? A.s0 : A.s1)[0], bA, st);
    if constexpr (!EPI) { bB[0] = bA[0]; bB[1] = bA[1]; }
    mpair<AR, NB, 0, EPI>(A, bA, bB, st, a0, a1, R);
    if constexpr (!EPI) {          // rule 17: without the dealt epilogue nothing reads m-tiles 1..7 -- keep their MFMAs from being deleted
        DH_UNROLL for (int m = 0; m < NM; ++m) asm volatile("" ::"v"((NB ? A.s1 : A.s0)[m]));
    }
}

// EPI = 0: timing-only arm -- no dealt epilogue (the pieces of m-tile 0 feed every k-step): MFMA + weight ring alone
template <class AR, int EPI>
__global__ __launch_bounds__(256, 1) void tchain(const void* __restrict__ wp, float* out, int layer_pairs, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) char lds[NSTAGE * stage_bytes<AR>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Ring R;
    R.lds = lds; R.lds_base = (unsigned)(uintptr_t)lds; R.wave = wave; R.lane = lane;
    R.gbytes = 8u * 16u * stage_bytes<AR>();
    R.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wp), 0, R.gbytes, 0x00020000);
    R.is_goff = 0; R.is_slot = 0; R.rd_slot = 0; R.rd_addr = R.lds_base + lane * 16;
    Acc A;
    DH_UNROLL for (int i = 0; i < 128; ++i) {
        unsigned h = (tid * 131u + i + blockIdx.x * 7919u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        A.s0[i / 16][i % 16] = (float)(h & 0xffffff) * (100.f / 16777216.f); A.s1[i / 16][i % 16] = 0.f;
    }
    Pieces<AR> bA[2], bB[2];
    EpiSt st;
    { unsigned h = (tid * 2654435761u) ^ (blockIdx.x * 40503u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; st.lane_rnd = (float)(h & 0xffff) * (2.f / 65536.f) - 1.f; }
    AFrag<AR> a0[2], a1[2];
    for (int d = 0; d < DEPTH; ++d) ring_issue<AR>(R);
    ring_issue_half<AR>(R);                       // the steady state enters a k-step with half a stage issued
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(dma_per_wave<AR>() * (DEPTH - 1) + dma_per_wave<AR>() / 2) : "memory");
    asm volatile("s_barrier" ::: "memory");
    read_group<AR, 0>(a0, R.rd_addr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    _Pragma("unroll 1") for (int lp = 0; lp < layer_pairs; ++lp) {
        layer<AR, 1, EPI>(A, bA, bB, st, a0, a1, R);
        layer<AR, 0, EPI>(A, bA, bB, st, a0, a1, R);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
    float sum = 0.f;
    DH_UNROLL for (int m = 0; m < NM; ++m) DH_UNROLL for (int r = 0; r < 16; ++r) sum += A.s0[m][r];
    out[blockIdx.x * 256 + tid] = sum;
}

template <class AR, int EPI>
double run_t(const char* name, const void* wp, float* out, unsigned long long* clk, hipEvent_t a, hipEvent_t b, int reps) {
    const int lp = 32;          // 256 workgroups x 128 points, 64 layers each
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((tchain<AR, EPI>), dim3(256), dim3(256), 0, 0, wp, out, lp, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((tchain<AR, EPI>), dim3(256), dim3(256), 0, 0, wp, out, lp, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = 2.0 * 128 * 256 * 256 * (2.0 * lp) * 256;
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz, cyc;
    for (int i = 0; i < 256; ++i) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); cyc.push_back((double)h[2 * i] / (2.0 * lp)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    printf("{\"arm\": \"%s\", \"ms\": %.3f, \"tflops_fp32_equiv\": %.1f, \"clock_ghz\": %.2f, \"cycles_per_layer\": %.0f, \"mfma_per_layer\": %d}\n", name, ms,
           flop / ms / 1e9, ghz[128], cyc[128], 128 * AR::NPROD);
    return flop / ms / 1e9;
}

// ---------------------------------------------------------------------------------------------------------------- accuracy
// out[32 x 32] (one wave) = A[32 x 256] * B[256 x 32] with pieces formed on the device exactly as the chains form them.
// MODE 0: f16x2, 1: bf16x3, 2: fp32 MFMA.  sa / sb: power-of-two operand scales of the f16x2 arm.
template <int MODE>
__global__ void gemm_check(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ out, float sa, float sb) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* Ab = A + (size_t)blockIdx.x * 32 * 256;
    f32x16 acc;
    DH_UNROLL for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int ks = 0; ks < 16; ++ks) {
        float av[8], bv[8];
        DH_UNROLL for (int j = 0; j < 8; ++j) { av[j] = Ab[r * 256 + 16 * ks + 8 * h + j]; bv[j] = B[(16 * ks + 8 * h + j) * 32 + r]; }
        if constexpr (MODE == 0) {
            u32x4 ah, al, bh, bl;
            DH_UNROLL for (int q = 0; q < 4; ++q) {
                f32x2 x = {av[2 * q] * sa, av[2 * q + 1] * sa}, y = {bv[2 * q] * sb, bv[2 * q + 1] * sb};
                ah[q] = pack_f16x2(x); al[q] = pack_f16x2(resid_f16x2(x, ah[q]));
                bh[q] = pack_f16x2(y); bl[q] = pack_f16x2(resid_f16x2(y, bh[q]));
            }
            acc = ArH2::mfma(ah, bl, acc); acc = ArH2::mfma(al, bh, acc); acc = ArH2::mfma(ah, bh, acc);
        } else if constexpr (MODE == 1) {
            const Bf3 a = split3(f32x4{av[0], av[1], av[2], av[3]}, f32x4{av[4], av[5], av[6], av[7]});
            const Bf3 b = split3(f32x4{bv[0], bv[1], bv[2], bv[3]}, f32x4{bv[4], bv[5], bv[6], bv[7]});
            acc = mfma6(a, b, acc);
        } else {
            (void)av; (void)bv;
            DH_UNROLL for (int j = 0; j < 8; ++j) {          // 32x32x2: lane (r, h) holds A[r][k0 + h], B[k0 + h][r]
                const int k0 = 16 * ks + 2 * j;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ab[r * 256 + k0 + h], B[(k0 + h) * 32 + r], acc, 0, 0, 0);
            }
        }
    }
    const float inv = MODE == 0 ? 1.f / (sa * sb) : 1.f;
    DH_UNROLL for (int i = 0; i < 16; ++i) out[(size_t)blockIdx.x * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i] * inv;
}

static double rel_err(const std::vector<float>& got, const std::vector<double>& ref) {
    double num = 0, den = 0;
    for (size_t i = 0; i < ref.size(); ++i) { const double d = got[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
    return std::sqrt(num / den);
}
static void accuracy(const char* cls, float amag, float spread, float sa, float sb) {
    const int NB = 128;
    std::vector<float> A((size_t)NB * 32 * 256), B(256 * 32);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffffff) * (2.f / 16777216.f) - 1.f; };
    for (size_t i = 0; i < A.size(); ++i) {
        const float z = rnd() * 0.6f;                                      // softplus(100 z)-like: half ~0, rest spread
        const float act = z > 0.f ? z : std::log1p(std::exp(100.f * z)) / 100.f;
        A[i] = act * amag * std::exp(spread * rnd());
    }
    for (auto& b : B) b = rnd() * 0.25f;
    std::vector<double> ref((size_t)NB * 1024);
    for (int blk = 0; blk < NB; ++blk)
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double acc = 0;
            for (int k = 0; k < 256; ++k) acc += (double)A[((size_t)blk * 32 + i) * 256 + k] * (double)B[k * 32 + j];
            ref[(size_t)blk * 1024 + i * 32 + j] = acc;
        }
    float *dA, *dB, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, ref.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> got(ref.size());
    double e[3];
    hipLaunchKernelGGL(gemm_check<0>, dim3(NB), dim3(64), 0, 0, dA, dB, dO, sa, sb);
    hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); e[0] = rel_err(got, ref);
    hipLaunchKernelGGL(gemm_check<1>, dim3(NB), dim3(64), 0, 0, dA, dB, dO, 1.f, 1.f);
    hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); e[1] = rel_err(got, ref);
    hipLaunchKernelGGL(gemm_check<2>, dim3(NB), dim3(64), 0, 0, dA, dB, dO, 1.f, 1.f);
    hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost); e[2] = rel_err(got, ref);
    printf("{\"accuracy\": \"%s\", \"scale_a\": %g, \"scale_b\": %g, \"rel_l2_f16x2\": %.3e, \"rel_l2_bf16x3\": %.3e, \"rel_l2_fp32_mfma\": %.3e}\n", cls, sa, sb, e[0], e[1], e[2]);
    hipFree(dA); hipFree(dB); hipFree(dO);
}

template <class AR>
void* make_weights() {
    const size_t nbytes = (size_t)8 * 16 * stage_bytes<AR>();
    std::vector<unsigned short> h(nbytes / 2);
    unsigned s = 12345u;
    for (size_t i = 0; i < h.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = s >> 8;
        if (AR::NPROD == 6) h[i] = (unsigned short)(((r & 1) << 15) | ((0x78 + ((r >> 1) & 7)) << 7) | ((r >> 4) & 0x7f));    // bf16 +-[2^-7, 1)
        else {
            // fp16 pieces as the packer would make them from 16 w, w in +-[2^-7, 1): high piece +-[2^-3, 16), low piece (the
            // unscaled residual) 2^-12 of it
            const bool lowp = ((i / 8 / 64) % 2) == 1;
            const unsigned e = (lowp ? 0 : 12) + ((r >> 1) & 7);
            h[i] = (unsigned short)(((r & 1) << 15) | (e << 10) | ((r >> 4) & 0x3ff));
        }
    }
    void* d; hipMalloc(&d, nbytes); hipMemcpy(d, h.data(), nbytes, hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char**) {
    if (argc > 1) goto timing;           // any argument: skip the accuracy part
    accuracy("activations O(1), S_x = 1", 1.f, 0.f, 1.f, 16.f);
    accuracy("activations O(1), S_x = 16", 1.f, 0.f, 16.f, 16.f);
    accuracy("activations O(1), weights unscaled too", 1.f, 0.f, 1.f, 1.f);
    accuracy("adjoints 1e-6, unscaled", 1e-6f, 0.f, 1.f, 16.f);
    accuracy("adjoints 1e-6, scaled to 2^9", 1e-6f, 0.f, 536870912.f, 16.f);
    accuracy("adjoints 1e-6 x e^(+-4), scaled to 2^9", 1e-6f, 4.f, 8388608.f, 16.f);
    accuracy("values 1e3, scaled to 2^9", 1e3f, 0.f, 0.5f, 16.f);
timing:
    void* w3 = make_weights<ArB3>();
    void* w2 = make_weights<ArH2>();
    float* out; hipMalloc(&out, (size_t)512 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 512 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 40;
    for (int round = 0; round < 3; ++round) {
        run_t<ArB3, 1>("bf16x3 chain (shipping core)", w3, out, clk, a, b, reps);
        run_t<ArH2, 1>("f16x2 chain", w2, out, clk, a, b, reps);
        run_t<ArB3, 0>("bf16x3 bare MFMA + ring (accumulators kept alive)", w3, out, clk, a, b, reps);
        run_t<ArH2, 0>("f16x2 bare MFMA + ring (accumulators kept alive)", w2, out, clk, a, b, reps);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return 0;
}
