// Register-resident ("transposed") SDF forward chains: sdf_nograd_t_kernel (sdf only) and sdf_fwd_train_t_kernel (sdf + every tile
// the backward pass needs)   (round 3; scripts/micro/tchain_micro.hip is the timing study they are built from,
// profiles/r03_ab_register_resident_chain.json its record, profiles/r03_ab_chain_t.json the A/B against the tile16.h kernels).
//
//   out^T[feature][point] = W[feature][k] * act^T[k][point]: the WEIGHTS are the MFMA A operand, the activations the B operand.  A
//   wave owns 32 points (the MFMA's column axis = its lane & 31) and ALL 256 features of a layer: 8 accumulators (m-tiles) of
//   32 x 32.  Accumulator register r of m-tile M on lane (p, h = lane >> 5) holds feature 32 M + 8 (r / 4) + 4 h + (r % 4) of
//   point p -- which IS the B-operand layout of the next layer's k-steps 2 M (registers 0..7) and 2 M + 1 (registers 8..15) when
//   the next layer's weights are packed with their k axis permuted to match (pack.hip, PackJob.rev == 2):
//       hardware k slot (h, i) of k-step s  <->  input feature 16 s + 8 (i / 4) + 4 h + (i % 4).
//   Activations therefore never leave registers: bias + softplus + ONE 3-way bf16 split per value; no LDS image, no hand-off
//   barrier, no redundant split.  Two accumulator sets (256 registers, one wave per SIMD): layer l accumulates into set l & 1
//   while the epilogue of m-tile M + 1 of layer l - 1 is dealt, one micro-step per MFMA, under k-steps 2 M, 2 M + 1 of layer l
//   (which need only m-tile M).  Only m-tile 0's epilogue is exposed.
//   Weights enter the CU once per 128 points: an LDS ring of 24 KB stages (one 16-deep k-step of all 256 output features, three
//   bf16 pieces) filled by LDS-DMA (buffer_load ... lds) dealt singly between MFMAs, one raw s_barrier per k-step; the stages
//   of the network (3 + 16 + 16 + 16 + (14 + 3) + 16 + 16 + 16 = 116; + 16 of lin8's rows 1..256 in the training kernel) are
//   one cyclic stream, so the ring never drains between layers or tiles.  The per-layer biases and lin8's row 0 sit in LDS
//   (10 KB); the embedding image of a wave's 32 points stays in LDS for the skip layer.
//   The training kernel writes the tiles of sdf_fwd_train_kernel (tile.h native layout, 64 points): an m-tile's activations are
//   transposed through a wave-private LDS patch (lane-per-point -> four consecutive points per lane) on their way out.
//   (Round 6, the fp16 form: the transposition runs on the matrix pipe instead -- T_SAVE_MFMA below -- and only the feature tile
//   still crosses the patch.)
// Arithmetic (template parameter AR): TArB3 = the same six bf16 products per fp32 product, smallest terms first, as tile16.h;
// TArH2 (round 4) = tile16h.h's two fp16 pieces / three products: a stage is 16 KB (two pieces), the chain carries its activations
// scaled by H2_XS = 16 (bias rows pre-scaled, softplus evaluated in the scaled variable: same operation count, bit-identical
// values after the exact division by 16 on the way to a saved tile -- of which the T_SAVE_MFMA form saves the two-piece sum hi + lo,
// 2^-22 apart), the weights by the linear's power of two (its reciprocal
// multiplies the accumulator in the epilogue's first fma), and the dealt epilogue runs two micro-steps per MFMA.
#include "mlp_common.h"
#include "tile16h.h"
#include "kernels.h"
#include "workspace.h"

namespace dh {

constexpr int T_NM = 8;                          // 32-feature m-tiles of a 256-wide layer
constexpr int T_PTS = 128;                       // points per workgroup tile
struct TArB3 {                                   // three bf16 pieces, six products
    static constexpr bool H = false;
    static constexpr int NP = 3, NPROD = 6;
    static constexpr int STAGE_BYTES = T_NM * 3 * 1024;      // one k-step of weight pieces: 8 m-tiles x 3 pieces x 1 KiB fragments
    static constexpr int DMA = T_NM * 3 / 4;                 // 6 LDS-DMA instructions per wave and k-step
    static constexpr int BIAS_BYTES = 10 * 1024;             // layout.h PACKT.bias10
    static constexpr int pw[6] = {2, 1, 0, 1, 0, 0}, px[6] = {0, 1, 2, 0, 1, 0};      // smallest terms first (tile16.h mfma6)
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
struct TArH2 {                                   // two fp16 pieces, three products (tile16h.h)
    static constexpr bool H = true;
    static constexpr int NP = 2, NPROD = 3;
    static constexpr int STAGE_BYTES = T_NM * 2 * 1024;
    static constexpr int DMA = T_NM * 2 / 4;                 // 4
    static constexpr int BIAS_BYTES = 11 * 1024;             // layout.h PACKH.bias11
    static constexpr int pw[3] = {1, 0, 0}, px[3] = {0, 1, 0};                        // w_lo x_hi, w_hi x_lo, w_hi x_hi
    static __device__ __forceinline__ f32x16 mfma(const u32x4& a, const u32x4& b, const f32x16& c) { return mfma_h(a, b, c); }
};
constexpr int T_EMB_LD = 52;                     // floats per point of the embedding image (48 + pad: b128 reads of 16 rows conflict free)
constexpr int T_EMB_BYTES = 32 * T_EMB_LD * 4;   // per wave: its 32 points x [39 embedding values, zero padded to 48]
// How the training kernel's fp16 form turns an m-tile (lane-per-point) into a native tile (lane-per-feature).  1 (round 6): on the
// MATRIX pipe -- the m-tile's finished pieces (the next layer's B operand: lane = point, slots = features) are fed as the A operand
// of four more MFMAs against a constant selector (1/16 where slot and output column name the same feature): D[point][feature] =
// (hi + lo) / 16 in accumulator layout, which IS the native tile's layout; no LDS patch, no vector instruction.  The saved value is
// the activation the next layer consumed (hi + lo: the fp32 activation to 2^-22, each product and both adds exact), and a consumer
// that splits it again gets the same two pieces back.  0: through the wave-private LDS patch (16 ds_write_b32 + 4 ds_read_b128 per
// m-tile; rounds 3-5; bf16x3 always) -- the fp32 activation itself.
#ifndef T_SAVE_MFMA
#define T_SAVE_MFMA 1
#endif
#ifndef T_TILE_ST_AUX              // cache policy of the T_SAVE_MFMA form's tile stores (2 = nt, as DH_TILE_ST; development: A/B of the others)
#define T_TILE_ST_AUX 2
#endif
constexpr int T_PATCH_LD = 40;                   // floats per feature row of the transposition patch (4 * 40 % 64 == 32: the two half-waves' writes hit different banks)
constexpr int T_PATCH_BYTES = 32 * T_PATCH_LD * 4;
static_assert(T_STREAM_STAGES_TRAIN * TArB3::STAGE_BYTES == PACKT_STREAM_FLOATS * 4, "layout.h PACKT");
static_assert(T_STREAM_STAGES_TRAIN * TArH2::STAGE_BYTES == PACKTH_STREAM_FLOATS * 4, "layout.h PACKH");

// NSTAGE ring slots / DEPTH k-steps in flight (NSTAGE >= DEPTH + 2: the barrier sits mid-step); STAGES: length of the cyclic
// weight stream (the no-grad kernel stops before lin8's rows 1..256)
// (development: -DT_NG_DEPTH / -DT_TR_DEPTH_H re-size the rings of the no-grad kernel / the fp16 training kernel, NSTAGE = DEPTH + 2;
// profiles/r05_ab_chain_t_ring.txt: depths 2 / 3 / 4 time the same within 1 %)
#ifndef T_NG_DEPTH
#define T_NG_DEPTH 3
#endif
#ifndef T_TR_DEPTH_H
#define T_TR_DEPTH_H 3
#endif
template <class AR_> struct TCfgNoGrad { using AR = AR_; static constexpr bool TRAIN = false; static constexpr int NSTAGE = AR_::H ? T_NG_DEPTH + 2 : 5, DEPTH = AR_::H ? T_NG_DEPTH : 3, STAGES = T_STREAM_STAGES_NOGRAD; };
template <class AR_> struct TCfgTrain { using AR = AR_; static constexpr bool TRAIN = true; static constexpr int NSTAGE = AR_::H ? T_TR_DEPTH_H + 2 : 4, DEPTH = AR_::H ? T_TR_DEPTH_H : 2, STAGES = T_STREAM_STAGES_TRAIN; };
template <class C> constexpr int t_lds_bytes() { return C::NSTAGE * C::AR::STAGE_BYTES + C::AR::BIAS_BYTES + 4 * T_EMB_BYTES + (C::TRAIN ? 4 * T_PATCH_BYTES : 0); }

template <class AR> struct TPieces { u32x4 p[AR::NP]; };      // one k-step of the activation (B) operand
template <class AR> struct TFrag { u32x4 p[AR::NP]; };        // one m-tile's weight (A) fragments of a k-step
struct TAcc { f32x16 s[2][T_NM]; };
// epilogue state.  b = the pair's biases (the next pair's are read into the same registers right after their last use).  Training
// kernel only: patch_wr = the lane's write address in the patch, hd = pair 0's activations on their way to the patch
// (TArH2: isw = 1 / S_w of the layer whose accumulators the epilogue reads)
// (DOT -- the training kernel's lin8 layer, T_SAVE_MFMA form: w = the pair's weights of lin8's row 0, s0 / s1 = the sdf's two partial sums)
struct TEpi { f32x2 x, t, e, u, b, hd, w; unsigned bias_addr, patch_wr; float isw, s0, s1; };
// saving an m-tile as a native tile (tile.h): patch_rd = the lane's read address in the patch, v = two float4 in flight, rsrc =
// buffer descriptor of the native tile (wave-uniform; zero records when the tile does not exist -- ragged last tile -- so the
// hardware drops the stores), loff = the lane's byte offset in it (its half of the tile and its lane slot)
// (T_SAVE_MFMA form: d = the transposed m-tile, sel = the selector fragments of the m-tile's two k-steps; v / patch_rd unused)
struct TSave { f32x4 v[2]; unsigned patch_rd, loff; __amdgpu_buffer_rsrc_t rsrc; f32x16 d; u32x4 sel[2]; };
template <class C> constexpr bool t_save_mfma() { return C::TRAIN && C::AR::H && T_SAVE_MFMA; }
template <class C> constexpr bool t_save_patch() { return C::TRAIN && !t_save_mfma<C>(); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t t_tile_rsrc(const void* tile, bool ok, int bytes) {
#ifdef T_PROBE_DROP_STORES                               // timing probe: zero records, the hardware drops every tile store (same instructions)
    ok = false;
#endif
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(tile), 0, ok ? bytes : 0, 0x00020000);
}

// Every LDS access of this kernel is inline asm: an LDS access the compiler can see makes it wait for ALL outstanding LDS-DMA
// (vmcnt(0): it cannot prove the access does not alias a DMA's destination), which would drain the ring.  The "_w" forms
// complete before they return (exposed code: the compiler may move other instructions across a separate s_waitcnt statement);
// the others are waited for by the lgkmcnt(0) that ends every 12-MFMA group.  An asm output nobody reads is given a register
// that holds something else by the time the data lands: never issue a read whose result is not used
// (scripts/isa_inflight_check.py scans the listing for both mistakes; tests/test_cpu_isa_inflight.py).
template <int OFF>
__device__ __forceinline__ u32x4 t_lds_b128(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void t_lds_read(f32x4& v, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF)); }
template <int OFF>
__device__ __forceinline__ f32x4 t_lds_read_w(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// four float4 at byte offsets O0..O3 from addr, complete when the statement ends (one LDS latency for the four)
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ void t_lds_read4_w(f32x4& a, f32x4& b, f32x4& c, f32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %4 offset:%7\n\tds_read_b128 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr), "n"(O0), "n"(O1), "n"(O2), "n"(O3) : "memory");
}
// four floats likewise
template <int O0, int O1, int O2, int O3>
__device__ __forceinline__ f32x4 t_lds_read32x4_w(unsigned addr) {
    float a, b, c, d;
    asm volatile("ds_read_b32 %0, %4 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7\n\tds_read_b32 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr), "n"(O0), "n"(O1), "n"(O2), "n"(O3) : "memory");
    return f32x4{a, b, c, d};
}
template <int OFF>
__device__ __forceinline__ float t_lds_read32_w(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ void t_lds_write_b32(unsigned addr, float v) {
#ifdef T_PROBE_NO_PATCH_WRITES                           // timing probe (results wrong): the save path without its 16 ds_write_b32 per m-tile
    asm volatile("" ::"v"(addr), "v"(v));
#else
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
#endif
}
// bias pair of (m-tile M, pair J): features 32 M + 8 (J / 2) + 2 (J % 2) + 4 h, +1   (addr carries the row and 16 h)
template <int M, int J, int ROWOFF = 0>
__device__ __forceinline__ void t_bias_read(f32x2& dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(ROWOFF + (32 * M + 8 * (J / 2) + 2 * (J % 2)) * 4));
}
template <int M, int J, int ROWOFF = 0>
__device__ __forceinline__ void t_bias_read_w(f32x2& dst, unsigned addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(dst) : "v"(addr), "n"(ROWOFF + (32 * M + 8 * (J / 2) + 2 * (J % 2)) * 4) : "memory");
}

// one micro-step of the epilogue of m-tile M whose 16 finished values are x: pair j = STEP / 12 = values 2j, 2j+1 -> u32 j % 4 of
// the pieces of k-step half j / 4.  12 steps of 2-4 vector ops: + bias, -|x| c, exp2, 1 + e, log2, max, fma, then the 3-way split.
// Step 1 also issues the bias read of the next pair; after the last pair that of the next m-tile's first pair -- only if NEXT
// says that m-tile's epilogue will run.  (m-tile 0's epilogue runs outside the MFMA stream: t_epi_exposed.)  SAVE (training
// kernel): the pair's activations also go to the transposition patch.
// TArH2's micro-steps, two per MFMA.  The pair lives in the scaled variable zs = 16 z:  16 softplus(z) = max(zs, 0) +
// (16 ln2 / beta) log2(1 + exp2(-|zs| beta log2e / 16)); the accumulator carries 16 S_w (W x), the bias row 16 b.  Split: convert,
// two residuals, convert.  SAVE: the activation itself (x 1/16, exact) goes to the patch in steps 10 / 11 -- behind the last read
// of the previous m-tile's patch (group 1, MFMA slots 2-3 = pair 1's steps 4..7); pair 0's waits in hd until pair 1's steps 8 / 9.
// How many patch writes (ds_write_b32) micro-step s of pair j issues in the SAVE forms -- the ONE table behind both the emitters
// below and the counted wait that ends a group (t_late_writes / t_group_wait: a hand-copied tally there was ADVICE r5's finding).
// H: the two-piece fp16 chain (pair 0's two writes deferred to pair 1's steps 8 / 9, a pair's own in steps 10 / 11); B: bf16x3
// (deferred in steps 4 / 5, own in steps 7 / 8).  The last LDS READ a pair issues is the bias read of step 1: every write comes later.
constexpr int t_epi_writes_h(int j, int s) { return (s == 8 || s == 9) ? (j == 1) : (s == 10 || s == 11) ? (j > 0) : 0; }
constexpr int t_epi_writes_b(int j, int s) { return (s == 4 || s == 5) ? (j == 1) : (s == 7 || s == 8) ? (j > 0) : 0; }
template <bool H>
constexpr int t_epi_writes(int j, int s) { return H ? t_epi_writes_h(j, s) : t_epi_writes_b(j, s); }
template <bool H>
constexpr int t_epi_pair_writes(int j) {                 // all writes of pair j; none may precede the pair's last LDS read (step 1)
    int n = 0;
    for (int s = 2; s < 12; ++s) n += t_epi_writes<H>(j, s);
    return n;
}
static_assert(t_epi_writes<true>(1, 0) + t_epi_writes<true>(1, 1) + t_epi_writes<false>(1, 0) + t_epi_writes<false>(1, 1) == 0, "a patch write before the bias read");
// DOT: the layer is lin8 of the training kernel -- its epilogue's activations are softplus(lin7), and lin8's row 0 (the sdf; bias
// table row 8, one row behind lin7's biases) is formed on them as they pass: even values into s0, odd into s1, m-tiles and values
// ascending -- t_final_dot's order, bit for bit, without evaluating the softplus a second time.  The pair's weights are read in
// step 9 of the previous pair (behind their last use in step 8) and have landed by the wait that ends that pair's group.
template <int M, int STEP, bool NEXT, bool SAVE, bool DOT>
__device__ __forceinline__ void t_epi_step_h(const f32x16& x, TPieces<TArH2> (&out)[2], TEpi& st) {
    constexpr int j = STEP / 12, s = STEP % 12, half = j / 4, q = j % 4, r0 = 2 * j;
    constexpr int PW = (8 * (r0 / 4) + r0 % 4) * T_PATCH_LD * 4;
    constexpr float C2 = SOFTPLUS_BETA * 1.44269504088896f / H2_XS, C7 = H2_XS * 0.69314718055995f / SOFTPLUS_BETA;
    if constexpr (s == 0) { st.x[0] = fmaf(x[r0], st.isw, st.b[0]); }
    else if constexpr (s == 1) {
        st.x[1] = fmaf(x[r0 + 1], st.isw, st.b[1]);
        if constexpr (j < 7) t_bias_read<M, j + 1>(st.b, st.bias_addr);
        else if constexpr (NEXT) t_bias_read<M + 1, 0>(st.b, st.bias_addr);
    }
    else if constexpr (s == 2) { st.t[0] = -fabsf(st.x[0]) * C2; st.t[1] = -fabsf(st.x[1]) * C2; }
    else if constexpr (s == 3) { st.e[0] = __builtin_amdgcn_exp2f(st.t[0]); st.e[1] = __builtin_amdgcn_exp2f(st.t[1]); }
    else if constexpr (s == 4) { st.e[0] = 1.f + st.e[0]; st.e[1] = 1.f + st.e[1]; }
    else if constexpr (s == 5) { st.e[0] = __builtin_amdgcn_logf(st.e[0]); st.e[1] = __builtin_amdgcn_logf(st.e[1]); }
    else if constexpr (s == 6) { st.t[0] = fmaxf(st.x[0], 0.f); st.t[1] = fmaxf(st.x[1], 0.f); }
    else if constexpr (s == 7) { st.x[0] = fmaf(st.e[0], C7, st.t[0]); st.x[1] = fmaf(st.e[1], C7, st.t[1]); }
    else if constexpr (s == 8) {
        const unsigned h = pack_f16x2(st.x); out[half].p[0][q] = h; st.u = resid_f16x2(st.x, h);
        if constexpr (SAVE && t_epi_writes_h(j, s)) t_lds_write_b32<0>(st.patch_wr, st.hd[0]);       // pair 0's, deferred
        if constexpr (DOT) { st.s0 = fmaf(st.x[0], st.w[0], st.s0); st.s1 = fmaf(st.x[1], st.w[1], st.s1); }
    }
    else if constexpr (s == 9) {
        out[half].p[1][q] = pack_f16x2(st.u);
        if constexpr (DOT) {
            if constexpr (j < 7) t_bias_read<M, j + 1, 1024>(st.w, st.bias_addr);
            else if constexpr (NEXT) t_bias_read<M + 1, 0, 1024>(st.w, st.bias_addr);
        }
        if constexpr (SAVE && t_epi_writes_h(j, s)) t_lds_write_b32<T_PATCH_LD * 4>(st.patch_wr, st.hd[1]);
    }
    else if constexpr (s == 10) {
        if constexpr (SAVE) {
            st.t[0] = st.x[0] * (1.f / H2_XS); st.t[1] = st.x[1] * (1.f / H2_XS);
            if constexpr (!t_epi_writes_h(j, s)) st.hd = st.t;
            else t_lds_write_b32<PW>(st.patch_wr, st.t[0]);
        }
    }
    else { if constexpr (SAVE && t_epi_writes_h(j, s)) t_lds_write_b32<PW + T_PATCH_LD * 4>(st.patch_wr, st.t[1]); }
}
template <class AR, int M, int STEP, bool NEXT, bool SAVE, bool DOT>
__device__ __forceinline__ void t_epi_step(const f32x16& x, TPieces<AR> (&out)[2], TEpi& st) {
    static_assert(AR::H || !DOT, "the fp16 form only");
    if constexpr (AR::H) { t_epi_step_h<M, STEP, NEXT, SAVE, DOT>(x, out, st); return; } else {
    constexpr int j = STEP / 12, s = STEP % 12, half = j / 4, q = j % 4, r0 = 2 * j;
    constexpr int PW = (8 * (r0 / 4) + r0 % 4) * T_PATCH_LD * 4;           // patch row of value r0 (this lane's 4 h rows are in patch_wr)
    if constexpr (s == 0) { st.x[0] = x[r0] + st.b[0]; }
    else if constexpr (s == 1) {
        st.x[1] = x[r0 + 1] + st.b[1];
        if constexpr (j < 7) t_bias_read<M, j + 1>(st.b, st.bias_addr);
        else if constexpr (NEXT) t_bias_read<M + 1, 0>(st.b, st.bias_addr);
    }
    else if constexpr (s == 2) { st.t[0] = -fabsf(st.x[0]) * (SOFTPLUS_BETA * 1.44269504088896f); st.t[1] = -fabsf(st.x[1]) * (SOFTPLUS_BETA * 1.44269504088896f); }
    else if constexpr (s == 3) { st.e[0] = __builtin_amdgcn_exp2f(st.t[0]); st.e[1] = __builtin_amdgcn_exp2f(st.t[1]); }
    else if constexpr (s == 4) {
        st.e[0] = 1.f + st.e[0]; st.e[1] = 1.f + st.e[1];
        if constexpr (SAVE && t_epi_writes_b(j, s)) t_lds_write_b32<0>(st.patch_wr, st.hd[0]);       // pair 0's, deferred (below)
    }
    else if constexpr (s == 5) {
        st.e[0] = __builtin_amdgcn_logf(st.e[0]); st.e[1] = __builtin_amdgcn_logf(st.e[1]);
        if constexpr (SAVE && t_epi_writes_b(j, s)) t_lds_write_b32<T_PATCH_LD * 4>(st.patch_wr, st.hd[1]);
    }
    else if constexpr (s == 6) { st.t[0] = fmaxf(st.x[0], 0.f); st.t[1] = fmaxf(st.x[1], 0.f); }
    else if constexpr (s == 7) {
        st.x[0] = fmaf(st.e[0], 0.69314718055995f / SOFTPLUS_BETA, st.t[0]); st.x[1] = fmaf(st.e[1], 0.69314718055995f / SOFTPLUS_BETA, st.t[1]);
        // pair 0 is finished in group 0 while the PREVIOUS m-tile's patch is still being read (its last two float4 leave in group
        // 1, slots 2-3): pair 0's two values wait in hd until pair 1's steps 4 and 5 (group 1, slots 4-5)
        if constexpr (SAVE && !t_epi_writes_b(j, s)) st.hd = st.x;
        else if constexpr (SAVE) t_lds_write_b32<PW>(st.patch_wr, st.x[0]);
    }
    else if constexpr (s == 8) {
        if constexpr (SAVE && t_epi_writes_b(j, s)) t_lds_write_b32<PW + T_PATCH_LD * 4>(st.patch_wr, st.x[1]);
        const unsigned h = pack_bf16x2(st.x); out[half].p[0][q] = h; st.u = unpack_bf16x2(h);
    }
    else if constexpr (s == 9) { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; }
    else if constexpr (s == 10) { const unsigned h = pack_bf16x2(st.x); out[half].p[1][q] = h; st.u = unpack_bf16x2(h); }
    else { st.x[0] -= st.u[0]; st.x[1] -= st.u[1]; out[half].p[2][q] = pack_bf16x2(st.x); }
    }
}

struct TRing {
    unsigned rd_addr;          // LDS byte address of the slot being read + lane * 16
    unsigned rd_slot;
    unsigned is_slot;          // slot the stage being issued goes to
    unsigned is_goff;          // byte offset of that stage in the packed stream (wraps over the kernel's stages)
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned lds_base;
    char* lds;
    int wave, lane;
};
// One LDS-DMA of the stage being issued: fragment wave + 4 i (1 KiB: 64 lanes x 16 B)
template <class C, int I>
__device__ __forceinline__ void t_ring_issue_one(TRing& R) {
    constexpr int SB = C::AR::STAGE_BYTES;
    const unsigned frag = R.wave + 4 * I;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(R.rsrc, (__attribute__((address_space(3))) void*)(R.lds + R.is_slot * SB + frag * 1024),
                                             16, R.lane * 16, R.is_goff + frag * 1024, 0, 0);
    if constexpr (I == C::AR::DMA - 1) {
        R.is_slot = (R.is_slot + 1 == C::NSTAGE) ? 0 : R.is_slot + 1;
        R.is_goff = (R.is_goff + SB == (unsigned)(C::STAGES * SB)) ? 0 : R.is_goff + SB;
    }
}
template <class C, int I, int N>
__device__ __forceinline__ void t_ring_issue_range(TRing& R) {
    if constexpr (I < N) { t_ring_issue_one<C, I>(R); t_ring_issue_range<C, I + 1, N>(R); }
}
template <class C>
__device__ __forceinline__ void t_ring_advance_read(TRing& R) {
    R.rd_slot = (R.rd_slot + 1 == C::NSTAGE) ? 0 : R.rd_slot + 1;
    R.rd_addr = R.lds_base + R.rd_slot * C::AR::STAGE_BYTES + R.lane * 16;
}
template <class AR, int G>
__device__ __forceinline__ void t_read_group(TFrag<AR> (&a)[2], unsigned addr) {
    DH_UNROLL for (int t = 0; t < 2; ++t) {
        a[t].p[0] = t_lds_b128<(2 * G * AR::NP + 0) * 1024>(addr + t * AR::NP * 1024);
        a[t].p[1] = t_lds_b128<(2 * G * AR::NP + 1) * 1024>(addr + t * AR::NP * 1024);
        if constexpr (AR::NP == 3) a[t].p[2] = t_lds_b128<(2 * G * AR::NP + 2) * 1024>(addr + t * AR::NP * 1024);
    }
}

// float offset of float4 r4 of m-tile M inside a wave's half of a native [64 x 256] tile (tile.h: float4 index
// (((w*MT + m)*2 + t)*4 + r4)*64 + lane with w = M / 2, t = M % 2; the m and lane terms are in TSave::base)
constexpr int t_native_off(int M, int r4) { return (M / 2) * 4096 + (M % 2) * 1024 + r4 * 256; }
template <int M, int R4>
__device__ __forceinline__ void t_save_store(const TSave& sv) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sv.v[R4 & 1]), sv.rsrc, sv.loff + 4 * t_native_off(M, R4), 0, 2);     // nt, as DH_TILE_ST
}
// T_SAVE_MFMA form: float4 R4 of the transposed m-tile = points 8 R4 + 4 h + (0..3) of this lane's feature
template <int M, int R4>
__device__ __forceinline__ void t_save_store_d(const TSave& sv) {
    const f32x4 v = {sv.d[4 * R4 + 0], sv.d[4 * R4 + 1], sv.d[4 * R4 + 2], sv.d[4 * R4 + 3]};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), sv.rsrc, sv.loff + 4 * t_native_off(M, R4), 0, T_TILE_ST_AUX);
}
// transposition MFMA I (0..3) of the m-tile whose pieces are bs (k-steps 2M, 2M+1): lo of k-step 0, lo of k-step 1, hi, hi --
// every output element receives exactly one non-zero product per MFMA: lo / 16, then + hi / 16 (exact in fp32)
template <int I>
__device__ __forceinline__ void t_save_mfma_step(TSave& sv, const TPieces<TArH2> (&bs)[2]) {
    constexpr int ks = I % 2, piece = I < 2 ? 1 : 0;
    if constexpr (I == 0) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        sv.d = mfma_h(bs[ks].p[piece], sv.sel[ks], z);
    } else sv.d = mfma_h(bs[ks].p[piece], sv.sel[ks], sv.d);
}
// selector fragment of k-step ks (0, 1) of an m-tile, as a B operand: lane (n, h) slot i = 1/16 iff n == 16 ks + 8 (i / 4) + 4 h + i % 4
__device__ __forceinline__ u32x4 t_selector(int n, int h, int ks) {
    const int t = n - 16 * ks - 4 * h;                     // 0..3 -> slot t, 8..11 -> slot t - 4
    const bool ok = (t >= 0 && t < 4) || (t >= 8 && t < 12);
    const int i = t < 8 ? t : t - 4;
    u32x4 e;
    DH_UNROLL for (int q = 0; q < 4; ++q) e[q] = (ok && (i >> 1) == q) ? (0x2C00u << (16 * (i & 1))) : 0u;      // fp16 1/16 = 0x2C00
    return e;
}

// MFMA I (0 .. 2 NPROD - 1) of group G (m-tiles 2G, 2G+1) into set NB: product-major, consecutive MFMAs hit different accumulators
template <class AR, int NB, int G, int I>
__device__ __forceinline__ void t_mfma_step(TAcc& A, const TFrag<AR> (&a)[2], const TPieces<AR>& b) {
    constexpr int p = I / 2, t = I % 2, mt = 2 * G + t;
    A.s[NB][mt] = AR::mfma(a[t].p[AR::pw[p]], b.p[AR::px[p]], A.s[NB][mt]);
}
// What is dealt under one k-step's MFMAs (compile-time): the accumulator set NB it writes; the epilogue m-tile EM of the OTHER
// set (-1: none) and which half EH of its 96 micro-steps; ENEXT: m-tile EM + 1 follows; SM (training kernel): the m-tile whose
// patch is complete -- two of its float4 are read in group 0 and stored in group 1, the other two read in group 1 and stored in
// group 2 (-1: none)
template <class C_, int NB_, int EM_, int EH_, bool ENEXT_, int SM_, bool DOT_ = false>
struct TK {
    using C = C_;
    static constexpr int NB = NB_, EM = EM_, EH = EH_, SM = SM_;
    static constexpr bool ENEXT = ENEXT_, DOT = DOT_;
};
// the 2 NPROD MFMAs of group G, each followed by its share of the dealt work (12 / (2 NPROD) epilogue micro-steps); DA / DB: the
// LDS-DMA piece issued after MFMA 3 / 9 (TArB3) or DA after MFMA 2 (TArH2)   (-1: none)
template <class K, int G, int DA, int DB, int I>
__device__ __forceinline__ void t_group_steps(TAcc& A, const TFrag<typename K::C::AR> (&a)[2], const TPieces<typename K::C::AR>& b,
                                              const TPieces<typename K::C::AR> (&bs)[2], TPieces<typename K::C::AR> (&bn)[2], TEpi& st,
                                              TRing& R, TSave& sv) {
    using AR = typename K::C::AR;
    constexpr int NMF = 2 * AR::NPROD, EPS = 12 / NMF, DAI = AR::H ? 2 : 3;
    if constexpr (I < NMF) {
        t_mfma_step<AR, K::NB, G, I>(A, a, b);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (I == DAI && DA >= 0) { t_ring_issue_one<typename K::C, DA>(R); __builtin_amdgcn_sched_barrier(0); }
        if constexpr (I == 9 && DB >= 0) { t_ring_issue_one<typename K::C, DB>(R); __builtin_amdgcn_sched_barrier(0); }
#ifdef T_PROBE_NO_SAVE_READS_STORES                      // timing probe: no patch reads, no tile stores in the dealt stream
        if constexpr (false) {
#else
        if constexpr (t_save_mfma<typename K::C>() && K::SM >= 0 && I < 4) {
            // the m-tile on the matrix pipe: four MFMAs behind group 0's first four, the stores where the patch form has them
            if constexpr (G == 0) { t_save_mfma_step<I>(sv, bs); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (G == 1 && I < 2) { t_save_store_d<K::SM, I>(sv); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (G == 2 && I < 2) { t_save_store_d<K::SM, I + 2>(sv); __builtin_amdgcn_sched_barrier(0); }
        } else if constexpr (K::C::TRAIN && K::SM >= 0 && I < 4) {
#endif
            if constexpr (G == 0 && I < 2) { t_lds_read<32 * I>(sv.v[I], sv.patch_rd); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (G == 1 && I < 2) { t_save_store<K::SM, I>(sv); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (G == 1 && I >= 2) { t_lds_read<32 * I>(sv.v[I - 2], sv.patch_rd); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (G == 2 && I < 2) { t_save_store<K::SM, I + 2>(sv); __builtin_amdgcn_sched_barrier(0); }
        }
        if constexpr (K::EM >= 0 && K::EM < T_NM) {
            t_epi_step<AR, K::EM, K::EH * 48 + 12 * G + EPS * I, K::ENEXT, t_save_patch<typename K::C>(), K::DOT>(A.s[1 - K::NB][K::EM], bn, st);
            if constexpr (EPS == 2) t_epi_step<AR, K::EM, K::EH * 48 + 12 * G + EPS * I + 1, K::ENEXT, t_save_patch<typename K::C>(), K::DOT>(A.s[1 - K::NB][K::EM], bn, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        t_group_steps<K, G, DA, DB, I + 1>(A, a, b, bs, bn, st, R, sv);
    }
}

// The wait that ends group G: the fragments read at its start (and the bias pair, the patch float4s) must have landed -- but not the
// patch WRITES of the pair the group deals (training kernel), which are its youngest LDS operations (micro-steps 8..11 of pair
// j = 4 EH + G: the deferred pair 0 under pair 1, then the pair's own two) and which nobody reads before several later waits.  LDS
// operations of a wave complete in order, so a counted lgkmcnt leaves exactly those in flight; waiting for them too (lgkmcnt(0))
// exposed a write's LDS latency four times per k-step: 0.98 -> 0.70 ms of the training forward in the probe that removed the
// writes (profiles/r05_chain_t_stamps.json).
// The count comes from the emitters' own table (t_epi_writes above): group G of half EH deals exactly the 12 micro-steps of pair j.
template <class K, int G>
constexpr int t_late_writes() {
    constexpr int j = 4 * K::EH + G;
    return (t_save_patch<typename K::C>() && K::EM >= 0 && K::EM < T_NM) ? t_epi_pair_writes<K::C::AR::H>(j) : 0;
}
static_assert(t_epi_pair_writes<true>(0) == 0 && t_epi_pair_writes<true>(1) == 4 && t_epi_pair_writes<true>(5) == 2 &&
              t_epi_pair_writes<false>(0) == 0 && t_epi_pair_writes<false>(1) == 4 && t_epi_pair_writes<false>(7) == 2, "round 5's tally");
template <class K, int G>
__device__ __forceinline__ void t_group_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(t_late_writes<K, G>()) : "memory"); }
// one k-step: entering, a0 holds group 0's fragments; leaving, a0 holds group 0 of the NEXT stage of the stream.
// LDS-DMA pieces of the stage being issued: TArB3 6 per wave and k-step (3, 4 | 5 | barrier | 0, 1 | 2), TArH2 4 (2 | 3 | barrier | 0 | 1)
template <class K>
__device__ __forceinline__ void t_kstep(TAcc& A, const TPieces<typename K::C::AR>& b, const TPieces<typename K::C::AR> (&bs)[2],
                                        TPieces<typename K::C::AR> (&bn)[2], TEpi& st,
                                        TFrag<typename K::C::AR> (&a0)[2], TFrag<typename K::C::AR> (&a1)[2], TRing& R, TSave& sv) {
    using C = typename K::C;
    using AR = typename C::AR;
    constexpr bool B3 = !AR::H;
    t_read_group<AR, 1>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    t_group_steps<K, 0, B3 ? 3 : 2, B3 ? 4 : -1, 0>(A, a0, b, bs, bn, st, R, sv);        // the second half of the stage begun last k-step
    t_group_wait<K, 0>();
    __builtin_amdgcn_sched_barrier(0);
    t_read_group<AR, 2>(a0, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    t_group_steps<K, 1, B3 ? 5 : 3, -1, 0>(A, a1, b, bs, bn, st, R, sv);                 // the stage is fully issued
    t_group_wait<K, 1>();
    __builtin_amdgcn_sched_barrier(0);
    // the NEXT k-step's pieces: this wave's DMAs for it have landed once at most DEPTH-1 younger groups are outstanding (tile
    // stores in flight count too and only make the wait stricter); after the barrier everyone's have, and everyone has left the
    // previous k-step (its slot may be refilled)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR::DMA * (C::DEPTH - 1)) : "memory");
    asm volatile("s_barrier" ::: "memory");
    t_read_group<AR, 3>(a1, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    t_group_steps<K, 2, 0, B3 ? 1 : -1, 0>(A, a0, b, bs, bn, st, R, sv);                 // a new stage: its slot was freed by the barrier
    t_group_wait<K, 2>();
    __builtin_amdgcn_sched_barrier(0);
    t_ring_advance_read<C>(R);
    t_read_group<AR, 0>(a0, R.rd_addr);
    __builtin_amdgcn_sched_barrier(0);
    t_group_steps<K, 3, B3 ? 2 : 1, -1, 0>(A, a1, b, bs, bn, st, R, sv);
    t_group_wait<K, 3>();
    __builtin_amdgcn_sched_barrier(0);
}
template <class C, int NB>
using TBare = TK<C, NB, -1, 0, false, -1>;

template <int NB>
__device__ __forceinline__ void t_zero(TAcc& A) {
    DH_UNROLL for (int m = 0; m < T_NM; ++m) DH_UNROLL for (int r = 0; r < 16; ++r) A.s[NB][m][r] = 0.f;
}
// 16 softplus(z) from zs = 16 z (TArH2's scaled variable; bit-identical to 16 * softplus100(z))
__device__ __forceinline__ float softplus100_xs(float zs) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(zs) * (SOFTPLUS_BETA * 1.44269504088896f / H2_XS));
    const float l = __builtin_amdgcn_logf(1.f + e);
    return fmaf(l, H2_XS * 0.69314718055995f / SOFTPLUS_BETA, fmaxf(zs, 0.f));
}
// the one exposed epilogue of a layer: m-tile 0 of the source set, outside the MFMA stream -- the same arithmetic as t_epi_step, its
// sixteen biases fetched with ONE LDS latency; leaves m-tile 1's first pair in st.b
template <class AR, bool SAVE, bool DOT>
__device__ __forceinline__ void t_epi_exposed(const f32x16& x, TPieces<AR> (&out)[2], TEpi& st) {
    f32x4 bb[4];
    t_lds_read4_w<0, 32, 64, 96>(bb[0], bb[1], bb[2], bb[3], st.bias_addr);
    [[maybe_unused]] f32x4 ww[4];
    if constexpr (DOT) t_lds_read4_w<1024, 1024 + 32, 1024 + 64, 1024 + 96>(ww[0], ww[1], ww[2], ww[3], st.bias_addr);
    DH_UNROLL for (int j = 0; j < 8; ++j) {
        const int g = j / 2, i0 = 2 * (j % 2), r0 = 2 * j, half = j / 4, q = j % 4;
        f32x2 v, sv2;
        if constexpr (AR::H) {
            v[0] = softplus100_xs(fmaf(x[r0], st.isw, bb[g][i0]));
            v[1] = softplus100_xs(fmaf(x[r0 + 1], st.isw, bb[g][i0 + 1]));
            sv2 = v * (1.f / H2_XS);
            if constexpr (DOT) { st.s0 = fmaf(v[0], ww[g][i0], st.s0); st.s1 = fmaf(v[1], ww[g][i0 + 1], st.s1); }
        } else {
            v[0] = softplus100(x[r0] + bb[g][i0]);
            v[1] = softplus100(x[r0 + 1] + bb[g][i0 + 1]);
            sv2 = v;
        }
        if constexpr (SAVE) {
            const unsigned a = st.patch_wr + (8 * (r0 / 4) + r0 % 4) * T_PATCH_LD * 4;
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:%3" ::"v"(a), "v"(sv2[0]), "v"(sv2[1]), "n"(T_PATCH_LD * 4) : "memory");
        }
        if constexpr (AR::H) {
            const unsigned h = pack_f16x2(v);
            out[half].p[0][q] = h; out[half].p[1][q] = pack_f16x2(resid_f16x2(v, h));
        } else {
            const unsigned h = pack_bf16x2(v);
            const f32x2 r1 = v - unpack_bf16x2(h);
            const unsigned m = pack_bf16x2(r1);
            const f32x2 r2 = r1 - unpack_bf16x2(m);
            out[half].p[0][q] = h; out[half].p[1][q] = m; out[half].p[2][q] = pack_bf16x2(r2);
        }
    }
    t_bias_read_w<1, 0>(st.b, st.bias_addr);
    if constexpr (DOT) t_bias_read_w<1, 0, 1024>(st.w, st.bias_addr);
}
// k-steps 2M, 2M+1 (input pieces = the epilogue of m-tile M of the source set) with the epilogue of m-tile M+1 dealt under them
// and (training kernel) m-tile M saved; MEND: m-tiles of the source the layer consumes (8; the skip layer takes 7 of lin3's)
template <class C, int NB, int M, int MEND, bool DOT>
__device__ __forceinline__ void t_mpair(TAcc& A, TPieces<typename C::AR> (&bA)[2], TPieces<typename C::AR> (&bB)[2], TEpi& st,
                                        TFrag<typename C::AR> (&a0)[2], TFrag<typename C::AR> (&a1)[2], TRing& R, TSave& sv) {
    if constexpr (M < MEND) {
        constexpr int EM = M + 1 < MEND ? M + 1 : -1;
        constexpr bool EN = M + 2 < MEND;                 // m-tile EM + 1 will have its epilogue dealt too
        using K0 = TK<C, NB, EM, 0, EN, M, DOT>;
        using K1 = TK<C, NB, EM, 1, EN, -1, DOT>;
        if constexpr (M % 2 == 0) {
            t_kstep<K0>(A, bA[0], bA, bB, st, a0, a1, R, sv);
            t_kstep<K1>(A, bA[1], bA, bB, st, a0, a1, R, sv);
        } else {
            t_kstep<K0>(A, bB[0], bB, bA, st, a0, a1, R, sv);
            t_kstep<K1>(A, bB[1], bB, bA, st, a0, a1, R, sv);
        }
        t_mpair<C, NB, M + 1, MEND, DOT>(A, bA, bB, st, a0, a1, R, sv);
    }
}
// the main part of a layer: accumulates MEND * 32 input features into set NB from the finished set 1 - NB, whose bias row is
// at bias_row (LDS byte address, + 16 h); training kernel: the source layer's activations go to the native tile sv.base points at
template <class C, int NB, int MEND, bool DOT = false>
__device__ __forceinline__ void t_layer(TAcc& A, TPieces<typename C::AR> (&bA)[2], TPieces<typename C::AR> (&bB)[2], TEpi& st,
                                        TFrag<typename C::AR> (&a0)[2], TFrag<typename C::AR> (&a1)[2], TRing& R, TSave& sv,
                                        unsigned bias_row, float isw) {
    t_zero<NB>(A);
    st.bias_addr = bias_row;
    st.isw = isw;
    t_epi_exposed<typename C::AR, t_save_patch<C>(), DOT>(A.s[1 - NB][0], bA, st);
    __builtin_amdgcn_sched_barrier(0);
    t_mpair<C, NB, 0, MEND, DOT>(A, bA, bB, st, a0, a1, R, sv);
}

// Embedding image of the wave's 32 points in its private LDS rows: [x, sin(2^k x), cos(2^k x)]_{k<6}, zero padded to 48
// (mlp_common.h embed_tile's order).  The two lanes of a point share the work: h = 0 takes frequencies 0..2, h = 1 3..5.
__device__ __forceinline__ void t_embed_rows(const float (&x)[3], unsigned row, int h) {
    if (h == 0) { t_lds_write_b32<0>(row, x[0]); t_lds_write_b32<4>(row, x[1]); t_lds_write_b32<8>(row, x[2]); }
    else {
        t_lds_write_b32<4 * 39>(row, 0.f); t_lds_write_b32<4 * 40>(row, 0.f); t_lds_write_b32<4 * 41>(row, 0.f);
        t_lds_write_b32<4 * 42>(row, 0.f); t_lds_write_b32<4 * 43>(row, 0.f); t_lds_write_b32<4 * 44>(row, 0.f);
        t_lds_write_b32<4 * 45>(row, 0.f); t_lds_write_b32<4 * 46>(row, 0.f); t_lds_write_b32<4 * 47>(row, 0.f);
    }
    _Pragma("unroll 1") for (int kk = 0; kk < 3; ++kk) {
        const int k = 3 * h + kk;
        const float f = (float)(1 << k);
        const unsigned rk = row + 24 * k;
        float sn, co;
        sincosf(x[0] * f, &sn, &co); t_lds_write_b32<12>(rk, sn); t_lds_write_b32<24>(rk, co);
        sincosf(x[1] * f, &sn, &co); t_lds_write_b32<16>(rk, sn); t_lds_write_b32<28>(rk, co);
        sincosf(x[2] * f, &sn, &co); t_lds_write_b32<20>(rk, sn); t_lds_write_b32<32>(rk, co);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// 8 fp32 of one k-step (this lane's slots i = 0..7) -> pieces (TArH2: of 16 x the values)
template <class AR>
__device__ __forceinline__ TPieces<AR> t_split8(const f32x4& lo, const f32x4& hi) {
    TPieces<AR> r;
    if constexpr (AR::H) {
        const H2 b = split2(lo * H2_XS, hi * H2_XS);
        r.p[0] = b.p[0]; r.p[1] = b.p[1];
    } else {
        const Bf3 b = split3(lo, hi);
        DH_UNROLL for (int p = 0; p < 3; ++p) r.p[p] = __builtin_bit_cast(u32x4, b.p[p]);
    }
    return r;
}
// the three k-steps of the embedding as B pieces: lane (p, h) holds features 16 s + 8 g + 4 h + (0..3), g = 0, 1 of k-step s
template <class AR>
__device__ __forceinline__ void t_embed_pieces(unsigned row_h, TPieces<AR>& e0, TPieces<AR>& e1, TPieces<AR>& e2) {
    e0 = t_split8<AR>(t_lds_read_w<0>(row_h), t_lds_read_w<32>(row_h));
    e1 = t_split8<AR>(t_lds_read_w<64>(row_h), t_lds_read_w<96>(row_h));
    e2 = t_split8<AR>(t_lds_read_w<128>(row_h), t_lds_read_w<160>(row_h));
}
// training kernel: the wave's half of the aux native tile ([64 x 64], tile.h aux_store_native) from the embedding image: lane L's
// float4 (t, r4) = column 32 t + (L & 31) of points 8 r4 + 4 (L >> 5) + (0..3).  col_addr = image + (4 (L>>5) rows, column L&31)
template <int I>
__device__ __forceinline__ void t_store_eaux(unsigned col_addr, int fl, __amdgpu_buffer_rsrc_t rsrc, unsigned loff) {
    if constexpr (I < 8) {
        constexpr int t = I / 4, r4 = I % 4;
        f32x4 v = t_lds_read32x4_w<((8 * r4 + 0) * T_EMB_LD + 32 * t) * 4, ((8 * r4 + 1) * T_EMB_LD + 32 * t) * 4,
                                   ((8 * r4 + 2) * T_EMB_LD + 32 * t) * 4, ((8 * r4 + 3) * T_EMB_LD + 32 * t) * 4>(col_addr);
        if (32 * t + fl >= EMB) v = f32x4{0.f, 0.f, 0.f, 0.f};                   // the image is zero only up to column 47
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc, loff + 4 * (t * 4 + r4) * 256, 0, 0);
        t_store_eaux<I + 1>(col_addr, fl, rsrc, loff);
    }
}

// lin8 row 0 on softplus(lin7 + bias): this lane's 128 features of its point (accumulator set 1)
// TArH2: isw = 1 / S_w of lin7; the sums come out 16 x (the caller divides)
template <class AR, int M>
__device__ __forceinline__ void t_final_dot(const TAcc& A, unsigned bias_base, float isw, float& s0, float& s1) {
    if constexpr (M < T_NM) {
        f32x4 bb[4], ww[4];
        t_lds_read4_w<7 * 1024 + 4 * (32 * M), 7 * 1024 + 4 * (32 * M + 8), 7 * 1024 + 4 * (32 * M + 16), 7 * 1024 + 4 * (32 * M + 24)>(bb[0], bb[1], bb[2], bb[3], bias_base);
        t_lds_read4_w<8 * 1024 + 4 * (32 * M), 8 * 1024 + 4 * (32 * M + 8), 8 * 1024 + 4 * (32 * M + 16), 8 * 1024 + 4 * (32 * M + 24)>(ww[0], ww[1], ww[2], ww[3], bias_base);
        DH_UNROLL for (int r4 = 0; r4 < 4; ++r4) {
            if constexpr (AR::H) {
                s0 = fmaf(softplus100_xs(fmaf(A.s[1][M][4 * r4 + 0], isw, bb[r4][0])), ww[r4][0], s0);
                s1 = fmaf(softplus100_xs(fmaf(A.s[1][M][4 * r4 + 1], isw, bb[r4][1])), ww[r4][1], s1);
                s0 = fmaf(softplus100_xs(fmaf(A.s[1][M][4 * r4 + 2], isw, bb[r4][2])), ww[r4][2], s0);
                s1 = fmaf(softplus100_xs(fmaf(A.s[1][M][4 * r4 + 3], isw, bb[r4][3])), ww[r4][3], s1);
            } else {
                s0 = fmaf(softplus100(A.s[1][M][4 * r4 + 0] + bb[r4][0]), ww[r4][0], s0);
                s1 = fmaf(softplus100(A.s[1][M][4 * r4 + 1] + bb[r4][1]), ww[r4][1], s1);
                s0 = fmaf(softplus100(A.s[1][M][4 * r4 + 2] + bb[r4][2]), ww[r4][2], s0);
                s1 = fmaf(softplus100(A.s[1][M][4 * r4 + 3] + bb[r4][3]), ww[r4][3], s1);
            }
        }
        t_final_dot<AR, M + 1>(A, bias_base, isw, s0, s1);
    }
}
// training kernel: lin8 rows 1..256 (accumulator set 0) + their bias (row 9) -> the feature tile, m-tile by m-tile through the patch
// (fs: 1 for TArB3; 1 / (16 S_w) of lin8 for TArH2)
template <int M>
__device__ __forceinline__ void t_store_feat(const TAcc& A, unsigned bias_base, const TEpi& st, TSave& sv, float fs) {
    if constexpr (M < T_NM) {
        f32x4 bb[4];
        t_lds_read4_w<9 * 1024 + 4 * (32 * M), 9 * 1024 + 4 * (32 * M + 8), 9 * 1024 + 4 * (32 * M + 16), 9 * 1024 + 4 * (32 * M + 24)>(bb[0], bb[1], bb[2], bb[3], bias_base);
        DH_UNROLL for (int g = 0; g < 4; ++g)
            DH_UNROLL for (int i = 0; i < 4; ++i) {
                const float v = fmaf(A.s[0][M][4 * g + i], fs, bb[g][i]);
                const unsigned a = st.patch_wr + (8 * g + i) * T_PATCH_LD * 4;
                asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
            }
        f32x4 o[4];
        t_lds_read4_w<0, 32, 64, 96>(o[0], o[1], o[2], o[3], sv.patch_rd);
        sv.v[0] = o[0]; sv.v[1] = o[1];
        t_save_store<M, 0>(sv); t_save_store<M, 1>(sv);
        sv.v[0] = o[2]; sv.v[1] = o[3];
        t_save_store<M, 2>(sv); t_save_store<M, 3>(sv);
        t_store_feat<M + 1>(A, bias_base, st, sv, fs);
    }
}

// this lane's point of the tile, and the NEXT tile's coordinates into xn (fetched while the tile's last exposed section runs: a load
// issued at a tile's start would be waited for at once, behind every LDS-DMA in flight).  The index is formed from a value the
// compiler cannot hoist: as a loop invariant it is a 64-bit register pair that gets spilled, and a scratch reload inside the
// stream waits for every LDS-DMA in flight.
__device__ __forceinline__ int64_t t_next_points(const float* __restrict__ pts, int64_t npts, int64_t tile, int lp, float (&xn)[3]) {
    asm volatile("" : "+v"(lp));
    const int64_t gp = tile * T_PTS + lp, gq = gp + (int64_t)gridDim.x * T_PTS;
    xn[0] = xn[1] = xn[2] = 0.f;
    if (gq < npts) { xn[0] = pts[gq * 3 + 0]; xn[1] = pts[gq * 3 + 1]; xn[2] = pts[gq * 3 + 2]; }
    return gp;
}

#ifdef DH_T_DEBUG
// development build: the raw accumulators (pre-bias) of layer dbg_layer as [npts][256]
template <int NB>
__device__ __forceinline__ void t_dump(const TAcc& A, float* dbg, int64_t gp, int64_t npts, int h) {
    if (gp >= npts) return;
    DH_UNROLL for (int m = 0; m < T_NM; ++m)
        DH_UNROLL for (int r = 0; r < 16; ++r) dbg[gp * 256 + 32 * m + 8 * (r / 4) + 4 * h + (r % 4)] = A.s[NB][m][r];
}
#define T_DUMP(NB, L) if (dbg_layer == (L)) t_dump<NB>(A, dbg, tile * T_PTS + wave * 32 + p, npts, h)
#define T_DBG_PARAMS , float* dbg, int dbg_layer
#else
#define T_DUMP(NB, L)
#define T_DBG_PARAMS
#endif

// Diagnostic build only (-DDH_STAMPS, scripts/stamps.sh): where a tile's time goes in the register-resident chains.  Stamps of the
// workgroup's SECOND tile are kept in scalar registers (a store inside the tile loop would count in the ring's vmcnt waits) and
// written after the loop; T_STAMP2 picks a static slot from the run-time q (a run-time index would send the array to scratch).
#ifdef DH_STAMPS
constexpr int T_NSTAMP = 16;
static __device__ unsigned long long dh_stamps_t[1024 * 4 * T_NSTAMP];
extern "C" int dh_dev_read_stamps_t(unsigned long long* host, long long n) {
    const long long total = (long long)(sizeof(dh_stamps_t) / sizeof(unsigned long long));
    if (n > total) n = total;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(dh_stamps_t), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -3;
}
#define T_STAMP(k)                                                                                        \
    do {                                                                                                  \
        if (t_it == 1) {                                                                                  \
            unsigned long long t_;                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            t_ts[k] = t_;                                                                                 \
        }                                                                                                 \
    } while (0)
#define T_STAMP2(k0, k1) do { if (q == 0) T_STAMP(k0); else T_STAMP(k1); } while (0)
#else
#define T_STAMP(k) do { } while (0)
#define T_STAMP2(k0, k1) do { } while (0)
#endif

// C = TCfgNoGrad: sdf only.  C = TCfgTrain: also the saved tiles of the training forward (act[l] = softplus(lin_l), l = 0..7, feat =
// lin8 rows 1..256, eaux = the embedding) in the layouts of sdf_fwd_train_kernel (tile.h native tiles of 64 points: a workgroup
// tile is two of them, waves 0-1 and 2-3).
template <class C>
__device__ __forceinline__ void sdf_chain_t_body(const void* __restrict__ stream, const float* __restrict__ bias10,
                                                 const float* __restrict__ b8_0, const float* __restrict__ pts, int64_t npts,
                                                 float* __restrict__ sdf_out, float* __restrict__ feat, float* __restrict__ act,
                                                 float* __restrict__ eaux, unsigned* __restrict__ absmax T_DBG_PARAMS) {
    __shared__ __attribute__((aligned(16))) char lds[t_lds_bytes<C>()];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, p = lane & 31;
    using AR = typename C::AR;
    constexpr int SB = AR::STAGE_BYTES, BB = AR::BIAS_BYTES;
    float* lbias = reinterpret_cast<float*>(lds + C::NSTAGE * SB);
    for (int i = tid; i < BB / 4; i += 256) lbias[i] = bias10[i];
    // TArH2: 1 / S_w of lin0..lin8 (row 10 of the table; scalar loads, before any LDS-DMA is in flight)
    float isw[N_SDF];
    DH_UNROLL for (int l = 0; l < N_SDF; ++l) isw[l] = AR::H ? bias10[10 * 256 + l] : 1.f;
    TRing R;
    R.lds = lds; R.lds_base = (unsigned)(uintptr_t)lds; R.wave = wave; R.lane = lane;
    R.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(stream), 0, C::STAGES * SB, 0x00020000);
    R.is_goff = 0; R.is_slot = 0; R.rd_slot = 0; R.rd_addr = R.lds_base + lane * 16;
    const unsigned bias_base = R.lds_base + C::NSTAGE * SB + 16 * h;
    const unsigned emb_wave = R.lds_base + C::NSTAGE * SB + BB + wave * T_EMB_BYTES;
    const unsigned emb_row = emb_wave + p * (T_EMB_LD * 4);
    TAcc A;
    TPieces<AR> bA[2], bB[2];
    TEpi st;
    st.isw = 1.f;
    TSave sv;
    TFrag<AR> a0[2], a1[2];
    if constexpr (C::TRAIN) {
        const unsigned patch = R.lds_base + C::NSTAGE * SB + BB + 4 * T_EMB_BYTES + wave * T_PATCH_BYTES;
        if constexpr (t_save_patch<C>()) {
            st.patch_wr = patch + (4 * h * T_PATCH_LD + p) * 4;
            sv.patch_rd = patch + (p * T_PATCH_LD + 4 * h) * 4;
        } else { sv.sel[0] = t_selector(p, h, 0); sv.sel[1] = t_selector(p, h, 1); }
    }
    // ring prologue: DEPTH full stages + the first half of the next (the steady state enters a k-step with half a stage issued)
    for (int d = 0; d < C::DEPTH; ++d) t_ring_issue_range<C, 0, AR::DMA>(R);
    t_ring_issue_range<C, 0, AR::DMA / 2>(R);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AR::DMA * (C::DEPTH - 1) + AR::DMA / 2) : "memory");
    __syncthreads();                                  // stage 0 has landed for every wave; the bias table is written
    t_read_group<AR, 0>(a0, R.rd_addr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    const int64_t ntiles = (npts + T_PTS - 1) / T_PTS, ntiles64 = (npts + TM - 1) / TM;
    float xn[3] = {0.f, 0.f, 0.f};
    {
        const int64_t g0 = (int64_t)blockIdx.x * T_PTS + wave * 32 + p;
        if (g0 < npts) { xn[0] = pts[g0 * 3 + 0]; xn[1] = pts[g0 * 3 + 1]; xn[2] = pts[g0 * 3 + 2]; }
    }
#ifdef DH_STAMPS
    unsigned long long t_ts[T_NSTAMP];
    DH_UNROLL for (int k = 0; k < T_NSTAMP; ++k) t_ts[k] = 0;
    int t_it = -1;
#endif
    _Pragma("unroll 1") for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#ifdef DH_STAMPS
        ++t_it;
#endif
        T_STAMP(0);
        // training kernel: this wave's half (m = wave & 1) of native tile tile64; lane part of every native address
        const int64_t tile64 = 2 * tile + (wave >> 1);
        const char* act_tile = nullptr;                 // act[0]'s native tile; act[l] is l * lstride further
        const int64_t lstride = ntiles64 * TILE_F * 4;
        const bool tile_ok = tile64 < ntiles64;
        if constexpr (C::TRAIN) {
            sv.loff = ((wave & 1) * 2048 + lane * 4) * 4;
            act_tile = reinterpret_cast<const char*>(act + tile64 * TILE_F);
        }
        {   // embedding
            const float x[3] = {xn[0], xn[1], xn[2]};
            t_embed_rows(x, emb_row, h);
            t_embed_pieces<AR>(emb_row + 16 * h, bA[0], bA[1], bB[0]);
            if constexpr (C::TRAIN)
                t_store_eaux<0>(emb_wave + (4 * h * T_EMB_LD + p) * 4, p, t_tile_rsrc(eaux + tile64 * AUXT_F, tile_ok, AUXT_F * 4), sv.loff);
        }
        __builtin_amdgcn_sched_barrier(0);
        T_STAMP(1);
        // lin0: 3 k-steps of the embedding into set 0
        t_zero<0>(A);
        t_kstep<TBare<C, 0>>(A, bA[0], bA, bB, st, a0, a1, R, sv);
        t_kstep<TBare<C, 0>>(A, bA[1], bA, bB, st, a0, a1, R, sv);
        t_kstep<TBare<C, 0>>(A, bB[0], bB, bA, st, a0, a1, R, sv);
        T_DUMP(0, 0);
        T_STAMP(2);
        _Pragma("unroll 1") for (int q = 0; q < 2; ++q) {
            // (the training kernel saves the SOURCE layer of each call: act[4q], act[4q+1], act[4q+2], act[3])
            if constexpr (C::TRAIN) sv.rsrc = t_tile_rsrc(act_tile + (4 * q + 0) * lstride, tile_ok, TILE_F * 4);
            t_layer<C, 1, 8>(A, bA, bB, st, a0, a1, R, sv, bias_base + (4 * q + 0) * 1024, q ? isw[4] : isw[0]);        // lin1 / lin5
            T_DUMP(1, 4 * q + 1);
            T_STAMP2(3, 7);
            if constexpr (C::TRAIN) sv.rsrc = t_tile_rsrc(act_tile + (4 * q + 1) * lstride, tile_ok, TILE_F * 4);
            t_layer<C, 0, 8>(A, bA, bB, st, a0, a1, R, sv, bias_base + (4 * q + 1) * 1024, q ? isw[5] : isw[1]);        // lin2 / lin6
            T_DUMP(0, 4 * q + 2);
            T_STAMP2(4, 8);
            if constexpr (C::TRAIN) sv.rsrc = t_tile_rsrc(act_tile + (4 * q + 2) * lstride, tile_ok, TILE_F * 4);
            t_layer<C, 1, 8>(A, bA, bB, st, a0, a1, R, sv, bias_base + (4 * q + 2) * 1024, q ? isw[6] : isw[2]);        // lin3 / lin7
            T_DUMP(1, 4 * q + 3);
            T_STAMP2(5, 9);
            if (q == 0) {
                // lin4: 14 k-steps of lin3's output (217 valid features: the packer zeroes the rest), then the embedding again
                if constexpr (C::TRAIN) sv.rsrc = t_tile_rsrc(act_tile + 3 * lstride, tile_ok, TILE_F * 4);
                t_layer<C, 0, 7>(A, bA, bB, st, a0, a1, R, sv, bias_base + 3 * 1024, isw[3]);
                if constexpr (C::TRAIN) {
                    // columns 224..255 of lin3's tile: its rows 217.. have zero weights and bias, the activation is softplus(0)
                    const float c0 = 0.69314718055995f / SOFTPLUS_BETA;
                    TSave sc;
                    sc.rsrc = sv.rsrc; sc.loff = sv.loff;
                    sc.v[0] = sc.v[1] = f32x4{c0, c0, c0, c0};
                    t_save_store<7, 0>(sc); t_save_store<7, 1>(sc); t_save_store<7, 2>(sc); t_save_store<7, 3>(sc);
                }
                t_embed_pieces<AR>(emb_row + 16 * h, bA[0], bA[1], bB[0]);
                __builtin_amdgcn_sched_barrier(0);
                t_kstep<TBare<C, 0>>(A, bA[0], bA, bB, st, a0, a1, R, sv);
                t_kstep<TBare<C, 0>>(A, bA[1], bA, bB, st, a0, a1, R, sv);
                t_kstep<TBare<C, 0>>(A, bB[0], bB, bA, st, a0, a1, R, sv);
                T_DUMP(0, 4);
                T_STAMP(6);
            }
        }
        // lin8 row 0 (the sdf) on softplus(lin7 + bias): per lane 128 features of its point, the two half-waves add up.  The patch-form
        // training kernels do this too, before lin8's rows 1..256 consume set 1 (forming row 0 inside that layer's dealt epilogue needs
        // two more in-flight LDS registers, and those were the first thing the compiler spilled there); the T_SAVE_MFMA form has the
        // registers and forms the row inside the layer (DOT), on the activations its epilogue evaluates anyway.
        if constexpr (t_save_mfma<C>()) {
            (void)t_next_points(pts, npts, tile, wave * 32 + p, xn);
            st.s0 = 0.f; st.s1 = 0.f;
            sv.rsrc = t_tile_rsrc(act_tile + 7 * lstride, tile_ok, TILE_F * 4);
            t_layer<C, 0, 8, true>(A, bA, bB, st, a0, a1, R, sv, bias_base + 7 * 1024, isw[7]);
            T_STAMP(10);
            {
                float s = (st.s0 + st.s1) * (1.f / H2_XS);
                s += __shfl_xor(s, 32);
                int lp = wave * 32 + p;
                asm volatile("" : "+v"(lp));
                const int64_t gp = tile * T_PTS + lp;
                if (h == 0 && gp < npts) sdf_out[gp] = s + b8_0[0];
            }
            sv.rsrc = t_tile_rsrc(feat + tile64 * TILE_F, tile_ok, TILE_F * 4);
            {
                // the feature tile still goes through the patch (raw fp32 accumulators, not pieces); its two lane addresses are formed
                // here, from a value the compiler cannot hoist, instead of living in two registers through the tile loop
                int pl = p, hl = h;
                asm volatile("" : "+v"(pl), "+v"(hl));
                const unsigned patch = R.lds_base + C::NSTAGE * SB + BB + 4 * T_EMB_BYTES + wave * T_PATCH_BYTES;
                st.patch_wr = patch + (4 * hl * T_PATCH_LD + pl) * 4;
                sv.patch_rd = patch + (pl * T_PATCH_LD + 4 * hl) * 4;
            }
            t_store_feat<0>(A, bias_base, st, sv, isw[8] * (1.f / H2_XS));
            T_STAMP(11);
        } else {
            const int64_t gp = t_next_points(pts, npts, tile, wave * 32 + p, xn);
            float s0 = 0.f, s1 = 0.f;
            t_final_dot<AR, 0>(A, bias_base, isw[7], s0, s1);
            float s = (s0 + s1) * (AR::H ? 1.f / H2_XS : 1.f);
            s += __shfl_xor(s, 32);
            if (h == 0 && gp < npts) sdf_out[gp] = s + b8_0[0];
            T_STAMP(10);
        }
        if constexpr (t_save_patch<C>()) {
            // lin8 rows 1..256 from softplus(lin7 + bias) into set 0 (its epilogue saves act[7]), then the feature tile
            sv.rsrc = t_tile_rsrc(act_tile + 7 * lstride, tile_ok, TILE_F * 4);
            t_layer<C, 0, 8>(A, bA, bB, st, a0, a1, R, sv, bias_base + 7 * 1024, isw[7]);
            sv.rsrc = t_tile_rsrc(feat + tile64 * TILE_F, tile_ok, TILE_F * 4);
            t_store_feat<0>(A, bias_base, st, sv, AR::H ? isw[8] * (1.f / H2_XS) : 1.f);
            T_STAMP(11);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring runs ahead of the last tile: let its DMAs land before the LDS goes away
#ifdef DH_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
        DH_UNROLL for (int k = 0; k < T_NSTAMP; ++k) dh_stamps_t[((size_t)blockIdx.x * 4 + wave) * T_NSTAMP + k] = t_ts[k];
    }
#endif
    if constexpr (AR::H && C::TRAIN) {
        // the arithmetic tag of this step's scale tables (workspace.h ABSMAX_TAG).  (The RANGE WATCH of this chain's constant
        // activation scale lives in sdf_grad_h_kernel, which reads every activation tile this kernel saves: this kernel has no vector
        // register left for a running maximum -- one more live value and the compiler spills inside the tile loop.)
        if (absmax && blockIdx.x == 0 && tid == 0) absmax[ABSMAX_TAG * ABSMAX_STRIDE] = ABSMAX_TAG_F16;
    }
}

// (the weight stream is passed as const void*: a bf16 vector type in a kernel's signature leaves rocprofv3 unable to demangle its name)
#ifdef DH_T_DEBUG
#define T_DBG_FWD , dbg, dbg_layer
#else
#define T_DBG_FWD
#endif
__global__ __launch_bounds__(256, 1) void sdf_nograd_t_kernel(const void* __restrict__ stream, const float* __restrict__ bias10,
                                                              const float* __restrict__ b8_0, const float* __restrict__ pts, int64_t npts,
                                                              float* __restrict__ sdf_out T_DBG_PARAMS) {
    sdf_chain_t_body<TCfgNoGrad<TArB3>>(stream, bias10, b8_0, pts, npts, sdf_out, nullptr, nullptr, nullptr, nullptr T_DBG_FWD);
}
__global__ __launch_bounds__(256, 1) void sdf_fwd_train_t_kernel(const void* __restrict__ stream, const float* __restrict__ bias10,
                                                                 const float* __restrict__ b8_0, const float* __restrict__ pts, int64_t npts,
                                                                 float* __restrict__ sdf_out, float* __restrict__ feat, float* __restrict__ act,
                                                                 float* __restrict__ eaux T_DBG_PARAMS) {
    sdf_chain_t_body<TCfgTrain<TArB3>>(stream, bias10, b8_0, pts, npts, sdf_out, feat, act, eaux, nullptr T_DBG_FWD);
}
// the two-piece fp16 arithmetic (DH_ARITH_SPLIT_F16): bias11 = layout.h PACKH.bias11
__global__ __launch_bounds__(256, 1) void sdf_nograd_h_kernel(const void* __restrict__ stream, const float* __restrict__ bias11,
                                                              const float* __restrict__ b8_0, const float* __restrict__ pts, int64_t npts,
                                                              float* __restrict__ sdf_out T_DBG_PARAMS) {
    sdf_chain_t_body<TCfgNoGrad<TArH2>>(stream, bias11, b8_0, pts, npts, sdf_out, nullptr, nullptr, nullptr, nullptr T_DBG_FWD);
}
__global__ __launch_bounds__(256, 1) void sdf_fwd_train_h_kernel(const void* __restrict__ stream, const float* __restrict__ bias11,
                                                                 const float* __restrict__ b8_0, const float* __restrict__ pts, int64_t npts,
                                                                 float* __restrict__ sdf_out, float* __restrict__ feat, float* __restrict__ act,
                                                                 float* __restrict__ eaux, unsigned* __restrict__ absmax T_DBG_PARAMS) {
    sdf_chain_t_body<TCfgTrain<TArH2>>(stream, bias11, b8_0, pts, npts, sdf_out, feat, act, eaux, absmax T_DBG_FWD);
}

#ifdef DH_T_DEBUG
static float* g_dbg = nullptr;
static int g_dbg_layer = -1;
extern "C" void dh_dev_nograd_t_debug(float* dbg, int layer) { g_dbg = dbg; g_dbg_layer = layer; }
#define T_DBG_ARGS , g_dbg, g_dbg_layer
#else
#define T_DBG_ARGS
#endif
// One workgroup per CU with up to 156 KB of static LDS and LDS-DMA: what the current device offers is asked once per device
// (the launchers hard-coded 256 workgroups and failed with a bare launch error where the LDS does not exist).
struct TDev { int cus; int lds; };
static TDev t_device() {
    static TDev cache[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return TDev{0, 0};
    TDev d = cache[dev];                                   // (a benign race: every thread writes the same two values)
    if (d.cus == 0) {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess) lds = 0;
        d = TDev{cus, lds};
        cache[dev] = d;
    }
    return d;
}
template <class C>
static int t_grid(int64_t ntiles) {                        // 0: the device cannot run this kernel
    const TDev d = t_device();
    if (d.cus <= 0 || d.lds < t_lds_bytes<C>()) return 0;
    return (int)(ntiles < d.cus ? ntiles : d.cus);
}
// h2: the two-piece fp16 arithmetic.  Returns DH_ERR_UNSUPPORTED (-2) where the device lacks the LDS these kernels are built on.
int launch_sdf_nograd_t(const float* packed, const float* pts, int64_t npts, float* sdf, bool h2, hipStream_t stream) {
    const int64_t ntiles = (npts + T_PTS - 1) / T_PTS;
    const int g = h2 ? t_grid<TCfgNoGrad<TArH2>>(ntiles) : t_grid<TCfgNoGrad<TArB3>>(ntiles);
    if (g <= 0) return -2;
    if (h2) hipLaunchKernelGGL(sdf_nograd_h_kernel, dim3(g), dim3(256), 0, stream, static_cast<const void*>(packed + PACKH.stream),
                               packed + PACKH.bias11, packed + PACK.sdf_b8_0, pts, npts, sdf T_DBG_ARGS);
    else hipLaunchKernelGGL(sdf_nograd_t_kernel, dim3(g), dim3(256), 0, stream, static_cast<const void*>(packed + PACKT.stream),
                            packed + PACKT.bias10, packed + PACK.sdf_b8_0, pts, npts, sdf T_DBG_ARGS);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_sdf_fwd_train_t(const float* packed, const float* pts, int64_t npts, float* sdf, float* feat, float* act, float* eaux,
                           unsigned* absmax, bool h2, hipStream_t stream) {
    const int64_t ntiles = (npts + T_PTS - 1) / T_PTS;
    const int g = h2 ? t_grid<TCfgTrain<TArH2>>(ntiles) : t_grid<TCfgTrain<TArB3>>(ntiles);
    if (g <= 0) return -2;
    if (h2) hipLaunchKernelGGL(sdf_fwd_train_h_kernel, dim3(g), dim3(256), 0, stream, static_cast<const void*>(packed + PACKH.stream),
                               packed + PACKH.bias11, packed + PACK.sdf_b8_0, pts, npts, sdf, feat, act, eaux, absmax T_DBG_ARGS);
    else hipLaunchKernelGGL(sdf_fwd_train_t_kernel, dim3(g), dim3(256), 0, stream, static_cast<const void*>(packed + PACKT.stream),
                            packed + PACKT.bias10, packed + PACK.sdf_b8_0, pts, npts, sdf, feat, act, eaux T_DBG_ARGS);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
