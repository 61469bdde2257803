// Fixed-architecture layout tables shared by every kernel and by the C ABI.
//
// Architecture = the NeuS configuration BASELINE.json names:
//   SDF MLP    39 -> 256 x8 (skip at 4, lin3 out 217) -> 257, softplus(beta=100), weight-norm
//   colour MLP 289 -> 256 x4 -> 3, ReLU, sigmoid, weight-norm ; variance scalar.
// Host + device header (plain C++14, constexpr only).
#pragma once
#include <stdint.h>

// This library is compiled WITHOUT packed-fp32 VALU instructions (v_pk_mul / v_pk_add / v_pk_fma_f32, v_pk_mov_b32): hipcc flags
// `-Xclang -target-feature -Xclang -packed-fp32-ops` (__graft_entry__.HIPCC_FLAGS; scripts/build_variant.sh and the micro Makefile
// carry them too).  Round 5 root cause of the weight-gradient kernel's irreproducible aux body (DESIGN.md section 4 "Reproducibility",
// profiles/r05_dw_aux_hazard_table.json, scripts/micro/dw_aux_hazard_micro.hip): hipcc turns `vector * scalar` into v_pk_mul_f32 /
// v_pk_fma_f32 that BROADCAST the scalar out of one dword of an aligned register pair through op_sel (op_sel:[0,1]: both results
// read src1's HIGH dword), and on gfx950 that form occasionally delivers wrong values in lanes 16-31 / 48-63 -- 1 % of the launches
// of the product's own kernel on synthetic operands; every launch once the compiler also forms v_pk_fma_f32 ... op_sel:[0,1,0];
// never with single v_mul_f32, with a packed multiply whose pair holds the scalar in BOTH dwords (no op_sel), or with the target
// feature off.  The whole step costs the same without it (same-box A/B: 8.09 against 8.12 ms of MLP stages).  A per-kernel
// __attribute__((target("no-packed-fp32-ops"))) was tried first and rejected: it stops the inliner at every callee that is not
// always_inline (the HIP headers' __shfl_*, lambdas: 456 real calls in the library).  tests/test_cpu_isa_inflight.py disassembles the
// built library and fails on any v_pk_*_f32 / v_pk_mov_b32 in it, so a build without the flags cannot ship unnoticed.

namespace dh {

constexpr int TM = 64;         // points per tile (= rows of every tile GEMM).  Fixed: the split-bf16 kernels are written for
                               // 64-point tiles (two workgroups per CU); 128 measured 9 % slower with the fp32-MFMA kernels
                               // (DESIGN.md "tried and rejected") and its build switch was removed in round 2
constexpr int MT = TM / 32;    // 32-row m-tiles per tile
static_assert(TM == 64, "tile height");
constexpr int HID = 256;       // hidden width == main-tile width
constexpr int AUXW = 40;       // aux tile logical width (39 embedding / 33 colour extras, zero padded)
constexpr int LDX = 260;       // LDS row stride (floats) of the main tile: 260 % 64 == 4 -> b128 reads conflict free
constexpr int LDA = 52;        // LDS row stride of the aux tile: 48 columns readable as three 16-deep k-chunks (39 / 33 valid,
                               // the rest zero) + 4 pad; 52 % 64 keeps 16 consecutive rows on distinct banks for b128 reads
constexpr int TILE_F = TM * HID;       // floats per native main tile (32768)
constexpr int AUXT_F = TM * 64;        // floats per native aux tile (8192)

constexpr int N_SDF = 9;       // SDF linears lin0..lin8
constexpr int N_COL = 5;       // colour linears lin0..lin4
constexpr int EMB = 39;        // 3 + 6*6
constexpr int VEMB = 27;       // 3 + 6*4
constexpr int CAUX = 33;       // p(3) + view-embed(27) + normal(3)
constexpr int SKIP_OUT = 217;  // lin3 out = 256 - 39
constexpr float SOFTPLUS_BETA = 100.0f;
constexpr float INV_SQRT2 = 0.70710678118654752440f;

// ---------------------------------------------------------------- flat parameter vector
// Order == state_dict order of the python modules: sdf lin0.{bias,weight_g,weight_v} ... lin8, variance,
// colour lin0.{bias,weight_g,weight_v} ... lin4.  802,491 fp32.
struct LinDims { int out, in; };
constexpr LinDims SDF_DIMS[N_SDF] = {{256, 39}, {256, 256}, {256, 256}, {217, 256}, {256, 256},
                                     {256, 256}, {256, 256}, {256, 256}, {257, 256}};
constexpr LinDims COL_DIMS[N_COL] = {{256, 289}, {256, 256}, {256, 256}, {256, 256}, {3, 256}};

struct LinOff { int64_t bias, g, v; };

constexpr int64_t lin_size(LinDims d) { return (int64_t)d.out * 2 + (int64_t)d.out * d.in; }

constexpr int64_t sdf_lin_base(int l) {
    int64_t o = 0;
    for (int i = 0; i < l; ++i) o += lin_size(SDF_DIMS[i]);
    return o;
}
constexpr int64_t VARIANCE_OFF = sdf_lin_base(N_SDF);
constexpr int64_t col_lin_base(int l) {
    int64_t o = VARIANCE_OFF + 1;
    for (int i = 0; i < l; ++i) o += lin_size(COL_DIMS[i]);
    return o;
}
constexpr int64_t N_PARAMS = col_lin_base(N_COL);
static_assert(N_PARAMS == 802491, "parameter count must match SURVEY §8 a12");

constexpr LinOff sdf_off(int l) {
    return LinOff{sdf_lin_base(l), sdf_lin_base(l) + SDF_DIMS[l].out, sdf_lin_base(l) + 2 * (int64_t)SDF_DIMS[l].out};
}
constexpr LinOff col_off(int l) {
    return LinOff{col_lin_base(l), col_lin_base(l) + COL_DIMS[l].out, col_lin_base(l) + 2 * (int64_t)COL_DIMS[l].out};
}

// ---------------------------------------------------------------- packed (MFMA-operand) weight buffer
// B-operand packing for v_mfma_f32_32x32x2_f32: a k-group = 8 consecutive k, an n-tile = 32 consecutive n.
//   float index = ((kg*NT + nt)*64 + lane)*4 + s   holds   M[k = kg*8 + 4*(lane>>5) + s][n = nt*32 + (lane&31)]
// NT = 8 (256 wide) or 2 (64 wide).  "fwd" = M[k=in][n=out] (x W^T), "rev" = M[k=out][n=in] (g W).
struct PackOff {
    int64_t sdf_fwd_main[N_SDF];   // l=1..8 (l=4: 28 k-groups, scaled 1/sqrt2 ; l=8: rows 1..256)
    int64_t sdf_fwd_aux[N_SDF];    // l=0, l=4 (5 k-groups)
    int64_t sdf_rev_main[N_SDF];   // l=1..8 (32 k-groups, NT=8)
    int64_t sdf_rev_aux[N_SDF];    // l=0, l=4 (32 k-groups, NT=2)
    int64_t sdf_bias[N_SDF];       // 256 each (l=8: bias rows 1..256)
    int64_t sdf_w8row0;            // 256
    int64_t sdf_b8_0;              // 4 (first used)
    int64_t col_fwd_main[N_COL];   // l=0..3
    int64_t col_fwd_aux0;          // 5 k-groups
    int64_t col_rev_main[N_COL];   // l=0..3
    int64_t col_rev_aux0;          // NT=2
    int64_t col_bias[N_COL];       // 256 each for l<4
    int64_t col_w4;                // 3*256 row-major
    int64_t col_b4;                // 4
    int64_t rowscale;              // g/||v|| per row of every linear: sdf (9 x 260) then colour (5 x 256)
    int64_t invnorm;               // 1/||v|| per row, same indexing
    int64_t total;
};

constexpr int sdf_kg_main(int l) { return l == 0 ? 0 : (l == 4 ? 28 : 32); }

constexpr PackOff make_pack_off() {
    PackOff p{};
    int64_t o = 0;
    for (int l = 0; l < N_SDF; ++l) {
        p.sdf_fwd_main[l] = o; o += (int64_t)sdf_kg_main(l) * 8 * HID;
        p.sdf_fwd_aux[l] = o;  o += (l == 0 || l == 4) ? (int64_t)AUXW * HID : 0;
        p.sdf_rev_main[l] = o; o += (l >= 1) ? (int64_t)HID * HID : 0;
        p.sdf_rev_aux[l] = o;  o += (l == 0 || l == 4) ? (int64_t)HID * 64 : 0;
        p.sdf_bias[l] = o;     o += HID;
    }
    p.sdf_w8row0 = o; o += HID;
    p.sdf_b8_0 = o; o += 4;
    for (int l = 0; l < N_COL; ++l) {
        p.col_fwd_main[l] = o; o += (l < 4) ? (int64_t)HID * HID : 0;
        p.col_rev_main[l] = o; o += (l < 4) ? (int64_t)HID * HID : 0;
        p.col_bias[l] = o;     o += HID;
    }
    p.col_fwd_aux0 = o; o += (int64_t)AUXW * HID;
    p.col_rev_aux0 = o; o += (int64_t)HID * 64;
    p.col_w4 = o; o += 3 * HID;
    p.col_b4 = o; o += 4;
    p.rowscale = o; o += (int64_t)N_SDF * 260 + (int64_t)N_COL * 256;
    p.invnorm = o; o += (int64_t)N_SDF * 260 + (int64_t)N_COL * 256;
    p.total = (o + 3) / 4 * 4;
    return p;
}
constexpr PackOff PACK = make_pack_off();

// ---------------------------------------------------------------- split-bf16 packed weights (tile16.h), appended after PACK
// bf16x8 (16 B = 4 floats) index ((kc*NT + nt)*3 + piece)*64 + lane; offsets below are FLOAT offsets into `packed`.
struct Pack16Off {
    int64_t sdf_fwd_main[N_SDF];   // l = 1..8 (l = 4: 14 k-chunks, scaled 1/sqrt2; l = 8: rows 1..256)
    int64_t sdf_fwd_aux[N_SDF];    // l = 0, 4 (3 k-chunks: 39 -> 48)
    int64_t sdf_rev_main[N_SDF];   // l = 1..8 (16 k-chunks, NT = 8): M[k = out][n = in]
    int64_t sdf_rev_aux[N_SDF];    // l = 0, 4 (16 k-chunks, NT = 2)
    int64_t col_fwd_main[N_COL];   // l = 0..3
    int64_t col_fwd_aux0;          // 3 k-chunks (33 -> 48)
    int64_t col_rev_main[N_COL];   // l = 0..3
    int64_t col_rev_aux0;          // NT = 2
    int64_t total;                 // end of the whole packed buffer
};
constexpr int sdf_kc_main(int l) { return l == 0 ? 0 : (l == 4 ? 14 : 16); }
constexpr int64_t pack16_floats(int nkc, int nt) { return (int64_t)nkc * nt * 3 * 64 * 4; }
constexpr Pack16Off make_pack16_off() {
    Pack16Off p{};
    int64_t o = PACK.total;
    for (int l = 0; l < N_SDF; ++l) {
        p.sdf_fwd_main[l] = o; o += pack16_floats(sdf_kc_main(l), 8);
        p.sdf_fwd_aux[l] = o;  o += (l == 0 || l == 4) ? pack16_floats(3, 8) : 0;
        p.sdf_rev_main[l] = o; o += (l >= 1) ? pack16_floats(16, 8) : 0;
        p.sdf_rev_aux[l] = o;  o += (l == 0 || l == 4) ? pack16_floats(16, 2) : 0;
    }
    for (int l = 0; l < N_COL; ++l) {
        p.col_fwd_main[l] = o; o += (l < 4) ? pack16_floats(16, 8) : 0;
        p.col_rev_main[l] = o; o += (l < 4) ? pack16_floats(16, 8) : 0;
    }
    p.col_fwd_aux0 = o; o += pack16_floats(3, 8);
    p.col_rev_aux0 = o; o += pack16_floats(16, 2);
    p.total = (o + 3) / 4 * 4;
    return p;
}
constexpr Pack16Off PACK16 = make_pack16_off();

// ---------------------------------------------------------------- register-resident chain (chain_t.hip), appended after PACK16
// `stream`: the SDF network's forward weights as ONE sequence of 132 stages (3 + 16 + 16 + 16 + (14 + 3) + 16 + 16 + 16 k-steps
// of lin0 .. lin7 = the 116 the no-grad kernel cycles through, then 16 of lin8's rows 1..256 for the training forward); a stage = 8 m-tiles x 3 pieces x 64 lanes of bf16x8 = 24 KB, bf16x8 index ((stage*8 + M)*3 + piece)*64
// + lane holding W[32 M + (lane&31)][k], k slot i <-> input feature 16 s + 8 (i/4) + 4 (lane>>5) + (i%4) of the layer's k-step s.
// `bias10`: bias rows of lin0..lin7 (256 each, zero past a layer's width), the effective weight row 0 of lin8, lin8's bias rows 1..256.
constexpr int T_STREAM_STAGES_NOGRAD = 116, T_STREAM_STAGES_TRAIN = 132;
constexpr int64_t PACKT_STREAM_FLOATS = (int64_t)T_STREAM_STAGES_TRAIN * 8 * 3 * 64 * 4;
struct PackTOff { int64_t stream, bias10, total; };
constexpr PackTOff make_packt_off() {
    PackTOff p{};
    p.stream = PACK16.total;
    p.bias10 = p.stream + PACKT_STREAM_FLOATS;
    p.total = p.bias10 + 10 * 256;
    return p;
}
constexpr PackTOff PACKT = make_packt_off();


// ---------------------------------------------------------------- two-piece fp16 packed weights (tile16h.h; round 4), appended after PACKT
// Same job list and operand geometry as PACK16 with TWO fp16 pieces per value instead of three bf16 ones: W' = S_w W with S_w a
// power of two chosen per linear from max |W| (pack.hip: the scaled maximum lies in [8, 16)), hi = fp16(W'), lo = fp16(W' - hi)
// (the unscaled residual; fp16 subnormals carry it below 2^-14).  f16x8 (16 B) index ((kc*NT + nt)*2 + piece)*64 + lane.
// `wabs`: one u32 per linear (sdf lin0..8, colour lin0..4) = the bits of max |W| (written by rowscale_kernel with atomicMax,
// zeroed before it); packers and consumers derive S_w / 1/S_w from it with wscale_from_bits() below.
// `stream` / `bias11`: chain_t.hip's H2 stream (132 stages x 8 m-tiles x 2 pieces = 16 KB per stage) and its bias table: rows
// 0..7 = 16 x bias of lin0..lin7 (the H2 chain carries activations scaled by 16), row 8 = effective weight row 0 of lin8, row 9 =
// lin8's bias rows 1..256 (unscaled), row 10 = 1/S_w of lin0..lin8 in its first nine floats.
constexpr int64_t packh_floats(int nkc, int nt) { return (int64_t)nkc * nt * 2 * 64 * 4; }
constexpr int64_t PACKTH_STREAM_FLOATS = (int64_t)T_STREAM_STAGES_TRAIN * 8 * 2 * 64 * 4;
struct PackHOff {
    int64_t sdf_fwd_main[N_SDF], sdf_fwd_aux[N_SDF], sdf_rev_main[N_SDF], sdf_rev_aux[N_SDF];
    int64_t col_fwd_main[N_COL], col_fwd_aux0, col_rev_main[N_COL], col_rev_aux0;
    int64_t wabs;                  // 16 u32
    int64_t stream, bias11;
    int64_t total;                 // end of the whole packed buffer
};
constexpr PackHOff make_packh_off() {
    PackHOff p{};
    int64_t o = PACKT.total;
    for (int l = 0; l < N_SDF; ++l) {
        p.sdf_fwd_main[l] = o; o += packh_floats(sdf_kc_main(l), 8);
        p.sdf_fwd_aux[l] = o;  o += (l == 0 || l == 4) ? packh_floats(3, 8) : 0;
        p.sdf_rev_main[l] = o; o += (l >= 1) ? packh_floats(16, 8) : 0;
        p.sdf_rev_aux[l] = o;  o += (l == 0 || l == 4) ? packh_floats(16, 2) : 0;
    }
    for (int l = 0; l < N_COL; ++l) {
        p.col_fwd_main[l] = o; o += (l < 4) ? packh_floats(16, 8) : 0;
        p.col_rev_main[l] = o; o += (l < 4) ? packh_floats(16, 8) : 0;
    }
    p.col_fwd_aux0 = o; o += packh_floats(3, 8);
    p.col_rev_aux0 = o; o += packh_floats(16, 2);
    p.wabs = o; o += 16;
    p.stream = o; o += PACKTH_STREAM_FLOATS;
    p.bias11 = o; o += 11 * 256;
    p.total = (o + 3) / 4 * 4;
    return p;
}
constexpr PackHOff PACKH = make_packh_off();
constexpr float H2_RANGE = 65504.f;  // the largest SCALED magnitude whose hi piece is finite (fp16 max; the convert rounds to nearest)
constexpr float H2_XS = 16.f;      // static power-of-two scale of O(1) operands (softplus / ReLU activations, embeddings, features)

}  // namespace dh
