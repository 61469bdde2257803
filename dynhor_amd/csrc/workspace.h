// Workspace carving for one render call (caller-allocated, SURVEY §8b ownership rule).  All saved tiles are
// "native" accumulator-layout tiles (tile.h); offsets in floats, every block 16-byte aligned.
#pragma once
#include "layout.h"

namespace dh {

constexpr int N_TILE_PART = 20;
constexpr int DW_G = 256;         // split-K factor of the weight-gradient GEMMs = one persistent workgroup per CU
constexpr int DW_NS = 64;         // split factor of the tile-partial reduction
int64_t dw_slab_floats(int G);   // per-tile partial-sum slots of 256 floats (bias grads, lin8 row 0, colour lin4)

// absmax: per-launch maxima of the saved-tile classes whose scale the two-piece fp16 weight-gradient kernel needs (tile16h.h):
// one u32 (the fp32 bits of the maximum |value|) per class, classes ABSMAX_STRIDE words apart; zeroed by every training forward
// (launch_sdf_fwd_train), raised with atomicMax by the SPLIT_F16 kernels that write (or, for act, first read) the tiles.
// Round 5: the softplus activations act[0..7] share ABSMAX_ACT (posted by sdf_grad_h_kernel, which reads them all), the colour extras
// [p, embed(view), n] have ABSMAX_CAUX (the normal is unbounded) -- only the embedding tile eaux is still carried at the constant
// H2_XS = 16 by the weight-gradient kernel (sin / cos and a point inside the unit sphere: bounded by construction).  The
// register-resident forward chain carries its activations at the constant H2_XS = 16 too (fp16 overflow beyond |act| = 4094):
// ABSMAX_ACT is what tells (dh_range_words; the Runner raises at its report iterations).
// ABSMAX_TAG: ABSMAX_TAG_F16 once the SPLIT_F16 training forward of this step has run (every training forward clears the table
// first): a SPLIT_F16 weight-gradient launch behind a forward of another arithmetic finds no tag and poisons its slabs with NaN
// instead of scaling by stale words (include/dynhor_hip.h "ONE ARITHMETIC PER STEP").
enum : int { ABSMAX_ASAVE = 0, ABSMAX_ZBAR = 8, ABSMAX_TSAVE = 16, ABSMAX_T0AUX = 23, ABSMAX_FEATBAR = 24, ABSMAX_CZBAR = 25,
             ABSMAX_CACT = 29, ABSMAX_FEAT = 33, ABSMAX_ACT = 34, ABSMAX_CAUX = 35, ABSMAX_TAG = 36, ABSMAX_N = 37 };
constexpr unsigned ABSMAX_TAG_F16 = 0x00F16F16u;
constexpr int ABSMAX_STRIDE = 64;         // class slot c lives 64 words (256 B) from the next
constexpr int ABSMAX_FLOATS = 64 * ABSMAX_STRIDE;
// tmax: PER-TILE maxima of the heavy-tailed classes (adjoints and tangents: a few sample points near the surface carry almost
// everything): [TMAX_N][ntiles] u32.  The two-piece fp16 weight-gradient kernel scales such an operand tile by tile (a launch-wide
// scale set by one outlier would push the typical tile into fp16's subnormals: measured 1.7e-4 on the gradient) and divides the
// tile's other operand by the same power of two, so that every product still carries one launch-wide scale.  Written (plain
// stores, every tile once) by the kernels that write the tiles.
enum : int { TMAX_ZBAR = 0, TMAX_FEATBAR = 8, TMAX_CZBAR = 9, TMAX_TSAVE = 13, TMAX_T0AUX = 20, TMAX_N = 21 };

struct Workspace {
    int64_t ntiles;
    float* base;
    float* absmax;  // [64][64] u32, see above
    float* tmax;    // [TMAX_N][nt] u32, see above
    // forward (saved for backward)
    float* act;     // [8][nt][TILE_F]   inputs of SDF layers 1..8 (post-softplus)
    float* eaux;    // [nt][AUXT_F]      positional embedding (aux native)
    float* feat;    // [nt][TILE_F]      lin8 rows 1..256
    float* asave;   // [8][nt][TILE_F]   a_l = u_{l+1} * sigma'(z_l) of the input-gradient pass
    float* cact;    // [4][nt][TILE_F]   colour post-ReLU activations (inputs of colour layers 1..4)
    float* caux;    // [nt][AUXT_F]      colour extras [p, embed(view), n]
    // backward
    float* featbar; // [nt][TILE_F]      d loss / d feat
    float* tsave;   // [7][nt][TILE_F]   tangents t_1..t_7 (inputs of the tangent GEMM of layers 1..7)
    float* t0aux;   // [nt][AUXT_F]      t_0 = J_e nbar
    float* rsave;   // [8][nt][TILE_F]   second-order contribution to zbar_l
    float* zbar;    // [8][nt][TILE_F]   d loss / d z_l, l = 0..7
    float* czbar;   // [4][nt][TILE_F]   colour d loss / d z_l
    float* tpart;   // [nt][N_TILE_PART][256]
    float* tred;    // [DW_NS][N_TILE_PART][256]
    float* slabs;   // dW split-K slabs (dw.hip)
    float* gesave;  // [nt*TM][40]       embedding-gradient vector ge of the normal pass (pose refinement only: dh_sdf_gradient save = 2)
    int64_t infer_floats, fwd_floats, total_floats;   // forward-only (no saves for backward) / forward / everything
};

inline Workspace carve_workspace(float* base, int64_t npts) {
    Workspace w{};
    const int64_t nt = (npts + TM - 1) / TM;
    w.ntiles = nt;
    w.base = base;
    int64_t o = 0;
    auto take = [&](int64_t n) { float* p = base ? base + o : nullptr; o += n; return p; };
    w.absmax = take(ABSMAX_FLOATS);
    w.tmax = take((TMAX_N * nt + 3) / 4 * 4);
    w.act = take(8 * nt * TILE_F);
    w.eaux = take(nt * AUXT_F);
    w.feat = take(nt * TILE_F);
    w.infer_floats = o;
    w.asave = take(8 * nt * TILE_F);
    w.cact = take(4 * nt * TILE_F);
    w.caux = take(nt * AUXT_F);
    w.fwd_floats = o;
    w.featbar = take(nt * TILE_F);
    w.tsave = take(7 * nt * TILE_F);
    w.t0aux = take(nt * AUXT_F);
    w.rsave = take(8 * nt * TILE_F);
    w.zbar = take(8 * nt * TILE_F);
    w.czbar = take(4 * nt * TILE_F);
    w.tpart = take(nt * N_TILE_PART * 256);
    w.tred = take((int64_t)DW_NS * N_TILE_PART * 256);
    w.slabs = take(dw_slab_floats(DW_G));
    w.gesave = take(nt * TM * 40);
    w.total_floats = o;
    return w;
}

}  // namespace dh
