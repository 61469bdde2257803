// Hash-grid model family (BASELINE.json configs[3]): fused hash-grid encoding + 64-wide SDF MLP with
// central-finite-difference normals, and the SH-4 colour MLP.  Gather-bound: one thread per sample point, the ~10 k
// small weights of both MLPs staged whole into LDS (41 KB) and read as wave-uniform broadcasts; the MLP arithmetic is
// VALU (a 35x64 + 64x13 network is ~3 kFMA per evaluation -- nothing for MFMA to win against the 128 table gathers).
// Algorithm: oracle/hashgrid_oracle.py (parity unpinned, see its header).
#include "tile.h"
#include "kernels.h"
#include "hash_layout.h"
#include "hashgrid_dev.h"

namespace dh {

// ---------------------------------------------------------------- small-weight packing (weight-norm)
struct HashLinDesc { int64_t b, g, v; int out, in, ld, dst, dstb, rowbase; };

__global__ __launch_bounds__(64) void hash_pack_kernel(const float* __restrict__ params, float* __restrict__ hp,
                                                       HashParamOff P) {
    const HashLinDesc D[5] = {{P.g0_b, P.g0_g, P.g0_v, 64, HM_GIN, 36, HP_G0, HP_G0B, HP_ROW_G0},
                              {P.g1_b, P.g1_g, P.g1_v, HM_GOUT, 64, 64, HP_G1, HP_G1B, HP_ROW_G1},
                              {P.c0_b, P.c0_g, P.c0_v, 64, HM_CIN, 32, HP_C0, HP_C0B, HP_ROW_C0},
                              {P.c1_b, P.c1_g, P.c1_v, 64, 64, 64, HP_C1, HP_C1B, HP_ROW_C1},
                              {P.c2_b, P.c2_g, P.c2_v, 3, 64, 64, HP_C2, HP_C2B, HP_ROW_C2}};
    const HashLinDesc L = D[blockIdx.y];
    const int row = blockIdx.x, lane = threadIdx.x;
    const int padded_rows = (blockIdx.y == 1) ? 16 : (blockIdx.y == 4 ? 4 : L.out);
    if (row >= padded_rows) return;
    if (row >= L.out) {                                   // zero padding rows
        for (int k = lane; k < L.ld; k += 64) hp[L.dst + row * L.ld + k] = 0.f;
        if (lane == 0) hp[L.dstb + row] = 0.f;
        return;
    }
    const float* v = params + L.v + (int64_t)row * L.in;
    float s = 0.f;
    for (int k = lane; k < L.in; k += 64) s += v[k] * v[k];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float inv = 1.f / sqrtf(s);
    const float rs = params[L.g + row] * inv;
    for (int k = lane; k < L.ld; k += 64) hp[L.dst + row * L.ld + k] = k < L.in ? rs * v[k] : 0.f;
    if (lane == 0) {
        hp[L.dstb + row] = params[L.b + row];
        hp[HP_RS + L.rowbase + row] = rs;
        hp[HP_INV + L.rowbase + row] = inv;
    }
}

__device__ __forceinline__ void stage_weights(float* lds, const float* __restrict__ hp) {
    for (int i = threadIdx.x; i < HP_WEIGHTS; i += blockDim.x) lds[i] = hp[i];
    __syncthreads();
}

__device__ __forceinline__ void hg_cell(const HashLevels& H, int l, const float (&x01)[3], uint32_t (&g)[3], float (&w)[3]) {
    const float s = H.scale[l];
    DH_UNROLL for (int c = 0; c < 3; ++c) {
        const float pos = x01[c] * s + 0.5f;
        const float f = floorf(pos);
        w[c] = pos - f;
        g[c] = (uint32_t)(int)f;
    }
}

// ---------------------------------------------------------------- geometry network, one evaluation
// x: world position; radius: scene box half-size (x01 = (x + r) / 2r).  FULL: all 13 outputs, else sdf only.
template <bool FULL>
__device__ __forceinline__ void geo_eval(const HashLevels& H, const float* __restrict__ table, const float* W,
                                         const float (&x)[3], float radius, float (&out)[HM_GOUT]) {
    float in[36];
    float x01[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) { x01[c] = (x[c] + radius) / (2.f * radius); in[c] = x01[c] * 2.f - 1.f; }
    DH_UNROLL for (int l = 0; l < HG_L; ++l) hg_encode_level(H, table, l, x01, in[3 + 2 * l], in[4 + 2 * l]);
    in[35] = 0.f;
    DH_UNROLL for (int c = 0; c < HM_GOUT; ++c) out[c] = W[HP_G1B + c];
    for (int j = 0; j < HM_HID; ++j) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_G0 + j * 36);
        float a = W[HP_G0B + j];
        DH_UNROLL for (int k4 = 0; k4 < 9; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], in[4 * k4], a); a = fmaf(w[1], in[4 * k4 + 1], a);
            a = fmaf(w[2], in[4 * k4 + 2], a); a = fmaf(w[3], in[4 * k4 + 3], a);
        }
        const float h = softplus100(a);
        out[0] = fmaf(W[HP_G1 + j], h, out[0]);
        if (FULL) { DH_UNROLL for (int c = 1; c < HM_GOUT; ++c) out[c] = fmaf(W[HP_G1 + c * 64 + j], h, out[c]); }
    }
}

// sdf only (hierarchical up-sampling evaluations)
__global__ __launch_bounds__(256) void hash_sdf_nograd_kernel(HashLevels H, const float* __restrict__ table,
                                                              const float* __restrict__ hp, const float* __restrict__ pts,
                                                              int64_t n, float radius, float* __restrict__ sdf) {
    __shared__ __attribute__((aligned(16))) float W[HP_WEIGHTS + 4];
    stage_weights(W, hp);
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float x[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    float out[HM_GOUT];
    geo_eval<false>(H, table, W, x, radius, out);
    sdf[p] = out[0];
}

// training / rendering forward: sdf, feature (all 13 outputs), finite-difference gradient (eps: oracle fd_eps = 1e-3).
// Two kernels so the gathers run at high occupancy and are shared between a point's E = 7 evaluations:
//   1. hash_encode7_kernel -- one thread per (point, level), grid.y = level (a wave = 64 consecutive samples of a ray
//      at one level).  The centre cell's 8 corners are gathered once; a +-eps evaluation re-uses them when it stays in
//      the cell and re-uses the shared face when it lands in the neighbour (4 new gathers), so a point costs ~216
//      8-byte gathers instead of 7 x 128.  Writes the encodings feature-major into rows 3..34 of GIN [36][lde]
//      (row r = e n + p) -- the same rows the backward's weight-gradient GEMM reads, so nothing is gathered twice.
//   2. hash_geo_mlp_fwd_kernel -- one thread per point: the 35 -> 64 -> 13 MLP on the 7 encoded inputs (weights in LDS).
// `lvl` = the level's entries, indices are level-local.  (Round 2 tried "LDS-resident grid tiles" as BASELINE.json configs[3] words
// it: levels 0 and 1 -- 16^3 and 23^3 entries = 32 / 95 KiB, the only ones that fit -- launched apart with persistent workgroups
// that copy the level's slice into LDS and gather from there.  At 262,144 points the staging costs more than the L1 / L2 hits it
// replaces: hash_geo_forward 0.48 vs 0.44 ms on one box, so the single launch below stays)
__device__ __forceinline__ uint32_t hg_index_local(const HashLevels& H, int l, uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t res = H.res[l];
    return H.dense[l] ? ((x + y * res + z * res * res) % H.size[l]) : (((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (HG_T - 1));
}
__device__ __forceinline__ void encode7_point(const HashLevels& H, const float2* lvl, int l, int64_t p,
                                              const float* __restrict__ pts, int64_t n, float radius, float eps,
                                              float* __restrict__ gin, int64_t lde) {
    const float x[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    float x01[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) x01[c] = (x[c] + radius) / (2.f * radius);
    uint32_t g0[3];
    float w0[3];
    hg_cell(H, l, x01, g0, w0);
    float2 cf[8];
    DH_UNROLL for (int corner = 0; corner < 8; ++corner)
        cf[corner] = lvl[hg_index_local(H, l, g0[0] + (corner & 1), g0[1] + ((corner >> 1) & 1), g0[2] + (corner >> 2))];
    float* o0 = gin + (int64_t)(3 + 2 * l) * lde + p;
    float* o1 = o0 + lde;
    {
        float a0 = 0.f, a1 = 0.f;
        DH_UNROLL for (int corner = 0; corner < 8; ++corner) {
            const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
            const float wt = (dx ? w0[0] : 1.f - w0[0]) * (dy ? w0[1] : 1.f - w0[1]) * (dz ? w0[2] : 1.f - w0[2]);
            a0 = fmaf(wt, cf[corner].x, a0);
            a1 = fmaf(wt, cf[corner].y, a1);
        }
        o0[0] = a0; o1[0] = a1;
    }
    const float s = H.scale[l];
    DH_UNROLL for (int e = 1; e < 7; ++e) {
        const int a = (e - 1) >> 1, b = (a + 1) % 3, c = (a + 2) % 3;
        const float xe = x[a] + ((e - 1) & 1 ? -eps : eps);
        const float pos = ((xe + radius) / (2.f * radius)) * s + 0.5f;
        const float fl = floorf(pos);
        const float wa = pos - fl;
        const uint32_t ga = (uint32_t)(int)fl;
        const int delta = (int)(ga - g0[a]);
        float a0 = 0.f, a1 = 0.f;
        DH_UNROLL for (int ba = 0; ba < 2; ++ba) {
            const int t = delta + ba;                    // plane position relative to the centre cell: 0 / 1 = already held
            const bool shared = (unsigned)t < 2u;
            const float wpa = ba ? wa : 1.f - wa;
            float2 v[4];
            DH_UNROLL for (int q = 0; q < 4; ++q) {
                const int c0 = ((q & 1) << b) | ((q >> 1) << c), c1 = c0 | (1 << a);
                v[q].x = t ? cf[c1].x : cf[c0].x;
                v[q].y = t ? cf[c1].y : cf[c0].y;
            }
            if (!shared) {
                DH_UNROLL for (int q = 0; q < 4; ++q) {
                    uint32_t cc[3];
                    cc[a] = ga + ba; cc[b] = g0[b] + (q & 1); cc[c] = g0[c] + (q >> 1);
                    v[q] = lvl[hg_index_local(H, l, cc[0], cc[1], cc[2])];
                }
            }
            DH_UNROLL for (int q = 0; q < 4; ++q) {
                const float wt = wpa * ((q & 1) ? w0[b] : 1.f - w0[b]) * ((q >> 1) ? w0[c] : 1.f - w0[c]);
                a0 = fmaf(wt, v[q].x, a0);
                a1 = fmaf(wt, v[q].y, a1);
            }
        }
        o0[(int64_t)e * n] = a0; o1[(int64_t)e * n] = a1;
    }
}

// Packed rays (occupancy-grid sampler): the number of samples of an iteration is known only on the device.  The launches are
// sized and the workspace is laid out for a host-known CAPACITY n; `n_act` (device pointer, may be null) holds how many of the
// n rows are real -- threads past it leave, so nothing waits for a device -> host read of the count.
__device__ __forceinline__ int64_t active_rows(const int64_t* __restrict__ n_act, int64_t n) {
    if (!n_act) return n;
    const int64_t a = *n_act;
    return a < n ? (a < 0 ? 0 : a) : n;
}

// levels l_base + blockIdx.y, gathers from global memory (one thread per point)
__global__ __launch_bounds__(256) void hash_encode7_kernel(HashLevels H, const float* __restrict__ table,
                                                           const float* __restrict__ pts, int64_t n, float radius, float eps,
                                                           float* __restrict__ gin, int64_t lde, int l_base,
                                                           const int64_t* __restrict__ n_act) {
    const int l = l_base + blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= active_rows(n_act, n)) return;
    encode7_point(H, reinterpret_cast<const float2*>(table) + H.offset[l], l, p, pts, n, radius, eps, gin, lde);
}

// the geometry MLP on one encoded input (rows 3..34 of GIN); FULL: all 13 outputs, else sdf only
template <bool FULL>
__device__ __forceinline__ void geo_mlp(const float* W, const float (&in)[36], float (&out)[HM_GOUT]) {
    DH_UNROLL for (int c = 0; c < HM_GOUT; ++c) out[c] = W[HP_G1B + c];
    for (int j = 0; j < HM_HID; ++j) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_G0 + j * 36);
        float a = W[HP_G0B + j];
        DH_UNROLL for (int k4 = 0; k4 < 9; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], in[4 * k4], a); a = fmaf(w[1], in[4 * k4 + 1], a);
            a = fmaf(w[2], in[4 * k4 + 2], a); a = fmaf(w[3], in[4 * k4 + 3], a);
        }
        const float h = softplus100(a);
        out[0] = fmaf(W[HP_G1 + j], h, out[0]);
        if (FULL) { DH_UNROLL for (int c = 1; c < HM_GOUT; ++c) out[c] = fmaf(W[HP_G1 + c * 64 + j], h, out[c]); }
    }
}

__device__ __forceinline__ void load_geo_input(const float* __restrict__ gin, int64_t lde, int64_t row, const float (&xe)[3],
                                               float radius, float (&in)[36]) {
    DH_UNROLL for (int c = 0; c < 3; ++c) in[c] = ((xe[c] + radius) / (2.f * radius)) * 2.f - 1.f;
    DH_UNROLL for (int k = 3; k < 35; ++k) in[k] = gin[(int64_t)k * lde + row];
    in[35] = 0.f;
}

__global__ __launch_bounds__(256) void hash_geo_mlp_fwd_kernel(const float* __restrict__ hp, const float* __restrict__ pts,
                                                               int64_t n, float radius, float eps,
                                                               const float* __restrict__ gin, int64_t lde,
                                                               float* __restrict__ sdf, float* __restrict__ feat,
                                                               float* __restrict__ grad, const int64_t* __restrict__ n_act) {
    __shared__ __attribute__((aligned(16))) float W[HP_WEIGHTS + 4];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t na = active_rows(n_act, n);
    if ((int64_t)blockIdx.x * 256 >= na) return;          // the whole workgroup is past the end: skip the weight staging too
    stage_weights(W, hp);
    if (p >= na) return;
    const float x[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    float in[36], out[HM_GOUT];
    load_geo_input(gin, lde, p, x, radius, in);
    geo_mlp<true>(W, in, out);
    sdf[p] = out[0];
    DH_UNROLL for (int c = 0; c < HM_FEAT; ++c) feat[p * HM_FEAT + c] = out[c];
    DH_UNROLL for (int i = 0; i < 3; ++i) {
        float sd[2];
        DH_UNROLL for (int sg = 0; sg < 2; ++sg) {
            float xe[3] = {x[0], x[1], x[2]};
            xe[i] += sg ? -eps : eps;
            load_geo_input(gin, lde, (int64_t)(1 + 2 * i + sg) * n + p, xe, radius, in);
            float o[HM_GOUT];
            geo_mlp<false>(W, in, o);
            sd[sg] = o[0];
        }
        grad[p * 3 + i] = (sd[0] - sd[1]) * (0.5f / eps);
    }
}

// ---------------------------------------------------------------- SH-4 colour network
__device__ __forceinline__ void sh4_eval(const float (&d)[3], float (&y)[16]) {
    const float x = d[0], yy_ = d[1], z = d[2];
    const float xx = x * x, yy = yy_ * yy_, zz = z * z, xy = x * yy_, yz = yy_ * z, xz = x * z;
    y[0] = 0.28209479177387814f;
    y[1] = -0.48860251190291987f * yy_; y[2] = 0.48860251190291987f * z; y[3] = -0.48860251190291987f * x;
    y[4] = 1.0925484305920792f * xy; y[5] = -1.0925484305920792f * yz;
    y[6] = 0.94617469575755997f * zz - 0.31539156525251999f; y[7] = -1.0925484305920792f * xz;
    y[8] = 0.54627421529603959f * (xx - yy);
    y[9] = 0.59004358992664352f * yy_ * (-3.f * xx + yy); y[10] = 2.8906114426405538f * xy * z;
    y[11] = 0.45704579946446572f * yy_ * (1.f - 5.f * zz); y[12] = 0.3731763325901154f * z * (5.f * zz - 3.f);
    y[13] = 0.45704579946446572f * x * (1.f - 5.f * zz); y[14] = 1.4453057213202769f * z * (xx - yy);
    y[15] = 0.59004358992664352f * x * (-xx + 3.f * yy);
}

__global__ __launch_bounds__(256) void sh_color_fwd_kernel(const float* __restrict__ hp, const float* __restrict__ feat,
                                                           const float* __restrict__ normals, const float* __restrict__ dirs,
                                                           int n_per_ray, int64_t n, float* __restrict__ color,
                                                           const int64_t* __restrict__ n_act) {
    __shared__ __attribute__((aligned(16))) float W[HP_WEIGHTS + 4];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t na = active_rows(n_act, n);
    if ((int64_t)blockIdx.x * 256 >= na) return;
    stage_weights(W, hp);
    if (p >= na) return;
    float in[HM_CIN];
    DH_UNROLL for (int c = 0; c < HM_FEAT; ++c) in[c] = feat[p * HM_FEAT + c];
    const int64_t ray = p / n_per_ray;
    const float d[3] = {dirs[ray * 3], dirs[ray * 3 + 1], dirs[ray * 3 + 2]};
    float y[16];
    sh4_eval(d, y);
    DH_UNROLL for (int c = 0; c < 16; ++c) in[HM_FEAT + c] = y[c];
    DH_UNROLL for (int c = 0; c < 3; ++c) in[HM_FEAT + 16 + c] = normals[p * 3 + c];
    float h1[HM_HID];
    DH_UNROLL for (int j = 0; j < HM_HID; ++j) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C0 + j * HM_CIN);
        float a = W[HP_C0B + j];
        DH_UNROLL for (int k4 = 0; k4 < HM_CIN / 4; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], in[4 * k4], a); a = fmaf(w[1], in[4 * k4 + 1], a);
            a = fmaf(w[2], in[4 * k4 + 2], a); a = fmaf(w[3], in[4 * k4 + 3], a);
        }
        h1[j] = fmaxf(a, 0.f);
    }
    float o[3] = {W[HP_C2B], W[HP_C2B + 1], W[HP_C2B + 2]};
    for (int j2 = 0; j2 < HM_HID; ++j2) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C1 + j2 * 64);
        float a = W[HP_C1B + j2];
        DH_UNROLL for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], h1[4 * k4], a); a = fmaf(w[1], h1[4 * k4 + 1], a);
            a = fmaf(w[2], h1[4 * k4 + 2], a); a = fmaf(w[3], h1[4 * k4 + 3], a);
        }
        const float h2 = fmaxf(a, 0.f);
        DH_UNROLL for (int c = 0; c < 3; ++c) o[c] = fmaf(W[HP_C2 + c * 64 + j2], h2, o[c]);
    }
    DH_UNROLL for (int c = 0; c < 3; ++c) color[p * 3 + c] = 1.f / (1.f + __expf(-o[c]));
}

// ================================================================ backward
// Workspace (floats, per point-count n; E = 7 geometry evaluations: centre, then +-eps per axis).  Everything is
// FEATURE-MAJOR ([feature][row], row stride ld = rows rounded up to 8) so the per-point kernels (lane = row) write
// full 256-B segments and small_dw_kernel reads 16-B row quads:
//   colour  : DO [4][ldn] | H2 [64][ldn] | DZ2 [64][ldn] | H1 [64][ldn] | DZ1 [64][ldn] | CIN [32][ldn]
//   geometry: X01 [3][lde] | DIN [16 levels][E n] float2 | DOUT [16][lde] | HH [64][lde] | DA [64][lde] | GIN [36][lde]
//   partial weight-gradient slabs
//   FIX: hashgrid_entries() x 2 int64 (+ one flag line) -- the fixed-point accumulators of the reproducible table scatter
// The per-point kernels recompute the (tiny) forward and leave each layer's input and pre-activation adjoint; the
// weight gradients are then plain  dW = sum_p dz[p]^T in[p]  reductions done by small_dw_kernel on MFMA.
struct HashWs {
    int64_t d_o, h2, dz2, h1, dz1, cin, x01, din, dout, hh, da, gin, slabs, fix, total;
    int64_t ldn, lde;
};
constexpr int HW_E = 7;
constexpr int HW_SLABS = 256;                    // partial-sum slabs of the weight-gradient reductions
constexpr int HW_DW_FLOATS = 64 * 64 + 64;       // one job's slab: padded [64][64] tile + column sums
constexpr int HW_JOBS = 5;
inline HashWs make_hash_ws(int64_t n) {
    HashWs w{};
    int64_t o = 0;
    auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 63) / 64 * 64; return r; };
    w.ldn = (n + 7) / 8 * 8;
    w.lde = (HW_E * n + 7) / 8 * 8;
    w.d_o = take(w.ldn * 4); w.h2 = take(w.ldn * 64); w.dz2 = take(w.ldn * 64); w.h1 = take(w.ldn * 64);
    w.dz1 = take(w.ldn * 64); w.cin = take(w.ldn * 32);
    w.x01 = take(w.lde * 3); w.din = take(HW_E * n * 32); w.dout = take(w.lde * 16); w.hh = take(w.lde * 64);
    w.da = take(w.lde * 64); w.gin = take(w.lde * 36);
    w.slabs = take((int64_t)HW_JOBS * HW_SLABS * HW_DW_FLOATS);
    w.total = o;                                   // hash_workspace_floats() appends the reduced slabs, then FIX
    w.fix = (o + (int64_t)HW_JOBS * HW_DW_FLOATS + 63) / 64 * 64;
    return w;
}

// feature-major store of a register row: element k of row r -> base[k * ld + r] (coalesced across lanes)
template <int N>
__device__ __forceinline__ void store_col(float* base, int64_t ld, int64_t r, const float (&v)[N], int cnt = N) {
    DH_UNROLL for (int k = 0; k < N; ++k) if (k < cnt) base[k * ld + r] = v[k];
}

// colour network adjoint.  d_normals is read-modify-written (the render-scan adjoint is already in it).
__global__ __launch_bounds__(256) void sh_color_bwd_kernel(const float* __restrict__ hp, const float* __restrict__ feat,
                                                           const float* __restrict__ normals, const float* __restrict__ dirs,
                                                           const float* __restrict__ d_color, int n_per_ray, int64_t n,
                                                           float* __restrict__ ws, HashWs O, float* __restrict__ d_feat,
                                                           float* __restrict__ d_normals, const int64_t* __restrict__ n_act) {
    __shared__ __attribute__((aligned(16))) float W[HP_WEIGHTS + 4];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t na = active_rows(n_act, n);
    if ((int64_t)blockIdx.x * 256 >= na) return;
    stage_weights(W, hp);
    if (p >= na) return;
    float in[HM_CIN];
    DH_UNROLL for (int c = 0; c < HM_FEAT; ++c) in[c] = feat[p * HM_FEAT + c];
    const int64_t ray = p / n_per_ray;
    const float d[3] = {dirs[ray * 3], dirs[ray * 3 + 1], dirs[ray * 3 + 2]};
    {
        float y[16];
        sh4_eval(d, y);
        DH_UNROLL for (int c = 0; c < 16; ++c) in[HM_FEAT + c] = y[c];
    }
    DH_UNROLL for (int c = 0; c < 3; ++c) in[HM_FEAT + 16 + c] = normals[p * 3 + c];
    store_col(ws + O.cin, O.ldn, p, in);
    float h1[HM_HID];
    DH_UNROLL for (int j = 0; j < HM_HID; ++j) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C0 + j * HM_CIN);
        float a = W[HP_C0B + j];
        DH_UNROLL for (int k4 = 0; k4 < HM_CIN / 4; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], in[4 * k4], a); a = fmaf(w[1], in[4 * k4 + 1], a);
            a = fmaf(w[2], in[4 * k4 + 2], a); a = fmaf(w[3], in[4 * k4 + 3], a);
        }
        h1[j] = fmaxf(a, 0.f);
    }
    store_col(ws + O.h1, O.ldn, p, h1);
    // pass 1 over the second layer: the outputs
    float o[3] = {W[HP_C2B], W[HP_C2B + 1], W[HP_C2B + 2]};
    for (int j2 = 0; j2 < HM_HID; ++j2) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C1 + j2 * 64);
        float a = W[HP_C1B + j2];
        DH_UNROLL for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], h1[4 * k4], a); a = fmaf(w[1], h1[4 * k4 + 1], a);
            a = fmaf(w[2], h1[4 * k4 + 2], a); a = fmaf(w[3], h1[4 * k4 + 3], a);
        }
        const float h2 = fmaxf(a, 0.f);
        DH_UNROLL for (int c = 0; c < 3; ++c) o[c] = fmaf(W[HP_C2 + c * 64 + j2], h2, o[c]);
    }
    float dob[4];
    DH_UNROLL for (int c = 0; c < 3; ++c) {
        const float col = 1.f / (1.f + __expf(-o[c]));
        dob[c] = d_color[p * 3 + c] * col * (1.f - col);
    }
    dob[3] = 0.f;
    store_col(ws + O.d_o, O.ldn, p, dob);
    // pass 2: recompute h2 row by row, its adjoint, and pull back onto h1
    float dh1[HM_HID];
    DH_UNROLL for (int k = 0; k < HM_HID; ++k) dh1[k] = 0.f;
    for (int j2 = 0; j2 < HM_HID; ++j2) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C1 + j2 * 64);
        float a = W[HP_C1B + j2];
        DH_UNROLL for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 w = wr[k4];
            a = fmaf(w[0], h1[4 * k4], a); a = fmaf(w[1], h1[4 * k4 + 1], a);
            a = fmaf(w[2], h1[4 * k4 + 2], a); a = fmaf(w[3], h1[4 * k4 + 3], a);
        }
        const float h2 = fmaxf(a, 0.f);
        float dz2 = W[HP_C2 + j2] * dob[0] + W[HP_C2 + 64 + j2] * dob[1] + W[HP_C2 + 128 + j2] * dob[2];
        dz2 = a > 0.f ? dz2 : 0.f;
        ws[O.h2 + j2 * O.ldn + p] = h2;
        ws[O.dz2 + j2 * O.ldn + p] = dz2;
        DH_UNROLL for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 w = wr[k4];
            dh1[4 * k4] = fmaf(w[0], dz2, dh1[4 * k4]); dh1[4 * k4 + 1] = fmaf(w[1], dz2, dh1[4 * k4 + 1]);
            dh1[4 * k4 + 2] = fmaf(w[2], dz2, dh1[4 * k4 + 2]); dh1[4 * k4 + 3] = fmaf(w[3], dz2, dh1[4 * k4 + 3]);
        }
    }
    DH_UNROLL for (int k = 0; k < HM_HID; ++k) dh1[k] = h1[k] > 0.f ? dh1[k] : 0.f;     // now dz1
    store_col(ws + O.dz1, O.ldn, p, dh1);
    float din[HM_CIN];
    DH_UNROLL for (int k = 0; k < HM_CIN; ++k) din[k] = 0.f;
    DH_UNROLL for (int j = 0; j < HM_HID; ++j) {
        const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_C0 + j * HM_CIN);
        DH_UNROLL for (int k4 = 0; k4 < HM_CIN / 4; ++k4) {
            if (k4 >= 4 && k4 < 7) continue;                      // SH inputs: the direction is not differentiated
            const f32x4 w = wr[k4];
            din[4 * k4] = fmaf(w[0], dh1[j], din[4 * k4]); din[4 * k4 + 1] = fmaf(w[1], dh1[j], din[4 * k4 + 1]);
            din[4 * k4 + 2] = fmaf(w[2], dh1[j], din[4 * k4 + 2]); din[4 * k4 + 3] = fmaf(w[3], dh1[j], din[4 * k4 + 3]);
        }
    }
    DH_UNROLL for (int c = 0; c < HM_FEAT; ++c) d_feat[p * HM_FEAT + c] = din[c];
    DH_UNROLL for (int c = 0; c < 3; ++c) d_normals[p * 3 + c] += din[HM_FEAT + 16 + c];
}

// geometry adjoint: E = 7 evaluations per point (centre with the feature/sdf cotangent, +-eps per axis with
// +-d_grad * 0.5/eps on the sdf output).  Writes the rows small_dw_kernel and the table scatter consume.
__global__ __launch_bounds__(256) void hash_geo_bwd_kernel(const float* __restrict__ hp, const float* __restrict__ pts,
                                                           const float* __restrict__ d_sdf, const float* __restrict__ d_feat,
                                                           const float* __restrict__ d_grad, int64_t n, float radius,
                                                           float eps, float* __restrict__ ws, HashWs O,
                                                           const int64_t* __restrict__ n_act) {
    __shared__ __attribute__((aligned(16))) float W[HP_WEIGHTS + 4];
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t na = active_rows(n_act, n);
    if ((int64_t)blockIdx.x * 256 >= na) return;
    stage_weights(W, hp);
    if (p >= na) return;
    const float x[3] = {pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2]};
    for (int e = 0; e < HW_E; ++e) {
        float xe[3] = {x[0], x[1], x[2]};
        float dout[16];
        DH_UNROLL for (int c = 0; c < 16; ++c) dout[c] = 0.f;
        if (e == 0) {
            DH_UNROLL for (int c = 0; c < HM_FEAT; ++c) dout[c] = d_feat[p * HM_FEAT + c];
            dout[0] += d_sdf[p];
        } else {
            const int axis = (e - 1) >> 1;
            const float sgn = (e - 1) & 1 ? -1.f : 1.f;
            DH_UNROLL for (int c = 0; c < 3; ++c) if (c == axis) xe[c] += sgn * eps;
            dout[0] = sgn * d_grad[p * 3 + axis] * (0.5f / eps);
        }
        const int64_t row = (int64_t)e * n + p;
        float in[36], x01[3];
        DH_UNROLL for (int c = 0; c < 3; ++c) x01[c] = (xe[c] + radius) / (2.f * radius);
        load_geo_input(ws + O.gin, O.lde, row, xe, radius, in);       // encodings: rows 3..34 left by the forward
        DH_UNROLL for (int c = 0; c < 3; ++c) ws[O.gin + c * O.lde + row] = in[c];
        ws[O.gin + 35 * O.lde + row] = 1.f;             // ones column: dW0[:,35] accumulates the bias gradient
        store_col(ws + O.x01, O.lde, row, x01);
        store_col(ws + O.dout, O.lde, row, dout);
        float din[36];
        DH_UNROLL for (int k = 0; k < 36; ++k) din[k] = 0.f;
        for (int j = 0; j < HM_HID; ++j) {
            const f32x4* wr = reinterpret_cast<const f32x4*>(W + HP_G0 + j * 36);
            float a = W[HP_G0B + j];
            DH_UNROLL for (int k4 = 0; k4 < 9; ++k4) {
                const f32x4 w = wr[k4];
                a = fmaf(w[0], in[4 * k4], a); a = fmaf(w[1], in[4 * k4 + 1], a);
                a = fmaf(w[2], in[4 * k4 + 2], a); a = fmaf(w[3], in[4 * k4 + 3], a);
            }
            const float h = softplus100(a);
            float dh = 0.f;
            if (e == 0) { DH_UNROLL for (int c = 0; c < HM_GOUT; ++c) dh = fmaf(W[HP_G1 + c * 64 + j], dout[c], dh); }
            else dh = W[HP_G1 + j] * dout[0];
            const float da = dh / (1.f + __builtin_amdgcn_exp2f(-a * (SOFTPLUS_BETA * 1.44269504088896f)));
            ws[O.hh + j * O.lde + row] = h;
            ws[O.da + j * O.lde + row] = da;
            DH_UNROLL for (int k4 = 0; k4 < 9; ++k4) {
                const f32x4 w = wr[k4];
                din[4 * k4] = fmaf(w[0], da, din[4 * k4]); din[4 * k4 + 1] = fmaf(w[1], da, din[4 * k4 + 1]);
                din[4 * k4 + 2] = fmaf(w[2], da, din[4 * k4 + 2]); din[4 * k4 + 3] = fmaf(w[3], da, din[4 * k4 + 3]);
            }
        }
        // encoding adjoint, level-major [16][E n] float2: coalesced here and in hash_table_bwd_kernel
        DH_UNROLL for (int l = 0; l < HG_L; ++l)
            *reinterpret_cast<float2*>(ws + O.din + ((int64_t)l * HW_E * n + row) * 2) = make_float2(din[3 + 2 * l], din[4 + 2 * l]);
    }
}

// Table gradient.  Float atomics execute at the memory side in 64-B requests (MI355X_MICROARCH.md "Global float
// atomics": ~20 G requests/s chip-wide when every lane hits its own segment), so the kernel is built to issue as few
// 64-B requests as possible:
//   * four lanes per (point, level): lane (dx, f) owns feature f of the corners with x-offset dx, so the two features of
//     an entry and -- hash prime 1 on x / dense x-major order -- with probability 7/8 its x-neighbour entry fall in one
//     request of the same wave instruction;
//   * (a) the corner planes of a point's E = 7 evaluations (centre, +-eps per axis) that coincide with corners of the
//     centre's cell -- the whole cell when the evaluation stays inside it (coarse levels), the shared face when it
//     lands in the neighbour (fine levels) -- are blended into the centre's accumulators in registers;
//   * (b) consecutive samples of a ray that share a cell are summed across lanes (segmented scan over quads, lane order
//     = ray order; grid.y = level, so a wave is 16 consecutive samples at one level) and only the last quad of each run
//     issues the centre's atomics.
// Against the per-evaluation scatter (tcnn's scheme: E x 8 corners x 2 features atomics per point and level) this is
// 24.9 ms -> see DESIGN.md for the measured ladder.  dh_hash_set_scatter_mode(1 / 2) switches (b) / (a) off for ablation.
// Development probes (scripts/build_variant.sh; timing only, results are wrong): -DHASH_PROBE_STORE replaces every atomic add by a
// plain store to the same address (what the kernel costs without read-modify-write at the memory side), -DHASH_PROBE_SPREAD sends
// every atomic to a line of its own (no two requests share an address: the atomic path without contention), -DHASH_PROBE_LEVEL=k
// runs level k only.  Their numbers: DESIGN_NEXT_ROWS.md section 7 "what bounds the table scatter".
// MODE 3, the reproducible scatter (dh_hash_weight_grads_parts, parts bit 4): the same merges, but every add is converted to 2^-48
// fixed point and added to an int64 accumulator by an INTEGER atomic -- integer addition is associative, so the sums do not depend
// on the order in which the memory side sees the requests, and two launches on the same inputs agree bit for bit.  Resolution
// 3.6e-15: the table's gradients are small (a 2^-40 grid, tried first in round 6, read 2e-6 relative against the float form where
// this one reads 7e-8 -- the float form's own rounding).  Range (round 6, ADVICE r5: round 5 allowed single contributions up to 16,384,
// so three near-limit adds on one entry wrapped the int64 unnoticed): ONE contribution must be below 64 = 2^6 (2^54 in fixed point), a
// SUM is exact up to +-16,384 = 2^14 (2^62), i.e. 256 same-signed contributions at the limit.
//   * a non-finite or out-of-range contribution raises the flag word behind the accumulators and hash_fix_to_float_kernel then
//     writes NaN to the WHOLE table gradient;
//   * an accumulator that ends in the guard band |acc| >= 2^62 becomes NaN for ITS entry (what the float form would show as inf):
//     every true sum of magnitude 2^14 ... 3 x 2^14 lands there, so a finite wrong value needs more than 768 same-signed
//     contributions at the limit on one entry.
constexpr double HG_FIX_ONE = 281474976710656.0;               // 2^48
constexpr float HG_FIX_LIMIT = 64.f;                           // one contribution
constexpr long long HG_FIX_GUARD = 1ll << 62;                  // |accumulator| from here on: the entry is NaN
constexpr int HG_FIX_FLAG_WORDS = 8;                           // one 64-B line of int64 behind the accumulators
#if defined(HASH_PROBE_STORE)
#define HG_SCATTER_ADD(ptr, v) (*(volatile float*)(ptr) = (v))
#elif defined(HASH_PROBE_SPREAD)
#define HG_SCATTER_ADD(ptr, v) atomicAdd(d_table + ((((size_t)blockIdx.x * 256 + threadIdx.x) * 16 + (size_t)blockIdx.y * 0x9E3779B1u) % ((size_t)1 << 19)) * 16, (v))
#else
#define HG_SCATTER_ADD(ptr, v) atomicAdd((ptr), (v))
#endif
template <int MODE>
__device__ __forceinline__ void hg_scatter_add(float* T, unsigned long long* F, unsigned long long* flag, size_t off, float v) {
    if constexpr (MODE == 3) {
        if (!(fabsf(v) < HG_FIX_LIMIT)) { atomicOr(flag, 1ull); return; }            // NaN / inf / out of range
        const long long q = __double2ll_rn((double)v * HG_FIX_ONE);
        atomicAdd(F + off, (unsigned long long)q);
    } else {
        (void)F; (void)flag;
        HG_SCATTER_ADD(T + off, v);
    }
}
template <int MODE>   // 0: both merges; 1: no lane-run merge; 2: no evaluation merge (ablation / debugging); 3: both merges, fixed-point accumulators
__global__ __launch_bounds__(256) void hash_table_bwd_kernel(HashLevels H, const float* __restrict__ ws, HashWs O, int64_t n,
                                                             float* __restrict__ d_table, unsigned long long* __restrict__ fix,
                                                             const int64_t* __restrict__ n_act) {
    const int l = blockIdx.y;
#ifdef HASH_PROBE_LEVEL
    if (l != HASH_PROBE_LEVEL) return;
#endif
    const int lane = threadIdx.x & 63;
    const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int dx = (threadIdx.x >> 1) & 1, f = threadIdx.x & 1;
    const int64_t na = active_rows(n_act, n);              // rows are laid out for the capacity n, na of them are real
    if ((int64_t)blockIdx.x * 64 >= na) return;            // workgroup-uniform: no sample of this block exists
    const bool valid = p < na;
    const int64_t pc = valid ? p : na - 1;
    const float* X = ws + O.x01;
    const float* D = ws + O.din + (int64_t)l * HW_E * n * 2 + f;          // feature f of the level-major float2 rows
    float* T = d_table + f;
    unsigned long long* F = fix + f;
    unsigned long long* flag = fix + (size_t)H.total * HG_F;
    const float x0[3] = {X[pc], X[O.lde + pc], X[2 * O.lde + pc]};
    uint32_t g0[3];
    float w0[3];
    hg_cell(H, l, x0, g0, w0);
    const float wx = dx ? w0[0] : 1.f - w0[0];
    float acc[4];                                                          // centre corners (dx, dy, dz): index dy + 2 dz
    {
        const float d = valid ? D[pc * 2] : 0.f;
        DH_UNROLL for (int k = 0; k < 4; ++k)
            acc[k] = wx * ((k & 1) ? w0[1] : 1.f - w0[1]) * ((k >> 1) ? w0[2] : 1.f - w0[2]) * d;
    }
    const float s = H.scale[l];
    DH_UNROLL for (int e = 1; e < HW_E; ++e) {
        const int a = (e - 1) >> 1;
        const int64_t row = (int64_t)e * n + pc;
        const float pos = X[a * O.lde + row] * s + 0.5f;
        const float fl = floorf(pos);
        const float wa = pos - fl;
        const uint32_t ga = (uint32_t)(int)fl;
        const int delta = (int)(ga - g0[a]);
        const float d = valid ? D[row * 2] : 0.f;
        DH_UNROLL for (int ba = 0; ba < 2; ++ba) {
            const int t = delta + ba;                    // plane position relative to the centre cell: 0 / 1 = shared
            const bool to_centre = MODE != 2 && (unsigned)t < 2u;
            const float wpa = (ba ? wa : 1.f - wa) * d;
            if (a == 0) {
                // plane of constant x: the lanes whose dx equals t own the matching centre corners
                const float m = (to_centre && t == dx) ? wpa : 0.f;
                DH_UNROLL for (int k = 0; k < 4; ++k)
                    acc[k] = fmaf(m, ((k & 1) ? w0[1] : 1.f - w0[1]) * ((k >> 1) ? w0[2] : 1.f - w0[2]), acc[k]);
                if (!to_centre && valid) {               // outer plane: the quad's dx bit picks the y half instead
                    DH_UNROLL for (int bz = 0; bz < 2; ++bz) {
                        const float wt = wpa * (dx ? w0[1] : 1.f - w0[1]) * (bz ? w0[2] : 1.f - w0[2]);
                        const uint32_t idx = hg_index(H, l, ga + ba, g0[1] + dx, g0[2] + bz);
                        hg_scatter_add<MODE>(T, F, flag, (size_t)idx * HG_F, wt);
                    }
                }
            } else {
                // plane of constant y (a = 1) or z (a = 2); o = the other in-plane axis besides x
                const int o = a == 1 ? 2 : 1;
                DH_UNROLL for (int bo = 0; bo < 2; ++bo) {
                    const float wt = wx * wpa * (bo ? w0[o] : 1.f - w0[o]);
                    const int k0 = a == 1 ? 2 * bo : bo, k1 = a == 1 ? 1 + 2 * bo : bo + 2;   // centre corner with bit_a = 0 / 1
                    acc[k0] += (to_centre && t == 0) ? wt : 0.f;
                    acc[k1] += (to_centre && t == 1) ? wt : 0.f;
                }
                if (!to_centre && valid) {
                    DH_UNROLL for (int bo = 0; bo < 2; ++bo) {
                        const float wt = wx * wpa * (bo ? w0[o] : 1.f - w0[o]);
                        uint32_t cc[3];
                        cc[0] = g0[0] + dx; cc[a] = ga + ba; cc[o] = g0[o] + bo;
                        const uint32_t idx = hg_index(H, l, cc[0], cc[1], cc[2]);
                        hg_scatter_add<MODE>(T, F, flag, (size_t)idx * HG_F, wt);
                    }
                }
            }
        }
    }
    // (b) segmented sum over runs of quads sharing the centre cell.  Keys are the full uint32 cell coordinates
    // (out-of-box samples wrap to large values: no packing); every shuffle executes on all lanes.
    const uint32_t k0 = valid ? g0[0] : 0x80000000u + (uint32_t)lane, k1 = g0[1], k2 = g0[2];
    const uint32_t p0 = __shfl_up(k0, 4), p1 = __shfl_up(k1, 4), p2 = __shfl_up(k2, 4);
    const uint32_t n0 = __shfl_down(k0, 4), n1 = __shfl_down(k1, 4), n2 = __shfl_down(k2, 4);
    const bool eq_prev = ((p0 ^ k0) | (p1 ^ k1) | (p2 ^ k2)) == 0u;
    const bool eq_next = ((n0 ^ k0) | (n1 ^ k1) | (n2 ^ k2)) == 0u;
    const bool head = MODE == 1 || lane < 4 || !eq_prev;
    const bool tail = MODE == 1 || lane >= 60 || !eq_next;
    int start = head ? lane : 0;                          // first lane (same sub-lane) of this quad's run
    DH_UNROLL for (int off = 4; off < 64; off <<= 1) {
        const int tt = __shfl_up(start, off);
        if (lane >= off) start = max(start, (tt & ~3) | (lane & 3));
    }
    DH_UNROLL for (int off = 4; off < 64; off <<= 1) {
        const bool take = lane - off >= start;
        DH_UNROLL for (int q = 0; q < 4; ++q) {
            const float tq = __shfl_up(acc[q], off);
            acc[q] += take ? tq : 0.f;
        }
    }
    if (valid && tail) {
        DH_UNROLL for (int k = 0; k < 4; ++k) {
            const uint32_t idx = hg_index(H, l, g0[0] + dx, g0[1] + (k & 1), g0[2] + (k >> 1));
            hg_scatter_add<MODE>(T, F, flag, (size_t)idx * HG_F, acc[k]);
        }
    }
}

// fixed point -> float of the reproducible scatter's accumulators (count = entries x 2); NaN everywhere when the flag word is set,
// NaN for an entry whose accumulator ended in the guard band (see HG_FIX_GUARD above)
__global__ __launch_bounds__(256) void hash_fix_to_float_kernel(const long long* __restrict__ fix, int64_t count, float* __restrict__ d_table) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const long long a = fix[i];
    const bool bad = fix[count] != 0 || a >= HG_FIX_GUARD || a <= -HG_FIX_GUARD;
    d_table[i] = bad ? __builtin_nanf("") : (float)((double)a * (1.0 / HG_FIX_ONE));
}

// dW[m][k] = sum_r A[m][r] B[k][r]  (feature-major operands, m < M <= 64, k < K <= 64), plus colsum[m] = sum_r A[m][r].
// One 4-wave workgroup per slab of rows: 64-row tiles of both operands are loaded as coalesced 16-B row quads (16
// lanes cover one feature's 256 B), staged in LDS (row stride 65: the operand reads of a 32-lane half hit 32 banks), and
// each wave accumulates one 32x32 quadrant of dW with fp32 MFMA 32x32x2; the next tile's global loads are in flight
// while the current one is multiplied.
// rows = blocks x blk: `blocks` evaluations of `blk` (= capacity) rows each, of which the first n_act (device count) are real
struct SmallDwJob { int64_t a, b; int M, K; int64_t rows, ld, blk; };
struct SmallDwJobs { SmallDwJob j[HW_JOBS]; };
constexpr int DWT = 64, DWS = DWT + 1;

__global__ __launch_bounds__(256) void small_dw_kernel(const float* __restrict__ ws, SmallDwJobs J, float* __restrict__ slabs,
                                                       const int64_t* __restrict__ n_act) {
    __shared__ float At[64 * DWS], Bt[64 * DWS];
    const SmallDwJob job = J.j[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, kk = lane >> 5;
    const int mt = wave >> 1, kt = wave & 1;
    const int64_t per = ((job.rows + HW_SLABS - 1) / HW_SLABS + DWT - 1) / DWT * DWT;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < job.rows ? r0 + per : job.rows;
    const int64_t na = active_rows(n_act, job.blk);      // packed rays: rows [e blk + na, (e + 1) blk) of every evaluation e are stale
    const bool ragged = na < job.blk;
    const int lf = tid >> 4, lc = tid & 15;              // loader: features lf + 16 q, rows 4 lc .. 4 lc + 3 of the tile
    const float* A = ws + job.a;
    const float* B = ws + job.b;
    f32x16 acc;
    DH_UNROLL for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4], rb[4];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // position of a tile's first row inside its evaluation block, carried along instead of recomputed (the recomputing form paid
    // three 64-bit divisions per 64-row tile and thread: 0.63 against 0.40 ms for FEWER rows on packed rays): off_ld belongs to the
    // tile the next load() fetches, off_cur to the tile the loop is at; both advance by one tile per trip
    int64_t off_ld = ragged ? r0 % job.blk : 0, off_cur = off_ld;
    auto advance = [&](int64_t& off) {
        if (!ragged) return;
        off += DWT;
        while (off >= job.blk) off -= job.blk;
    };
    auto tile_dead = [&](int64_t off) {                  // wholly inside one block's stale rows (workgroup-uniform)
        return ragged && off + DWT - 1 < job.blk && off >= na;
    };
    auto load = [&](int64_t rt) {                        // called once per tile, in order
        const int64_t r = rt + 4 * lc;
        const int64_t off = off_ld;
        advance(off_ld);
        if (tile_dead(off)) {                            // nothing of this tile exists: no memory traffic either
            DH_UNROLL for (int q = 0; q < 4; ++q) { ra[q] = z4; rb[q] = z4; }
            return;
        }
        int64_t rin = off + 4 * lc;                      // blk is a multiple of 4 (API contract), so the 4 rows share a block
        while (ragged && rin >= job.blk) rin -= job.blk;
        // how many of this thread's 4 rows are real: one number per load, the same for every feature (the per-element 64-bit compares
        // this replaces cost 0.22 of the kernel's 0.58 ms on packed rays); 4 everywhere but at the seam of a block
        const int nreal = !ragged ? 4 : (na - rin >= 4 ? 4 : (na - rin <= 0 ? 0 : (int)(na - rin)));
        DH_UNROLL for (int q = 0; q < 4; ++q) {
            const int f = lf + 16 * q;
            ra[q] = (f < job.M && r < r1) ? *reinterpret_cast<const f32x4*>(A + (int64_t)f * job.ld + r) : z4;
            rb[q] = (f < job.K && r < r1) ? *reinterpret_cast<const f32x4*>(B + (int64_t)f * job.ld + r) : z4;
            if (r + 4 > r1) {                            // ragged tail: rows past the end hold garbage
                DH_UNROLL for (int u = 0; u < 4; ++u) if (r + u >= r1) { ra[q][u] = 0.f; rb[q][u] = 0.f; }
            }
            if (nreal < 4) {                             // rows past the device-side count: not written this iteration
                DH_UNROLL for (int u = 0; u < 4; ++u) if (u >= nreal) { ra[q][u] = 0.f; rb[q][u] = 0.f; }
            }
        }
    };
    if (r0 < r1) load(r0);
    const bool active = mt * 32 < job.M && kt * 32 < job.K;          // wave-uniform
    for (int64_t rt = r0; rt < r1; rt += DWT) {
        const bool dead = tile_dead(off_cur);
        advance(off_cur);
        if (dead) {                                      // its (zero) registers were loaded by the previous trip: replace them
            if (rt + DWT < r1) load(rt + DWT);
            continue;
        }
        __syncthreads();                                 // the previous tile's readers are done
        DH_UNROLL for (int q = 0; q < 4; ++q) {
            const int f = lf + 16 * q;
            DH_UNROLL for (int u = 0; u < 4; ++u) {
                At[f * DWS + 4 * lc + u] = ra[q][u];
                Bt[f * DWS + 4 * lc + u] = rb[q][u];
            }
            cs[q] += (ra[q][0] + ra[q][1]) + (ra[q][2] + ra[q][3]);
        }
        __syncthreads();
        if (rt + DWT < r1) load(rt + DWT);
        if (active) {
            const float* ap = At + (mt * 32 + i) * DWS + kk;
            const float* bp = Bt + (kt * 32 + i) * DWS + kk;
            DH_UNROLL for (int s2 = 0; s2 < DWT / 2; ++s2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s2], bp[2 * s2], acc, 0, 0, 0);
        }
    }
    float* out = slabs + ((int64_t)blockIdx.y * HW_SLABS + blockIdx.x) * HW_DW_FLOATS;
    DH_UNROLL for (int q = 0; q < 16; ++q) {
        const int m = mt * 32 + 8 * (q >> 2) + 4 * kk + (q & 3);     // accumulator layout of 32x32x2: row = 8*(q/4)+4*(lane/32)+q%4
        out[m * 64 + kt * 32 + i] = acc[q];
    }
    DH_UNROLL for (int q = 0; q < 4; ++q) {              // column sums: reduce over the 16 lanes sharing a feature
        float t = cs[q];
        DH_UNROLL for (int off = 1; off < 16; off <<= 1) t += __shfl_xor(t, off);
        if (lc == 0) out[64 * 64 + lf + 16 * q] = t;
    }
}

// sum the slabs of every job (fixed order: deterministic) -> dwsum [HW_JOBS][HW_DW_FLOATS]
__global__ __launch_bounds__(256) void small_dw_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ dwsum) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= HW_DW_FLOATS) return;
    const float* s = slabs + (int64_t)blockIdx.y * HW_SLABS * HW_DW_FLOATS + e;
    float t = 0.f;
    for (int b = 0; b < HW_SLABS; ++b) t += s[(int64_t)b * HW_DW_FLOATS];
    dwsum[blockIdx.y * HW_DW_FLOATS + e] = t;
}

// weight-norm fold of the five small linears: W = g v/||v||:  g_bar = (dW . v)/||v||,  v_bar = g/||v|| (dW - (dW . v) v/||v||^2)
__global__ __launch_bounds__(64) void hash_fold_kernel(const float* __restrict__ params, const float* __restrict__ hp,
                                                       const float* __restrict__ dwsum, HashParamOff P,
                                                       float* __restrict__ grad) {
    const HashLinDesc D[5] = {{P.g0_b, P.g0_g, P.g0_v, 64, HM_GIN, 36, HP_G0, HP_G0B, HP_ROW_G0},
                              {P.g1_b, P.g1_g, P.g1_v, HM_GOUT, 64, 64, HP_G1, HP_G1B, HP_ROW_G1},
                              {P.c0_b, P.c0_g, P.c0_v, 64, HM_CIN, 32, HP_C0, HP_C0B, HP_ROW_C0},
                              {P.c1_b, P.c1_g, P.c1_v, 64, 64, 64, HP_C1, HP_C1B, HP_ROW_C1},
                              {P.c2_b, P.c2_g, P.c2_v, 3, 64, 64, HP_C2, HP_C2B, HP_ROW_C2}};
    const int job = blockIdx.y;
    const HashLinDesc L = D[job];
    const int row = blockIdx.x, lane = threadIdx.x;
    if (row >= L.out) return;
    const float* dW = dwsum + job * HW_DW_FLOATS + row * 64;
    const float* v = params + L.v + (int64_t)row * L.in;
    float dot = 0.f;
    for (int k = lane; k < L.in; k += 64) dot += dW[k] * v[k];
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    const float inv = hp[HP_INV + L.rowbase + row], rs = hp[HP_RS + L.rowbase + row];
    for (int k = lane; k < L.in; k += 64) grad[L.v + (int64_t)row * L.in + k] = rs * (dW[k] - dot * v[k] * inv * inv);
    if (lane == 0) {
        grad[L.g + row] = dot * inv;
        // bias: geometry lin0 keeps it in the ones column (k = 35); everyone else in the column sums of A
        grad[L.b + row] = job == 0 ? dW[35] : dwsum[job * HW_DW_FLOATS + 64 * 64 + row];
    }
}

// ---------------------------------------------------------------- launchers
static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }

int64_t hash_num_params() { return make_hash_param_off(hashgrid_entries()).total; }
HashParamOff hash_param_off() { return make_hash_param_off(hashgrid_entries()); }

int launch_hash_pack(const float* params, float* hp, hipStream_t st) {
    hipLaunchKernelGGL(hash_pack_kernel, dim3(64, 5), dim3(64), 0, st, params, hp, hash_param_off());
    return ok();
}
int launch_hash_sdf_nograd(const float* params, const float* hp, const float* pts, int64_t n, float radius, float* sdf,
                           hipStream_t st) {
    hipLaunchKernelGGL(hash_sdf_nograd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hashgrid_levels(),
                       params + hash_param_off().table, hp, pts, n, radius, sdf);
    return ok();
}
int launch_hash_geo_fwd(const float* params, const float* hp, const float* pts, int64_t n, float radius, float eps,
                        float* ws, int save, float* sdf, float* feat, float* grad, const int64_t* n_act, hipStream_t st) {
    const HashWs O = make_hash_ws(n);
    float* gin = save ? ws + O.gin : ws;               // forward-only callers hand over just the GIN rows
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(hash_encode7_kernel, dim3(nb, HG_L), dim3(256), 0, st, hashgrid_levels(), params + hash_param_off().table, pts,
                       n, radius, eps, gin, O.lde, 0, n_act);
    hipLaunchKernelGGL(hash_geo_mlp_fwd_kernel, dim3(nb), dim3(256), 0, st, hp, pts, n, radius, eps, gin, O.lde, sdf, feat, grad,
                       n_act);
    return ok();
}
int launch_sh_color_fwd(const float* hp, const float* feat, const float* normals, const float* dirs, int n_per_ray,
                        int64_t n, float* color, const int64_t* n_act, hipStream_t st) {
    hipLaunchKernelGGL(sh_color_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hp, feat, normals, dirs,
                       n_per_ray, n, color, n_act);
    return ok();
}

}  // namespace dh

namespace dh {

int64_t hash_workspace_floats(int64_t n) { return make_hash_ws(n).fix + ((int64_t)hashgrid_entries() * 2 + HG_FIX_FLAG_WORDS) * 2; }
int64_t hash_infer_workspace_floats(int64_t n) { return make_hash_ws(n).lde * 36; }

int launch_sh_color_bwd(const float* hp, const float* feat, const float* normals, const float* dirs, const float* d_color,
                        int n_per_ray, int64_t n, float* ws, float* d_feat, float* d_normals, const int64_t* n_act,
                        hipStream_t st) {
    hipLaunchKernelGGL(sh_color_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hp, feat, normals, dirs,
                       d_color, n_per_ray, n, ws, make_hash_ws(n), d_feat, d_normals, n_act);
    return ok();
}

int launch_hash_geo_bwd(const float* params, const float* hp, const float* pts, const float* d_sdf, const float* d_feat,
                        const float* d_grad, int64_t n, float radius, float eps, float* ws, const int64_t* n_act,
                        hipStream_t st) {
    (void)params;
    hipLaunchKernelGGL(hash_geo_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hp, pts, d_sdf, d_feat,
                       d_grad, n, radius, eps, ws, make_hash_ws(n), n_act);
    return ok();
}

// all parameter gradients of the hash family from the rows the two kernels above left in ws: grad [hash_num_params].
// parts: 1 = the table gradient (the first hashgrid_entries() x 2 floats of grad), 2 = the five small linears, 3 = both; + 4 = the table
// scatter in its reproducible fixed-point form (hg_scatter_add MODE 3) -- table FIRST, so that
// a data-parallel caller can start the 49 MB table all-reduce on a side stream while the small weight-gradient GEMMs still run
// (dynhor_amd/hash_fields.py).
int launch_hash_weight_grads(const float* params, const float* hp, int64_t n, float* ws, float* grad, const int64_t* n_act,
                             int parts, hipStream_t st) {
    const HashWs O = make_hash_ws(n);
    const HashParamOff P = hash_param_off();
    const int64_t en = (int64_t)HW_E * n;
    if (parts & 1) {
        // table: scatter the encoding adjoint of all E n evaluations
        const int64_t count = (int64_t)hashgrid_entries() * 2;
        const dim3 grid((unsigned)((4 * n + 255) / 256), HG_L);
        unsigned long long* fix = reinterpret_cast<unsigned long long*>(ws + O.fix);
        if (parts & 4) {
            // reproducible form: int64 fixed-point accumulators in the workspace, converted once
            if (hipMemsetAsync(fix, 0, (size_t)(count + HG_FIX_FLAG_WORDS) * 8, st) != hipSuccess) return -3;
            hipLaunchKernelGGL(hash_table_bwd_kernel<3>, grid, dim3(256), 0, st, hashgrid_levels(), ws, O, n, grad + P.table, fix, n_act);
            hipLaunchKernelGGL(hash_fix_to_float_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const long long*>(fix), count, grad + P.table);
        } else {
            if (hipMemsetAsync(grad + P.table, 0, (size_t)count * sizeof(float), st) != hipSuccess) return -3;
            const int mode = hash_scatter_mode();      // dh_hash_set_scatter_mode: 0 shipping, 1 / 2 ablations (test / diagnosis only)
            if (mode == 1) hipLaunchKernelGGL(hash_table_bwd_kernel<1>, grid, dim3(256), 0, st, hashgrid_levels(), ws, O, n, grad + P.table, fix, n_act);
            else if (mode == 2) hipLaunchKernelGGL(hash_table_bwd_kernel<2>, grid, dim3(256), 0, st, hashgrid_levels(), ws, O, n, grad + P.table, fix, n_act);
            else hipLaunchKernelGGL(hash_table_bwd_kernel<0>, grid, dim3(256), 0, st, hashgrid_levels(), ws, O, n, grad + P.table, fix, n_act);
        }
    }
    if (parts & 2) {
        SmallDwJobs J;
        J.j[0] = {O.da, O.gin, 64, 36, en, O.lde, n};         // geometry lin0 (ones column -> bias)
        J.j[1] = {O.dout, O.hh, HM_GOUT, 64, en, O.lde, n};   // geometry lin1
        J.j[2] = {O.dz1, O.cin, 64, 32, n, O.ldn, n};         // colour lin0
        J.j[3] = {O.dz2, O.h1, 64, 64, n, O.ldn, n};          // colour lin1
        J.j[4] = {O.d_o, O.h2, 3, 64, n, O.ldn, n};           // colour lin2
        float* slabs = ws + O.slabs;
        float* dwsum = ws + O.total;
        hipLaunchKernelGGL(small_dw_kernel, dim3(HW_SLABS, HW_JOBS), dim3(256), 0, st, ws, J, slabs, n_act);
        hipLaunchKernelGGL(small_dw_reduce_kernel, dim3((HW_DW_FLOATS + 255) / 256, HW_JOBS), dim3(256), 0, st, slabs, dwsum);
        hipLaunchKernelGGL(hash_fold_kernel, dim3(64, HW_JOBS), dim3(64), 0, st, params, hp, dwsum, P, grad);
    }
    return ok();
}

}  // namespace dh
