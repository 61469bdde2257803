// Weight-gradient GEMMs, cross-tile reduction and weight-norm fold.
//
// Wbar[256 x NB*32] = sum over points of A[p,:]^T B[p,:]  with the POINT axis as the MFMA k dimension.  Both operands
// are saved "native" tiles (tile.h): a float4 at ((..m..t..)*4 + r4)*64 + lane IS four consecutive A (or B)
// fragments of v_mfma_f32_32x32x2_f32 for the k-pairs {row, row+4}, so operands stream HBM -> VGPR with 1 KiB
// coalesced loads and no LDS.  8 waves per workgroup, wave w owns output rows [32w, 32w+32) x all NB n-tiles
// (128 accumulator VGPRs at NB=8); split-K over tiles across gridDim.x workgroups, slabs reduced in fold_kernel
// (deterministic: no float atomics).
#include "tile.h"
#include "kernels.h"
#include "workspace.h"

namespace dh {

struct DwJob {
    const float* A1; const float* B1;
    const float* A2; const float* B2;      // optional second (A,B) pair accumulated into the same output
    float* out;                            // [G][8][nb][16][64]
    int nb;                                // 8: B is a main native tile; 2: B is an aux native tile
};
struct DwJobs { DwJob j[16]; int n; };

template <int NB>
__device__ __forceinline__ void dw_body(const DwJob& J, int64_t t0, int64_t t1, int g, int wave, int lane) {
    f32x16 acc[NB];
    DH_UNROLL for (int j = 0; j < NB; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int abase = (((wave >> 1) * MT) * 2 + (wave & 1)) * 4 * 64 + lane;
    for (int pair = 0; pair < 2; ++pair) {
        const float* A = pair ? J.A2 : J.A1;
        const float* Bm = pair ? J.B2 : J.B1;
        if (!A) continue;
        for (int64_t tile = t0; tile < t1; ++tile) {
            const f32x4* ap = reinterpret_cast<const f32x4*>(A + tile * TILE_F) + abase;
            const f32x4* bp = reinterpret_cast<const f32x4*>(Bm + tile * (NB == 8 ? TILE_F : AUXT_F)) + lane;
            _Pragma("unroll 2") for (int kq = 0; kq < MT * 4; ++kq) {
                const int m = kq >> 2, r4 = kq & 3;
                const f32x4 a = ap[(m * 8 + r4) * 64];
                f32x4 b[NB];
                DH_UNROLL for (int j = 0; j < NB; ++j) {
                    const int bi = (NB == 8) ? ((j >> 1) * MT * 8 + m * 8 + (j & 1) * 4 + r4) : ((m * 2 + j) * 4 + r4);
                    b[j] = bp[bi * 64];
                }
                DH_UNROLL for (int rr = 0; rr < 4; ++rr)
                    DH_UNROLL for (int j = 0; j < NB; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rr], b[j][rr], acc[j], 0, 0, 0);
            }
        }
    }
    float* o = J.out + ((int64_t)(g * 8 + wave) * NB) * 1024 + lane;
    DH_UNROLL for (int j = 0; j < NB; ++j)
        DH_UNROLL for (int r = 0; r < 16; ++r) o[j * 1024 + r * 64] = acc[j][r];
}

__global__ __launch_bounds__(512, 2) void dw_kernel(DwJobs jobs, int64_t ntiles) {
    const DwJob J = jobs.j[blockIdx.y];
    const int G = gridDim.x, g = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t t0 = ntiles * g / G, t1 = ntiles * (g + 1) / G;
    if (J.nb == 8) dw_body<8>(J, t0, t1, g, wave, lane);
    else dw_body<2>(J, t0, t1, g, wave, lane);
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

__device__ __forceinline__ float slab_elem(const float* __restrict__ slab, int nb, int G, int o, int i) {
    const int w = o >> 5, ro = o & 31, j = i >> 5;
    const int r = (ro & 3) + 4 * (ro >> 3);
    const int lane = (i & 31) + 32 * ((ro >> 2) & 1);
    const int64_t stride = (int64_t)8 * nb * 1024;
    const float* p = slab + ((int64_t)w * nb + j) * 1024 + r * 64 + lane;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += p[g * stride];
    return s;
}

__global__ __launch_bounds__(256) void fold_kernel(FoldTable T, SlabPtrs S, int G, const float* __restrict__ tred, int nS,
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
                if (i >= Ln.a_c0 && i < Ln.a_c1) d = Ln.a_scale * slab_elem(S.out[Ln.jobA], S.nb[Ln.jobA], G, so, i - Ln.a_c0);
                else if (Ln.jobB >= 0 && i >= Ln.b_c0 && i < Ln.b_c1)
                    d = Ln.b_scale * slab_elem(S.out[Ln.jobB], S.nb[Ln.jobB], G, so, i - Ln.b_c0);
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
    if (tid == 0) {
        grad[Ln.goff + o] = dot * inv;
        float b;
        if (Ln.special == 2) b = tsum(19, o);
        else if (Ln.special == 1 && o == 0) b = tsum(11, 0);
        else b = tsum(Ln.bias_slot, o - Ln.row_shift);
        grad[Ln.boff + o] = b;
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

// dW slab workspace: 15 jobs x G x (256 x nb*32) floats
int64_t dw_slab_floats(int G) {
    int64_t n = 0;
    const int nbs[15] = {2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8};
    for (int j = 0; j < 15; ++j) n += (int64_t)G * 8 * nbs[j] * 1024;
    return n;
}

int launch_weight_grads(const Workspace& w, float* slabs, float* tred, int G, int nS, const float* params,
                        const float* packed, float* grad, hipStream_t st) {
    const int64_t nt = w.ntiles;
    DwJobs J{};
    SlabPtrs S{};
    const int nbs[15] = {2, 8, 8, 8, 8, 8, 8, 8, 2, 8, 8, 2, 8, 8, 8};
    float* so = slabs;
    for (int j = 0; j < 15; ++j) { J.j[j].nb = nbs[j]; J.j[j].out = so; S.out[j] = so; S.nb[j] = nbs[j]; so += (int64_t)G * 8 * nbs[j] * 1024; }
    J.n = 15;
    auto T_ = [&](float* base, int idx) { return base + (int64_t)idx * nt * TILE_F; };
    J.j[0].A1 = T_(w.zbar, 0); J.j[0].B1 = w.eaux; J.j[0].A2 = T_(w.asave, 0); J.j[0].B2 = w.t0aux;
    for (int l = 1; l <= 7; ++l) {
        J.j[l].A1 = T_(w.zbar, l); J.j[l].B1 = T_(w.act, l - 1);
        J.j[l].A2 = T_(w.asave, l); J.j[l].B2 = T_(w.tsave, l - 1);
    }
    J.j[8].A1 = T_(w.zbar, 4); J.j[8].B1 = w.eaux; J.j[8].A2 = T_(w.asave, 4); J.j[8].B2 = w.t0aux;
    J.j[9].A1 = w.featbar; J.j[9].B1 = T_(w.act, 7); J.j[9].A2 = nullptr; J.j[9].B2 = nullptr;
    J.j[10].A1 = T_(w.czbar, 0); J.j[10].B1 = w.feat; J.j[10].A2 = nullptr; J.j[10].B2 = nullptr;
    J.j[11].A1 = T_(w.czbar, 0); J.j[11].B1 = w.caux; J.j[11].A2 = nullptr; J.j[11].B2 = nullptr;
    for (int l = 1; l <= 3; ++l) {
        J.j[11 + l].A1 = T_(w.czbar, l); J.j[11 + l].B1 = T_(w.cact, l - 1); J.j[11 + l].A2 = nullptr; J.j[11 + l].B2 = nullptr;
    }
    hipLaunchKernelGGL(dw_kernel, dim3(G, 15), dim3(512), 0, st, J, nt);
    hipLaunchKernelGGL(tpart_reduce_kernel, dim3(N_TILE_PART, nS), dim3(256), 0, st, w.tpart, nt, tred);
    static const FoldTable T = build_fold_table();
    hipLaunchKernelGGL(fold_kernel, dim3(T.total_rows), dim3(256), 0, st, T, S, G, tred, nS, params, packed, grad);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
