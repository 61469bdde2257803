// Multiresolution hash-grid encoding (SURVEY.md §8f n3 / BASELINE.json configs[3]): the gather-bound building block of
// the instant-nsr-pl variant the reference README names as its direction (README.md:11,13; the code itself is on an
// unmounted branch -- the algorithm restated here is Mueller et al. 2022 §3, see oracle/hashgrid_oracle.py).
//   L = 16 levels x F = 2 features, table size T = 2^19 per level, resolutions 16 -> 2049 (geometric), dense indexing while
//   a level's grid fits the table, spatial hash (x*1) ^ (y*2654435761) ^ (z*805459861) mod T above; trilinear blend.
// Forward: one thread per (point, level) -- the 16 lanes of a point write its 32 features as one 128-B segment; every
// gather is an 8-byte float2 (both features of a corner).  Backward: float atomics into the table gradient
// (order-dependent in the last bits by nature; the NeuS path above has none).  HBM/L2-bound by construction: 1 KiB of
// gathers + 128 B written per point forward.
#include <cmath>
#include "tile.h"
#include "kernels.h"
#include "hashgrid_dev.h"

namespace dh {

__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(HashLevels H, const float* __restrict__ table,
                                                           const float* __restrict__ x01, int64_t n, float* __restrict__ out) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = gid >> 4;
    const int l = (int)(gid & 15);
    if (p >= n) return;
    const float s = H.scale[l];
    float pos[3], w[3];
    uint32_t g[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) {
        pos[c] = x01[p * 3 + c] * s + 0.5f;
        const float f = floorf(pos[c]);
        w[c] = pos[c] - f;
        g[c] = (uint32_t)(int)f;
    }
    float a0 = 0.f, a1 = 0.f;
    DH_UNROLL for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const uint32_t idx = hg_index(H, l, g[0] + dx, g[1] + dy, g[2] + dz);
        const float wt = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        const float2 f = *reinterpret_cast<const float2*>(table + (size_t)idx * HG_F);
        a0 = fmaf(wt, f.x, a0);
        a1 = fmaf(wt, f.y, a1);
    }
    *reinterpret_cast<float2*>(out + p * (HG_L * HG_F) + l * HG_F) = make_float2(a0, a1);
}

__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(HashLevels H, const float* __restrict__ x01,
                                                           const float* __restrict__ d_out, int64_t n,
                                                           float* __restrict__ d_table) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t p = gid >> 4;
    const int l = (int)(gid & 15);
    if (p >= n) return;
    const float s = H.scale[l];
    float w[3];
    uint32_t g[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) {
        const float pos = x01[p * 3 + c] * s + 0.5f;
        const float f = floorf(pos);
        w[c] = pos - f;
        g[c] = (uint32_t)(int)f;
    }
    const float2 d = *reinterpret_cast<const float2*>(d_out + p * (HG_L * HG_F) + l * HG_F);
    DH_UNROLL for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const uint32_t idx = hg_index(H, l, g[0] + dx, g[1] + dy, g[2] + dz);
        const float wt = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        atomicAdd(d_table + (size_t)idx * HG_F + 0, wt * d.x);
        atomicAdd(d_table + (size_t)idx * HG_F + 1, wt * d.y);
    }
}

int64_t hashgrid_entries() { return levels().total; }

int hashgrid_level(int l, float* scale, uint32_t* res, uint32_t* offset, uint32_t* dense) {
    if (l < 0 || l >= HG_L) return -1;
    const HashLevels& H = levels();
    *scale = H.scale[l]; *res = H.res[l]; *offset = H.offset[l]; *dense = H.dense[l];
    return 0;
}

int launch_hashgrid_fwd(const float* table, const float* x01, int64_t n, float* out, hipStream_t st) {
    const int64_t threads = n * 16;
    hipLaunchKernelGGL(hashgrid_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, levels(), table, x01, n, out);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int launch_hashgrid_bwd(const float* x01, const float* d_out, int64_t n, float* d_table, hipStream_t st) {
    const int64_t threads = n * 16;
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, levels(), x01, d_out, n, d_table);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
