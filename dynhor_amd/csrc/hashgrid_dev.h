// Device-side helpers of the multiresolution hash grid (shared by hashgrid.hip and hash_mlp.hip).
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dh {

constexpr int HG_L = 16, HG_F = 2;
constexpr uint32_t HG_T = 1u << 19;
constexpr int HG_BASE = 16, HG_MAX = 2048;

struct HashLevels {
    float scale[HG_L];
    uint32_t res[HG_L];
    uint32_t offset[HG_L];      // in entries (one entry = F floats)
    uint32_t dense[HG_L];
    uint32_t size[HG_L];        // entries of the level (dense: res^3 rounded up to 8; hashed: T)
    uint32_t total;
};

static inline HashLevels make_levels() {
    HashLevels h{};
    const double pls = std::exp((std::log((double)HG_MAX) - std::log((double)HG_BASE)) / (HG_L - 1));
    uint32_t off = 0;
    for (int l = 0; l < HG_L; ++l) {
        const double scale = HG_BASE * std::pow(pls, l) - 1.0;
        const uint32_t res = (uint32_t)std::ceil(scale) + 1;
        const uint64_t n = (uint64_t)res * res * res;
        const bool dense = n <= HG_T;
        const uint32_t size = dense ? (uint32_t)((n + 7) / 8 * 8) : HG_T;
        h.scale[l] = (float)scale; h.res[l] = res; h.offset[l] = off; h.dense[l] = dense ? 1u : 0u; h.size[l] = size;
        off += size;
    }
    h.total = off;
    return h;
}
static inline const HashLevels& levels() { static const HashLevels h = make_levels(); return h; }

__device__ __forceinline__ uint32_t hg_index(const HashLevels& H, int l, uint32_t x, uint32_t y, uint32_t z) {
    const uint32_t res = H.res[l];
    // dense corner coordinates reach `res` at the far faces: wrap inside the level (tcnn does the same)
    const uint32_t idx = H.dense[l] ? ((x + y * res + z * res * res) % H.size[l])
                                    : (((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (HG_T - 1));
    return idx + H.offset[l];
}


inline HashLevels hashgrid_levels() { return levels(); }

// trilinear blend of the 8 corner rows of level l at x01 in [0,1]^3
__device__ __forceinline__ void hg_encode_level(const HashLevels& H, const float* __restrict__ table, int l,
                                                const float (&x01)[3], float& f0, float& f1) {
    const float s = H.scale[l];
    float w[3];
    uint32_t g[3];
    _Pragma("unroll") for (int c = 0; c < 3; ++c) {
        const float pos = x01[c] * s + 0.5f;
        const float f = floorf(pos);
        w[c] = pos - f;
        g[c] = (uint32_t)(int)f;
    }
    float a0 = 0.f, a1 = 0.f;
    _Pragma("unroll") for (int corner = 0; corner < 8; ++corner) {
        const int dx = corner & 1, dy = (corner >> 1) & 1, dz = corner >> 2;
        const uint32_t idx = hg_index(H, l, g[0] + dx, g[1] + dy, g[2] + dz);
        const float wt = (dx ? w[0] : 1.f - w[0]) * (dy ? w[1] : 1.f - w[1]) * (dz ? w[2] : 1.f - w[2]);
        const float2 f = *reinterpret_cast<const float2*>(table + (size_t)idx * HG_F);
        a0 = fmaf(wt, f.x, a0);
        a1 = fmaf(wt, f.y, a1);
    }
    f0 = a0; f1 = a1;
}

}  // namespace dh
