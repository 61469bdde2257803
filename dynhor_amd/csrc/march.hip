// Occupancy-grid ray marching with packed variable-length rays (BASELINE.json configs[3]: the sampler of
// the instant-nsr-pl variant the reference names as its direction, README.md:11,13).  Specification: oracle/occgrid_oracle.py
// (parity unpinned: instant-nsr-pl / nerfacc are absent from /root/reference).
//
// One WAVE per ray.  Step k of a ray is the interval [t_k, t_k + step], t_k = near + (k + u) step (one stratified offset u per
// ray); it becomes a sample iff t_k + step <= far and the grid cell holding its mid-point is occupied.  64 steps are tested per
// trip (one per lane, one byte gather from the L2-resident res^3 occupancy image), the wave ballots the result and the lanes'
// prefix pop-counts give every kept step its slot, so samples come out front to back with no atomics and no sort -- packed
// order is a pure function of the inputs (bitwise reproducible).  Two launches: count -> (exclusive scan on the caller's side)
// -> emit, both running the identical test.
// Positions and cell indices use separately rounded fp32 operations in the order the oracle's tensor expressions evaluate, so
// the two select exactly the same cells and produce bit-identical interval starts.  HIP's __fmul_rn / __fadd_rn are plain
// operators that hipcc contracts into fma (also under `#pragma clang fp contract(off)` once inlined): the multiplies go
// through a one-instruction asm, which nothing can fuse.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.h"

namespace dh {

__device__ __forceinline__ float mul_rn(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

struct MarchRay {
    float o[3], d[3], near, far, u;
};

// returns whether step k of the ray is a sample; x = mid-point position, t0 = interval start
__device__ __forceinline__ bool march_test(const MarchRay& r, int k, float step, float half_step, float inv2r, int res,
                                           const uint8_t* __restrict__ occ, float& t0, float (&x)[3]) {
    t0 = r.near + mul_rn((float)k + r.u, step);
    const float tm = t0 + half_step;
    bool inside = true;
    int idx[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        x[c] = r.o[c] + mul_rn(r.d[c], tm);
        const float g = mul_rn(mul_rn(x[c], inv2r) + 0.5f, (float)res);
        const float f = floorf(g);
        inside = inside && f >= 0.f && f < (float)res;
        idx[c] = (int)f;
    }
    if (!(t0 + step <= r.far) || !inside) return false;
    return occ[((int64_t)idx[0] * res + idx[1]) * res + idx[2]] != 0;
}

template <bool EMIT>
__global__ __launch_bounds__(256) void march_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ near, const float* __restrict__ far,
                                                    const float* __restrict__ u, const uint8_t* __restrict__ occ, int res,
                                                    float inv2r, float step, float half_step, int max_samples, int64_t B,
                                                    int32_t* __restrict__ cnt, const int64_t* __restrict__ off,
                                                    const int32_t* __restrict__ keep,
                                                    float* __restrict__ t_start, float* __restrict__ pts,
                                                    float* __restrict__ dirs_pts, int32_t* __restrict__ ray_idx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= B) return;
    MarchRay r;
#pragma unroll
    for (int c = 0; c < 3; ++c) { r.o[c] = rays_o[ray * 3 + c]; r.d[c] = rays_d[ray * 3 + c]; }
    r.near = near[ray]; r.far = far[ray]; r.u = u ? u[ray] : 0.5f;
    // steps can only qualify while t_k + step <= far: an upper bound on k (one spare trip covers fp32 rounding)
    const int kmax = (int)ceilf((r.far - r.near) / step) + 1;
    const int64_t base = EMIT ? off[ray] : 0;
    if (EMIT && keep) { const int kp = keep[ray]; max_samples = kp < max_samples ? (kp < 0 ? 0 : kp) : max_samples; }   // wave-uniform
    int c = 0;
    for (int k0 = 0; k0 < kmax && c < max_samples; k0 += 64) {
        float t0, x[3];
        const bool hit = march_test(r, k0 + lane, step, half_step, inv2r, res, occ, t0, x);
        const unsigned long long m = __ballot(hit);
        const int slot = c + __popcll(m & ((1ull << lane) - 1ull));
        if (EMIT && hit && slot < max_samples) {
            const int64_t gp = base + slot;
            t_start[gp] = t0;
            ray_idx[gp] = (int32_t)ray;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) { pts[gp * 3 + cc] = x[cc]; dirs_pts[gp * 3 + cc] = r.d[cc]; }
        }
        c += __popcll(m);
    }
    if (!EMIT && lane == 0) cnt[ray] = c < max_samples ? c : max_samples;
}

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }

int launch_march_count(const float* o, const float* d, const float* near, const float* far, const float* u, const uint8_t* occ,
                       int res, float radius, float step, float half_step, int max_samples, int64_t B, int32_t* cnt,
                       hipStream_t st) {
    hipLaunchKernelGGL(march_kernel<false>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, o, d, near, far, u, occ, res,
                       (float)(0.5 / (double)radius), step, half_step, max_samples, B, cnt, nullptr, nullptr, nullptr, nullptr, nullptr,
                       nullptr);
    return ok();
}
int launch_march_emit(const float* o, const float* d, const float* near, const float* far, const float* u, const uint8_t* occ,
                      int res, float radius, float step, float half_step, int max_samples, int64_t B, const int64_t* off,
                      const int32_t* keep, float* t_start, float* pts, float* dirs_pts, int32_t* ray_idx, hipStream_t st) {
    hipLaunchKernelGGL(march_kernel<true>, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, o, d, near, far, u, occ, res,
                       (float)(0.5 / (double)radius), step, half_step, max_samples, B, nullptr, off, keep, t_start, pts, dirs_pts, ray_idx);
    return ok();
}

}  // namespace dh
