// Per-ray kernels: mask-conditioned ray generation, coarse sampling, hierarchical up-sampling (one wave per ray),
// sorted merge, section mid-points, and the volume-rendering scan forward/backward (wave-level scans, no HBM
// intermediates).  HBM-bound / latency-bound integer+fp32 work -- no MFMA here by design.
#include "tile.h"
#include "kernels.h"

namespace dh {

// ---------------------------------------------------------------- wave64 primitives
__device__ __forceinline__ float wave_sum(float v) {
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
// exclusive multiplicative scan over lanes (lane 0 -> 1)
__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
    float x = v;
    DH_UNROLL for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(x, off);
        if (lane >= off) x *= y;
    }
    const float e = __shfl_up(x, 1);
    return lane == 0 ? 1.f : e;
}
// inclusive additive scan over lanes
__device__ __forceinline__ float wave_incl_sum(float v, int lane) {
    float x = v;
    DH_UNROLL for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    return x;
}
// exclusive suffix sum over lanes: sum of v over lanes > lane
__device__ __forceinline__ float wave_excl_suffix_sum(float v, int lane) {
    float x = v;
    DH_UNROLL for (int off = 1; off < 64; off <<= 1) {
        const float y = __shfl_down(x, off);
        if (lane + off < 64) x += y;
    }
    const float e = __shfl_down(x, 1);
    return lane == 63 ? 0.f : e;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// ---------------------------------------------------------------- a1 + a2: ray generation
// Frames resident in HBM: rgb u8 [F,H,W,3], label i8 [F,H,W] (1 object / 0 background / -1 hand:
// reference ObjTracker/run.py:66, utils/maskutils.py:24-28), normal u8 [F,H,W,3]; pose x_cam = R x_obj + T
// (run.py:166, vis.py:52) as R [F,9] row-major, T [F,3]; Kinv [9] row-major (K per run.py:119-123).
// rays [B,14] = o(3) d(3) rgb(3) obj(1) keep(1) mono_normal(3); near/far [B] = unit-sphere bounds (App. A.8).
__global__ __launch_bounds__(256) void gen_rays_kernel(const uint8_t* __restrict__ rgb, const int8_t* __restrict__ label,
                                                       const uint8_t* __restrict__ normal, const float* __restrict__ R,
                                                       const float* __restrict__ T, const float* __restrict__ Kinv,
                                                       int H, int W, int frame, const int64_t* __restrict__ px,
                                                       const int64_t* __restrict__ py, int64_t B, float* __restrict__ rays,
                                                       float* __restrict__ near, float* __restrict__ far) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    const int64_t x = px[i], y = py[i];
    const float u = (float)x, v = (float)y;
    float p[3], dcam[3];
    DH_UNROLL for (int r = 0; r < 3; ++r) p[r] = Kinv[r * 3 + 0] * u + Kinv[r * 3 + 1] * v + Kinv[r * 3 + 2];
    const float inv = 1.f / sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    DH_UNROLL for (int r = 0; r < 3; ++r) dcam[r] = p[r] * inv;
    const float* Rf = R + (int64_t)frame * 9;
    const float* Tf = T + (int64_t)frame * 3;
    float o[3], d[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) {          // d = R^T dcam ; o = -R^T T
        d[c] = Rf[0 * 3 + c] * dcam[0] + Rf[1 * 3 + c] * dcam[1] + Rf[2 * 3 + c] * dcam[2];
        o[c] = -(Rf[0 * 3 + c] * Tf[0] + Rf[1 * 3 + c] * Tf[1] + Rf[2 * 3 + c] * Tf[2]);
    }
    const int64_t pix = ((int64_t)frame * H + y) * W + x;
    float* out = rays + i * 14;
    DH_UNROLL for (int c = 0; c < 3; ++c) { out[c] = o[c]; out[3 + c] = d[c]; }
    DH_UNROLL for (int c = 0; c < 3; ++c) out[6 + c] = (float)rgb[pix * 3 + c] / 255.0f;
    const int lab = (int)label[pix];
    out[9] = lab > 0 ? 1.f : 0.f;
    out[10] = lab >= 0 ? 1.f : 0.f;
    DH_UNROLL for (int c = 0; c < 3; ++c) out[11 + c] = (float)normal[pix * 3 + c] / 255.0f * 2.0f - 1.0f;
    const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float b = 2.f * (o[0] * d[0] + o[1] * d[1] + o[2] * d[2]);
    const float mid = 0.5f * (-b) / a;
    near[i] = mid - 1.f;
    far[i] = mid + 1.f;
}

// ---------------------------------------------------------------- a3: coarse samples (+ per-ray perturbation)
// z[b,j] = near + (far-near) * j/(n-1) + (t_rand[b]-0.5) * 2/n  (t_rand may be null = no perturbation); pts = o + d z.
__global__ __launch_bounds__(256) void coarse_samples_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                             const float* __restrict__ near, const float* __restrict__ far,
                                                             const float* __restrict__ t_rand, int64_t B, int n,
                                                             float* __restrict__ z, float* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * n) return;
    const int64_t b = i / n;
    const int j = (int)(i - b * n);
    const float lin = (n > 1) ? (float)j / (float)(n - 1) : 0.f;
    float zz = near[b] + (far[b] - near[b]) * lin;
    if (t_rand) zz += (t_rand[b] - 0.5f) * 2.0f / (float)n;
    z[i] = zz;
    DH_UNROLL for (int c = 0; c < 3; ++c) pts[i * 3 + c] = rays_o[b * 3 + c] + rays_d[b * 3 + c] * zz;
}

// ---------------------------------------------------------------- a6: one up-sampling step (App. A.6)
// One wave per ray, elements 2*lane, 2*lane+1.  n <= 128.  Produces n_new = 16 (<= 64) deterministic
// inverse-CDF samples and their positions.
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                       const float* __restrict__ z, const float* __restrict__ sdf,
                                                       int64_t B, int n, int n_new, float inv_s,
                                                       float* __restrict__ z_new, float* __restrict__ pts_new) {
    __shared__ float s_cdf[4][132];
    __shared__ float s_z[4][132];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= B) return;                       // whole wave exits together (no block-level sync below)
    const float ox = rays_o[ray * 3], oy = rays_o[ray * 3 + 1], oz = rays_o[ray * 3 + 2];
    const float dx = rays_d[ray * 3], dy = rays_d[ray * 3 + 1], dz = rays_d[ray * 3 + 2];
    const int e0 = 2 * lane, e1 = e0 + 1;
    const float* zr = z + ray * n;
    const float* sr = sdf + ray * n;
    const float z0 = e0 < n ? zr[e0] : 0.f, z1 = e1 < n ? zr[e1] : 0.f;
    const float f0 = e0 < n ? sr[e0] : 0.f, f1 = e1 < n ? sr[e1] : 0.f;
    auto radius = [&](float t) {
        const float x = ox + dx * t, y = oy + dy * t, w = oz + dz * t;
        return sqrtf(x * x + y * y + w * w);
    };
    const float r0 = radius(z0), r1 = radius(z1);
    // element 2*lane+2 (first element of the next lane)
    const float z2 = __shfl_down(z0, 1), f2 = __shfl_down(f0, 1), r2 = __shfl_down(r0, 1);
    // sections: sA = (e0,e1) index e0 ; sB = (e1,e2) index e1.  valid if index < n-1
    const bool vA = e0 < n - 1, vB = e1 < n - 1;
    const float cosA_raw = (f1 - f0) / (z1 - z0 + 1e-5f);
    const float cosB_raw = (f2 - f1) / (z2 - z1 + 1e-5f);
    float prevA = __shfl_up(cosB_raw, 1);
    if (lane == 0) prevA = 0.f;
    const float insA = ((r0 < 1.f) || (r1 < 1.f)) ? 1.f : 0.f;
    const float insB = ((r1 < 1.f) || (r2 < 1.f)) ? 1.f : 0.f;
    const float cosA = fminf(fmaxf(fminf(prevA, cosA_raw), -1e3f), 0.f) * insA;
    const float cosB = fminf(fmaxf(fminf(cosA_raw, cosB_raw), -1e3f), 0.f) * insB;
    auto alpha_of = [&](float fa, float fb, float za, float zb, float cs) {
        const float mid = (fa + fb) * 0.5f, dist = zb - za;
        const float pe = mid - cs * dist * 0.5f, ne = mid + cs * dist * 0.5f;
        const float pc = sigmoidf(pe * inv_s), nc = sigmoidf(ne * inv_s);
        return (pc - nc + 1e-5f) / (pc + 1e-5f);
    };
    const float aA = vA ? alpha_of(f0, f1, z0, z1, cosA) : 0.f;
    const float aB = vB ? alpha_of(f1, f2, z1, z2, cosB) : 0.f;
    const float tA = 1.f - aA + 1e-7f, tB = 1.f - aB + 1e-7f;
    const float excl = wave_excl_prod(tA * tB, lane);
    const float wA = vA ? aA * excl + 1e-5f : 0.f;           // weights + 1e-5 (sample_pdf)
    const float wB = vB ? aB * excl * tA + 1e-5f : 0.f;
    const float tot = wave_sum(wA + wB);
    const float pA = wA / tot, pB = wB / tot;
    const float incl = wave_incl_sum(pA + pB, lane);        // cdf after section e1
    // cdf has n entries: cdf[0] = 0, cdf[i+1] = sum_{j<=i} pdf_j
    float* cdf = s_cdf[wave];
    float* zs = s_z[wave];
    if (lane == 0) cdf[0] = 0.f;
    if (vA) cdf[e0 + 1] = incl - pB;
    if (vB) cdf[e1 + 1] = incl;
    if (e0 < n) zs[e0] = z0;
    if (e1 < n) zs[e1] = z1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < n_new) {
        const float u = (0.5f + (float)lane) / (float)n_new;
        // inds = #entries of cdf[0..n) <= u  (searchsorted right=True); cdf non-decreasing
        int lo = 0, hi = n;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
        }
        const int below = max(lo - 1, 0), above = min(lo, n - 1);
        const float c0 = cdf[below], c1 = cdf[above];
        float denom = c1 - c0;
        if (denom < 1e-5f) denom = 1.f;
        const float t = (u - c0) / denom;
        const float b0 = zs[below], b1 = zs[above];
        const float zn = b0 + t * (b1 - b0);
        z_new[ray * n_new + lane] = zn;
        float* pp = pts_new + (ray * n_new + lane) * 3;
        pp[0] = ox + dx * zn; pp[1] = oy + dy * zn; pp[2] = oz + dz * zn;
    }
}

// ---------------------------------------------------------------- a6: cat_z_vals = stable sorted merge
// old z [B,n] ascending, new z [B,n_new] ascending (inverse CDF of increasing u).  rank(old i) = i + #new < z_i,
// rank(new k) = k + #old <= znew_k  == torch.sort(cat[z, z_new], stable) ; sdf gathered alongside (skipped if null).
__global__ __launch_bounds__(256) void merge_kernel(const float* __restrict__ z, const float* __restrict__ z_new,
                                                    const float* __restrict__ sdf, const float* __restrict__ sdf_new,
                                                    int64_t B, int n, int n_new, float* __restrict__ z_out,
                                                    float* __restrict__ sdf_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= B) return;
    const int e0 = 2 * lane, e1 = e0 + 1;
    const float INF = __builtin_inff();
    const float z0 = e0 < n ? z[ray * n + e0] : INF, z1 = e1 < n ? z[ray * n + e1] : INF;
    const float zn = lane < n_new ? z_new[ray * n_new + lane] : INF;
    int c0 = 0, c1 = 0, cn = 0;
    for (int k = 0; k < n_new; ++k) {
        const float v = __shfl(zn, k);
        c0 += (v < z0) ? 1 : 0;
        c1 += (v < z1) ? 1 : 0;
    }
    for (int k = 0; k < n_new; ++k) {
        const float v = __shfl(zn, k);
        const int cnt = __popcll(__ballot(z0 <= v)) + __popcll(__ballot(z1 <= v));
        if (lane == k) cn = cnt;
    }
    const int no = n + n_new;
    float* zo = z_out + ray * no;
    if (e0 < n) zo[e0 + c0] = z0;
    if (e1 < n) zo[e1 + c1] = z1;
    if (lane < n_new) zo[lane + cn] = zn;
    if (sdf_out) {
        float* so = sdf_out + ray * no;
        if (e0 < n) so[e0 + c0] = sdf[ray * n + e0];
        if (e1 < n) so[e1 + c1] = sdf[ray * n + e1];
        if (lane < n_new) so[lane + cn] = sdf_new[ray * n_new + lane];
    }
}

// ---------------------------------------------------------------- a10 prologue: section mid-points
// dists = cat[z_{j+1}-z_j, sample_dist]; mid = z + dists/2; pts = o + d*mid  (render_core, App. A.7)
__global__ __launch_bounds__(256) void midpoints_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                        const float* __restrict__ z, int64_t B, int n, float sample_dist,
                                                        float* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * n) return;
    const int64_t b = i / n;
    const int j = (int)(i - b * n);
    const float dist = (j + 1 < n) ? z[i + 1] - z[i] : sample_dist;
    const float mid = z[i] + dist * 0.5f;
    DH_UNROLL for (int c = 0; c < 3; ++c) pts[i * 3 + c] = rays_o[b * 3 + c] + rays_d[b * 3 + c] * mid;
}

// ---------------------------------------------------------------- a10: volume-rendering scan, forward
struct RenderElem {
    float alpha, alpha_raw, cden, prev, next, ep, en, dist, dtc;   // dtc = d iter_cos / d true_cos
    float relax, nn;                                              // relax-inside flag, ||n||
    float n[3], c[3];
};

__device__ __forceinline__ RenderElem render_elem(bool valid, int64_t gp, float zc, float znext, bool last, float sample_dist,
                                                  const float (&o)[3], const float (&d)[3], const float* __restrict__ sdf,
                                                  const float* __restrict__ normals, const float* __restrict__ colors,
                                                  float inv_s, float car, float& inside) {
    RenderElem e{};
    inside = 0.f;
    if (!valid) return e;
    e.dist = last ? sample_dist : znext - zc;
    const float mid = zc + e.dist * 0.5f;
    float pn2 = 0.f;
    DH_UNROLL for (int c = 0; c < 3; ++c) { const float p = o[c] + d[c] * mid; pn2 += p * p; }
    const float pn = sqrtf(pn2);
    inside = pn < 1.0f ? 1.f : 0.f;
    e.relax = pn < 1.2f ? 1.f : 0.f;
    const float s = sdf[gp];
    float tc = 0.f, n2 = 0.f;
    DH_UNROLL for (int c = 0; c < 3; ++c) {
        e.n[c] = normals[gp * 3 + c];
        e.c[c] = colors[gp * 3 + c];
        tc += d[c] * e.n[c];
        n2 += e.n[c] * e.n[c];
    }
    e.nn = sqrtf(n2);
    const float a1 = -tc * 0.5f + 0.5f, a2 = -tc;
    const float ic = -(fmaxf(a1, 0.f) * (1.f - car) + fmaxf(a2, 0.f) * car);
    e.dtc = (a1 > 0.f ? 0.5f * (1.f - car) : 0.f) + (a2 > 0.f ? car : 0.f);
    e.en = s + ic * e.dist * 0.5f;
    e.ep = s - ic * e.dist * 0.5f;
    e.prev = sigmoidf(e.ep * inv_s);
    e.next = sigmoidf(e.en * inv_s);
    e.cden = e.prev + 1e-5f;
    e.alpha_raw = (e.prev - e.next + 1e-5f) / e.cden;
    e.alpha = fminf(fmaxf(e.alpha_raw, 0.f), 1.f);
    return e;
}

// outputs: weights [B,n], color [B,3], wsum [B], wmax [B], cdf [B,n] (prev_cdf), inside [B,n], eik [B,2] (num, den)
// One wave per ray, two samples per lane, 128 samples per trip.  The NeuS sampler has n <= 128 (one trip).  Packed rays
// (occupancy-grid marching: the ray owns the segment [seg_off, seg_off + seg_cnt) of the packed arrays, z holds the interval starts
// and every interval is sample_dist long) may be longer: further trips carry the transmittance reached so far.
constexpr int RENDER_MAX_CHUNKS = 8;           // packed segments up to 1024 samples
__global__ __launch_bounds__(256) void render_fwd_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z, const float* __restrict__ sdf,
                                                         const float* __restrict__ normals, const float* __restrict__ colors,
                                                         const float* __restrict__ inv_s_p, float car, float sample_dist,
                                                         const float* __restrict__ bg_rgb, int64_t B, int n,
                                                         float* __restrict__ weights, float* __restrict__ color_out,
                                                         float* __restrict__ wsum_out, float* __restrict__ wmax_out,
                                                         float* __restrict__ cdf_out, float* __restrict__ inside_out,
                                                         float* __restrict__ eik_out, float* __restrict__ nmap_out,
                                                         const int64_t* __restrict__ seg_off, const int32_t* __restrict__ seg_cnt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= B) return;
    const float inv_s = inv_s_p[0];
    float o[3], d[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) { o[c] = rays_o[ray * 3 + c]; d[c] = rays_d[ray * 3 + c]; }
    const bool packed = seg_off != nullptr;
    const int64_t rb = packed ? seg_off[ray] : ray * n;
    if (packed) n = min(seg_cnt[ray], 128 * RENDER_MAX_CHUNKS);
    float T_in = 1.f, ws = 0.f, wm = 0.f, en = 0.f, ed = 0.f;
    float col[3] = {0.f, 0.f, 0.f}, nm[3] = {0.f, 0.f, 0.f};
    for (int c0 = 0; c0 == 0 || c0 < n; c0 += 128) {
        const int e0 = c0 + 2 * lane, e1 = e0 + 1;
        const float z0 = e0 < n ? z[rb + e0] : 0.f, z1 = e1 < n ? z[rb + e1] : 0.f;
        float z2 = __shfl_down(z0, 1);
        if (lane == 63) z2 = (e1 + 1 < n) ? z[rb + e1 + 1] : 0.f;       // first sample of the next trip (never used by a last sample)
        float in0, in1;
        const RenderElem A = render_elem(e0 < n, rb + e0, z0, z1, packed || e0 == n - 1, sample_dist, o, d, sdf, normals, colors, inv_s, car, in0);
        const RenderElem Bq = render_elem(e1 < n, rb + e1, z1, z2, packed || e1 == n - 1, sample_dist, o, d, sdf, normals, colors, inv_s, car, in1);
        const float tA = (e0 < n) ? 1.f - A.alpha + 1e-7f : 1.f, tB = (e1 < n) ? 1.f - Bq.alpha + 1e-7f : 1.f;
        const float excl = T_in * wave_excl_prod(tA * tB, lane);
        const float w0 = A.alpha * excl, w1 = Bq.alpha * excl * tA;
        ws += wave_sum(w0 + w1);
        wm = fmaxf(wm, wave_max(fmaxf(w0, w1)));
        DH_UNROLL for (int c = 0; c < 3; ++c) col[c] += wave_sum(w0 * A.c[c] + w1 * Bq.c[c]);
        DH_UNROLL for (int c = 0; c < 3; ++c) nm[c] += wave_sum(w0 * A.n[c] + w1 * Bq.n[c]);
        const float g0 = (A.nn - 1.f) * (A.nn - 1.f) * A.relax, g1 = (Bq.nn - 1.f) * (Bq.nn - 1.f) * Bq.relax;
        en += wave_sum(g0 + g1); ed += wave_sum(A.relax + Bq.relax);
        if (e0 < n) { weights[rb + e0] = w0; cdf_out[rb + e0] = A.prev; inside_out[rb + e0] = in0; }
        if (e1 < n) { weights[rb + e1] = w1; cdf_out[rb + e1] = Bq.prev; inside_out[rb + e1] = in1; }
        T_in = __shfl(excl * tA * tB, 63);                               // transmittance after this trip's last sample
    }
    if (lane == 0) {
        DH_UNROLL for (int c = 0; c < 3; ++c) color_out[ray * 3 + c] = col[c] + (bg_rgb ? bg_rgb[c] * (1.f - ws) : 0.f);
        wsum_out[ray] = ws;
        wmax_out[ray] = wm;
        eik_out[ray * 2 + 0] = en;
        eik_out[ray * 2 + 1] = ed;
        if (nmap_out) { DH_UNROLL for (int c = 0; c < 3; ++c) nmap_out[ray * 3 + c] = nm[c]; }
    }
}

// ---------------------------------------------------------------- a10/a12: volume-rendering scan, backward
// inputs: d_color [B,3], d_wsum [B], d_weights [B,n] (nullable), d_gradients [P,3] (nullable),
//         eik_coef[0] = d_gradient_error / (sum relax + 1e-5)   (device scalar)
// outputs: d_sdf [P], d_normals [P,3], d_colors [P,3] (wrt post-sigmoid colour), d_inv_s [B] (per-ray partial)
// A sample's adjoint needs the transmittance before it and the sum q = wbar * w over the samples behind it.  With more than one
// 128-sample trip (packed rays only) a first sweep records each trip's entry transmittance and q total, the second sweep runs the
// per-trip scans with those carries.
struct RenderPair { RenderElem A, Bq; float tA, tB, wb0, wb1; bool v0, v1; };
__global__ __launch_bounds__(256) void render_bwd_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z, const float* __restrict__ sdf,
                                                         const float* __restrict__ normals, const float* __restrict__ colors,
                                                         const float* __restrict__ inv_s_p, float car, float sample_dist,
                                                         const float* __restrict__ bg_rgb, int64_t B, int n,
                                                         const float* __restrict__ d_color, const float* __restrict__ d_wsum,
                                                         const float* __restrict__ d_weights, const float* __restrict__ d_gradients,
                                                         const float* __restrict__ d_nmap,
                                                         const float* __restrict__ eik_coef, float* __restrict__ d_sdf,
                                                         float* __restrict__ d_normals, float* __restrict__ d_colors,
                                                         float* __restrict__ d_inv_s, float* __restrict__ d_rays_d,
                                                         const int64_t* __restrict__ seg_off, const int32_t* __restrict__ seg_cnt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave;
    if (ray >= B) return;
    const bool packed = seg_off != nullptr;
    const int64_t rb = packed ? seg_off[ray] : ray * n;
    if (packed) n = min(seg_cnt[ray], 128 * RENDER_MAX_CHUNKS);
    const float inv_s = inv_s_p[0];
    const float ec = eik_coef[0];
    // per-trip carries of the first sweep live in REGISTERS (lane ch holds trip ch's entry transmittance and q total, read back
    // by __shfl): the waves of a workgroup leave / iterate divergently, so an LDS hand-off could not be fenced by a barrier
    float c_tin = 1.f, c_q = 0.f;
    float o[3], d[3], dC[3];
    DH_UNROLL for (int c = 0; c < 3; ++c) { o[c] = rays_o[ray * 3 + c]; d[c] = rays_d[ray * 3 + c]; dC[c] = d_color[ray * 3 + c]; }
    float dN[3] = {0.f, 0.f, 0.f};
    if (d_nmap) { DH_UNROLL for (int c = 0; c < 3; ++c) dN[c] = d_nmap[ray * 3 + c]; }
    float dws = d_wsum ? d_wsum[ray] : 0.f;
    if (bg_rgb) dws -= dC[0] * bg_rgb[0] + dC[1] * bg_rgb[1] + dC[2] * bg_rgb[2];
    auto load_pair = [&](int c0) {
        RenderPair P;
        const int e0 = c0 + 2 * lane, e1 = e0 + 1;
        P.v0 = e0 < n; P.v1 = e1 < n;
        const float z0 = P.v0 ? z[rb + e0] : 0.f, z1 = P.v1 ? z[rb + e1] : 0.f;
        float z2 = __shfl_down(z0, 1);
        if (lane == 63) z2 = (e1 + 1 < n) ? z[rb + e1 + 1] : 0.f;
        float in0, in1;
        P.A = render_elem(P.v0, rb + e0, z0, z1, packed || e0 == n - 1, sample_dist, o, d, sdf, normals, colors, inv_s, car, in0);
        P.Bq = render_elem(P.v1, rb + e1, z1, z2, packed || e1 == n - 1, sample_dist, o, d, sdf, normals, colors, inv_s, car, in1);
        P.tA = P.v0 ? 1.f - P.A.alpha + 1e-7f : 1.f; P.tB = P.v1 ? 1.f - P.Bq.alpha + 1e-7f : 1.f;
        P.wb0 = dws + dC[0] * P.A.c[0] + dC[1] * P.A.c[1] + dC[2] * P.A.c[2] + dN[0] * P.A.n[0] + dN[1] * P.A.n[1] + dN[2] * P.A.n[2];
        P.wb1 = dws + dC[0] * P.Bq.c[0] + dC[1] * P.Bq.c[1] + dC[2] * P.Bq.c[2] + dN[0] * P.Bq.n[0] + dN[1] * P.Bq.n[1] + dN[2] * P.Bq.n[2];
        if (d_weights) { if (P.v0) P.wb0 += d_weights[rb + e0]; if (P.v1) P.wb1 += d_weights[rb + e1]; }
        return P;
    };
    const int nchunks = n > 128 ? (n + 127) / 128 : 1;
    if (nchunks > 1) {                                     // sweep 1: per-trip entry transmittance and q total
        float T_in = 1.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            const RenderPair P = load_pair(ch * 128);
            const float excl = T_in * wave_excl_prod(P.tA * P.tB, lane);
            const float q = (P.v0 ? P.wb0 * P.A.alpha * excl : 0.f) + (P.v1 ? P.wb1 * P.Bq.alpha * excl * P.tA : 0.f);
            const float qs = wave_sum(q);
            if (lane == ch) { c_tin = T_in; c_q = qs; }         // T_in and qs are wave-uniform
            T_in = __shfl(excl * P.tA * P.tB, 63);
        }
    }
    float dinv = 0.f;
    float ddir[3] = {0.f, 0.f, 0.f};          // d loss / d rays_d through true_cos = d . n (pose refinement only)
    auto elem_bwd = [&](const RenderElem& E, bool valid, float ab, float w, int64_t gp) {
        if (!valid) return;
        const float ar = (E.alpha_raw >= 0.f && E.alpha_raw <= 1.f) ? ab : 0.f;
        const float pb = ar * (1.f - E.alpha_raw) / E.cden;
        const float nb = -ar / E.cden;
        const float xp = pb * E.prev * (1.f - E.prev), xn = nb * E.next * (1.f - E.next);
        d_sdf[gp] = inv_s * (xp + xn);
        const float icb = inv_s * E.dist * 0.5f * (xn - xp);
        dinv += xp * E.ep + xn * E.en;
        const float tcb = icb * E.dtc;
        DH_UNROLL for (int c = 0; c < 3; ++c) ddir[c] += tcb * E.n[c];
        const float ek = (E.nn > 0.f) ? ec * E.relax * 2.f * (E.nn - 1.f) / E.nn : 0.f;
        DH_UNROLL for (int c = 0; c < 3; ++c) {
            float g = tcb * d[c] + ek * E.n[c] + w * dN[c];
            if (d_gradients) g += d_gradients[gp * 3 + c];
            d_normals[gp * 3 + c] = g;
            d_colors[gp * 3 + c] = w * dC[c];
        }
    };
    for (int ch = 0; ch < nchunks; ++ch) {
        const RenderPair P = load_pair(ch * 128);
        float T_in = 1.f, q_after = 0.f;
        if (nchunks > 1) {
            T_in = __shfl(c_tin, ch);
            for (int k = ch + 1; k < nchunks; ++k) q_after += __shfl(c_q, k);
        }
        const float excl = T_in * wave_excl_prod(P.tA * P.tB, lane);
        const float T0 = excl, T1 = excl * P.tA;
        const float w0 = P.A.alpha * T0, w1 = P.Bq.alpha * T1;
        const float q0 = P.v0 ? P.wb0 * w0 : 0.f, q1 = P.v1 ? P.wb1 * w1 : 0.f;
        const float suf = wave_excl_suffix_sum(q0 + q1, lane) + q_after;
        const float ab0 = P.wb0 * T0 - (suf + q1) / P.tA;
        const float ab1 = P.wb1 * T1 - suf / P.tB;
        elem_bwd(P.A, P.v0, ab0, w0, rb + ch * 128 + 2 * lane);
        elem_bwd(P.Bq, P.v1, ab1, w1, rb + ch * 128 + 2 * lane + 1);
    }
    dinv = wave_sum(dinv);
    if (lane == 0) d_inv_s[ray] = dinv;
    if (d_rays_d) {
        DH_UNROLL for (int c = 0; c < 3; ++c) { const float v = wave_sum(ddir[c]); if (lane == 0) d_rays_d[ray * 3 + c] = v; }
    }
}

// ================================================================ launchers
static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -3; }

int launch_gen_rays(const uint8_t* rgb, const int8_t* label, const uint8_t* normal, const float* R, const float* T,
                    const float* Kinv, int H, int W, int frame, const int64_t* px, const int64_t* py, int64_t B,
                    float* rays, float* near, float* far, hipStream_t st) {
    hipLaunchKernelGGL(gen_rays_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, rgb, label, normal, R, T, Kinv,
                       H, W, frame, px, py, B, rays, near, far);
    return ok();
}
int launch_coarse_samples(const float* o, const float* d, const float* near, const float* far, const float* t_rand,
                          int64_t B, int n, float* z, float* pts, hipStream_t st) {
    hipLaunchKernelGGL(coarse_samples_kernel, dim3((unsigned)((B * n + 255) / 256)), dim3(256), 0, st, o, d, near, far, t_rand,
                       B, n, z, pts);
    return ok();
}
int launch_upsample(const float* o, const float* d, const float* z, const float* sdf, int64_t B, int n, int n_new,
                    float inv_s, float* z_new, float* pts_new, hipStream_t st) {
    hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, o, d, z, sdf, B, n, n_new, inv_s,
                       z_new, pts_new);
    return ok();
}
int launch_merge(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int64_t B, int n, int n_new,
                 float* z_out, float* sdf_out, hipStream_t st) {
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, z, z_new, sdf, sdf_new, B, n, n_new,
                       z_out, sdf_out);
    return ok();
}
int launch_midpoints(const float* o, const float* d, const float* z, int64_t B, int n, float sample_dist, float* pts,
                     hipStream_t st) {
    hipLaunchKernelGGL(midpoints_kernel, dim3((unsigned)((B * n + 255) / 256)), dim3(256), 0, st, o, d, z, B, n, sample_dist, pts);
    return ok();
}
int launch_render_fwd(const float* o, const float* d, const float* z, const float* sdf, const float* normals,
                      const float* colors, const float* inv_s, float car, float sample_dist, const float* bg, int64_t B,
                      int n, float* weights, float* color, float* wsum, float* wmax, float* cdf, float* inside, float* eik,
                      float* nmap, const int64_t* seg_off, const int32_t* seg_cnt, hipStream_t st) {
    hipLaunchKernelGGL(render_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, o, d, z, sdf, normals, colors, inv_s,
                       car, sample_dist, bg, B, n, weights, color, wsum, wmax, cdf, inside, eik, nmap, seg_off, seg_cnt);
    return ok();
}
int launch_render_bwd(const float* o, const float* d, const float* z, const float* sdf, const float* normals,
                      const float* colors, const float* inv_s, float car, float sample_dist, const float* bg, int64_t B,
                      int n, const float* d_color, const float* d_wsum, const float* d_weights, const float* d_gradients,
                      const float* d_nmap, const float* eik_coef, float* d_sdf, float* d_normals, float* d_colors,
                      float* d_inv_s, float* d_rays_d, const int64_t* seg_off, const int32_t* seg_cnt, hipStream_t st) {
    hipLaunchKernelGGL(render_bwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, o, d, z, sdf, normals, colors, inv_s,
                       car, sample_dist, bg, B, n, d_color, d_wsum, d_weights, d_gradients, d_nmap, eik_coef, d_sdf, d_normals,
                       d_colors, d_inv_s, d_rays_d, seg_off, seg_cnt);
    return ok();
}

}  // namespace dh
