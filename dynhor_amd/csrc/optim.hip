// Fused Adam over the flat parameter vector (torch.optim.Adam semantics) and the fused loss.
#include "tile.h"
#include "kernels.h"

namespace dh {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

// ---------------------------------------------------------------- a11: losses + their adjoints, one workgroup
__device__ __forceinline__ float block_sum(float v, float* s_red) {
    DH_UNROLL for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];     // fixed order -> deterministic
    return t;
}

__global__ __launch_bounds__(1024) void loss_kernel(const float* __restrict__ color, const float* __restrict__ wsum,
                                                    const float* __restrict__ nmap, const float* __restrict__ eik,
                                                    const float* __restrict__ rays, const float* __restrict__ R, int64_t B,
                                                    float igr_w, float mask_w, float normal_w, float* __restrict__ stats,
                                                    float* __restrict__ d_color, float* __restrict__ d_wsum,
                                                    float* __restrict__ d_nmap, float* __restrict__ eik_coef) {
    __shared__ float s_red[16];
    const int tid = threadIdx.x;
    float a_m = 0.f, a_k = 0.f, a_en = 0.f, a_ed = 0.f;
    for (int64_t b = tid; b < B; b += blockDim.x) {
        const float obj = rays[b * 14 + 9], keep = rays[b * 14 + 10];
        a_m += obj * keep; a_k += keep; a_en += eik[b * 2]; a_ed += eik[b * 2 + 1];
    }
    const float msum = block_sum(a_m, s_red) + 1e-5f;
    const float ksum = block_sum(a_k, s_red) + 1e-5f;
    const float en = block_sum(a_en, s_red), ed = block_sum(a_ed, s_red);
    float Rm[9];
    if (normal_w > 0.f) { DH_UNROLL for (int i = 0; i < 9; ++i) Rm[i] = R[i]; }
    float a_c = 0.f, a_sq = 0.f, a_bce = 0.f, a_n = 0.f;
    for (int64_t b = tid; b < B; b += blockDim.x) {
        const float obj = rays[b * 14 + 9], keep = rays[b * 14 + 10];
        const float m = obj * keep;
        DH_UNROLL for (int c = 0; c < 3; ++c) {
            const float e = color[b * 3 + c] - rays[b * 14 + 6 + c];
            a_c += fabsf(e) * m;
            a_sq += e * e * m;
            d_color[b * 3 + c] = (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * m / msum;
        }
        const float ws = wsum[b];
        const float wc = fminf(fmaxf(ws, 1e-3f), 1.f - 1e-3f);
        a_bce += -(obj * logf(wc) + (1.f - obj) * logf(1.f - wc)) * keep;
        const float pass = (ws >= 1e-3f && ws <= 1.f - 1e-3f) ? 1.f : 0.f;
        d_wsum[b] = mask_w * keep / ksum * (-obj / wc + (1.f - obj) / (1.f - wc)) * pass;
        if (normal_w > 0.f) {
            float no[3], nc[3], mono[3], nh[3], dnh[3];
            DH_UNROLL for (int c = 0; c < 3; ++c) { no[c] = nmap[b * 3 + c]; mono[c] = rays[b * 14 + 11 + c]; }
            DH_UNROLL for (int i = 0; i < 3; ++i) nc[i] = Rm[i * 3] * no[0] + Rm[i * 3 + 1] * no[1] + Rm[i * 3 + 2] * no[2];
            const float nn = sqrtf(nc[0] * nc[0] + nc[1] * nc[1] + nc[2] * nc[2]);
            const float nrm = nn + 1e-6f;
            float l1 = 0.f, cs = 0.f, dd = 0.f;
            DH_UNROLL for (int i = 0; i < 3; ++i) {
                nh[i] = nc[i] / nrm;
                const float e = nh[i] - mono[i];
                l1 += fabsf(e);
                cs += nh[i] * mono[i];
                dnh[i] = normal_w * m / msum * ((e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) - mono[i]);
                dd += dnh[i] * nc[i];
            }
            a_n += (l1 + 1.f - cs) * m;
            float dnc[3];
            DH_UNROLL for (int i = 0; i < 3; ++i) dnc[i] = dnh[i] / nrm - (nn > 0.f ? nc[i] * dd / (nn * nrm * nrm) : 0.f);
            DH_UNROLL for (int j = 0; j < 3; ++j) d_nmap[b * 3 + j] = Rm[j] * dnc[0] + Rm[3 + j] * dnc[1] + Rm[6 + j] * dnc[2];
        }
    }
    const float csum = block_sum(a_c, s_red), sq = block_sum(a_sq, s_red), bce = block_sum(a_bce, s_red);
    const float nsum = block_sum(a_n, s_red);
    if (tid == 0) {
        const float closs = csum / msum, ge = en / (ed + 1e-5f), mloss = bce / ksum, nloss = nsum / msum;
        stats[0] = closs + igr_w * ge + mask_w * mloss + (normal_w > 0.f ? normal_w * nloss : 0.f);
        stats[1] = closs; stats[2] = ge; stats[3] = mloss; stats[4] = nloss;
        stats[5] = 20.f * log10f(1.f / sqrtf(sq / (msum * 3.f)));
        stats[6] = msum; stats[7] = ksum;
        eik_coef[0] = igr_w / (ed + 1e-5f);
    }
}

// ---------------------------------------------------------------- n4: dense-correspondence reprojection loss (one workgroup)
// Specification: oracle/neus_oracle.py:correspondence_loss (the reference names only the input folder,
// README.md:43).  corr [B,4] = (u_j, v_j, certainty, frame_j).  Pass 1: one ray per thread -> expected depth t^ = sum_k w_k m_k,
// x = o + t^ d, reprojection into frame j, Huber cost and d cost / d t^ (un-normalised, kept in residual_px's neighbour
// buffer dt); fixed-order block sums -> deterministic.  Pass 2: d_weights[r,k] = dt[r] / (sum c v + 1e-5) * m_k, coalesced.
__global__ __launch_bounds__(1024) void corr_loss_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z, const float* __restrict__ weights,
                                                         const float* __restrict__ corr, const float* __restrict__ R_all,
                                                         const float* __restrict__ T_all, int n_frames,
                                                         const float* __restrict__ K, int64_t B, int n, float sample_dist,
                                                         float delta_px, float corr_w, float* __restrict__ stats,
                                                         float* __restrict__ residual_px, float* __restrict__ d_weights,
                                                         float* __restrict__ pose_adj) {
    __shared__ float s_red[16];
    const int tid = threadIdx.x;
    const float fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const float delta = delta_px / fx;
    float a_cv = 0.f, a_l = 0.f, a_r = 0.f;
    for (int64_t b = tid; b < B; b += blockDim.x) {
        const float conf = corr[b * 4 + 2];
        float dt = 0.f;
        // a match that cannot be evaluated (no partner frame, point behind the partner camera) is a GROSS failure, not a perfect
        // one: its residual is +inf so that the outlier vote counts it against its frame pair; certainty-0 rays
        // (no match at all) report 0
        float res = 0.f;
        float dy[3] = {0.f, 0.f, 0.f}, dx[3] = {0.f, 0.f, 0.f}, depth_out = 0.f;
        const int j = (int)corr[b * 4 + 3];
        if (conf > 0.f) res = __builtin_inff();
        if (conf > 0.f && j >= 0 && j < n_frames) {
            float depth = 0.f;
            const float* zr = z + b * n;
            const float* wr = weights + b * n;
            for (int k = 0; k < n; ++k) {
                const float dist = (k + 1 < n) ? zr[k + 1] - zr[k] : sample_dist;
                depth = fmaf(wr[k], zr[k] + 0.5f * dist, depth);
            }
            float x[3], y[3], g[3];
            DH_UNROLL for (int c = 0; c < 3; ++c) x[c] = rays_o[b * 3 + c] + depth * rays_d[b * 3 + c];
            const float* Rj = R_all + (int64_t)j * 9;
            DH_UNROLL for (int i = 0; i < 3; ++i) {
                y[i] = Rj[i * 3] * x[0] + Rj[i * 3 + 1] * x[1] + Rj[i * 3 + 2] * x[2] + T_all[j * 3 + i];
                g[i] = Rj[i * 3] * rays_d[b * 3] + Rj[i * 3 + 1] * rays_d[b * 3 + 1] + Rj[i * 3 + 2] * rays_d[b * 3 + 2];
            }
            if (y[2] > 1e-3f) {
                const float iz = 1.f / y[2];
                const float eu = fx * y[0] * iz + cx - corr[b * 4 + 0], ev = fy * y[1] * iz + cy - corr[b * 4 + 1];
                const float e2 = eu * eu + ev * ev;
                const float en = sqrtf(e2 + 1e-24f);
                const float sres = en / fx;
                res = en;
                const float rho = sres <= delta ? sres * sres / (2.f * delta) : sres - 0.5f * delta;
                const float dpu = fx * (g[0] * y[2] - y[0] * g[2]) * iz * iz, dpv = fy * (g[1] * y[2] - y[1] * g[2]) * iz * iz;
                const float ep = eu * dpu + ev * dpv;                    // e . d pi / d t^
                const float drho = sres <= delta ? ep / (fx * fx * delta) : ep / (en * fx);
                a_cv += conf; a_l += conf * rho; a_r += conf * en;
                dt = conf * drho;
                if (pose_adj) {
                    // d rho / d y (y = R_j x + T_j): kappa (eu d pu + ev d pv); d rho / d x = R_j^T (d rho / d y)
                    const float kappa = conf * (sres <= delta ? 1.f / (fx * fx * delta) : 1.f / (en * fx));
                    dy[0] = kappa * eu * fx * iz;
                    dy[1] = kappa * ev * fy * iz;
                    dy[2] = -kappa * (eu * fx * y[0] + ev * fy * y[1]) * iz * iz;
                    DH_UNROLL for (int c = 0; c < 3; ++c) dx[c] = Rj[c] * dy[0] + Rj[3 + c] * dy[1] + Rj[6 + c] * dy[2];
                }
            }
            depth_out = depth;
        }
        residual_px[b] = res;
        d_weights[b * n] = dt;          // parked in the ray's first slot until the normaliser is known
        if (pose_adj) {                 // [B,7] = d_x(3), d_y(3), t^ ; un-normalised until the second pass
            float* pa = pose_adj + b * 7;
            pa[0] = dx[0]; pa[1] = dx[1]; pa[2] = dx[2]; pa[3] = dy[0]; pa[4] = dy[1]; pa[5] = dy[2]; pa[6] = depth_out;
        }
    }
    const float cv = block_sum(a_cv, s_red), ls = block_sum(a_l, s_red), rs = block_sum(a_r, s_red);
    const float inv = corr_w / (cv + 1e-5f);
    __syncthreads();
    __shared__ float s_dt[1024];
    for (int64_t b0 = 0; b0 < B; b0 += blockDim.x) {      // a chunk of 1024 rays at a time: coefficients through LDS
        const int64_t b = b0 + tid;
        s_dt[tid] = b < B ? d_weights[b * n] * inv : 0.f;
        __syncthreads();
        const int64_t nb = (B - b0) < (int64_t)blockDim.x ? (B - b0) : (int64_t)blockDim.x;
        for (int64_t e = tid; e < nb * n; e += blockDim.x) {
            const int64_t r = e / n; const int k = (int)(e % n);
            const float* zr = z + (b0 + r) * n;
            const float dist = (k + 1 < n) ? zr[k + 1] - zr[k] : sample_dist;
            d_weights[(b0 + r) * n + k] = s_dt[r] * (zr[k] + 0.5f * dist);
        }
        __syncthreads();
    }
    if (pose_adj)
        for (int64_t e = tid; e < B * 7; e += blockDim.x) if (e % 7 != 6) pose_adj[e] *= inv;
    if (tid == 0) {
        stats[0] = ls / (cv + 1e-5f);            // L_corr (unweighted)
        stats[1] = cv;                           // sum of certainties of the valid matches
        stats[2] = rs / (cv + 1e-5f);            // certainty-weighted mean reprojection residual, pixels
        stats[3] = corr_w * stats[0];            // its contribution to the total loss
    }
}

int launch_loss(const float* color, const float* wsum, const float* nmap, const float* eik, const float* rays,
                const float* R, int64_t B, float igr_w, float mask_w, float normal_w, float* stats, float* d_color,
                float* d_wsum, float* d_nmap, float* eik_coef, hipStream_t st) {
    hipLaunchKernelGGL(loss_kernel, dim3(1), dim3(1024), 0, st, color, wsum, nmap, eik, rays, R, B, igr_w, mask_w, normal_w,
                       stats, d_color, d_wsum, d_nmap, eik_coef);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_corr_loss(const float* rays_o, const float* rays_d, const float* z, const float* weights, const float* corr,
                     const float* R_all, const float* T_all, int n_frames, const float* K, int64_t B, int n, float sample_dist,
                     float delta_px, float corr_w, float* stats, float* residual_px, float* d_weights, float* pose_adj,
                     hipStream_t st) {
    hipLaunchKernelGGL(corr_loss_kernel, dim3(1), dim3(1024), 0, st, rays_o, rays_d, z, weights, corr, R_all, T_all, n_frames, K, B, n,
                       sample_dist, delta_px, corr_w, stats, residual_px, d_weights, pose_adj);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int64_t step, float grad_scale, hipStream_t st) {
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps,
                       (float)bc1, (float)sqrt(bc2), grad_scale);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
