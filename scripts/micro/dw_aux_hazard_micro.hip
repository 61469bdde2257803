// Standalone repro of the weight-gradient kernel's irreproducible two-piece AUX body (profiles/r04_dw_aux_reproducibility.json,
// VERDICT r4 "next" 1a).  This file INCLUDES csrc/dw.hip (job tables, scales, the main body) and dw_aux_variants.h (round 6: the aux
// body WITH its development switches and a kernel identical to dw_f16x2_kernel that calls it -- the product file holds the shipping
// body only).  With no -DDW_AUX_V_* / -DDW_AUX_THREE_PIECE switch the variants header compiles to the product's body, line for line;
// only the job list (run-time data) is synthetic.  No torch, no workspace: synthetic heavy-tailed A tiles, O(1) aux
// B tiles, exact per-tile / per-class maxima.
//
//   dw_aux_hazard_micro [launches] [aux_workgroups (1..256; the rest run main jobs)] [ntiles] [aux_jobs]
//
// Every launch's slabs are compared bit for bit with the first launch's; differing launches are counted and the first few are
// decoded (workgroup, job, output tile, n-tile, accumulator lanes / registers, magnitude).  One JSON line at the end.
#include "../../dynhor_amd/csrc/dw.hip"
#include "dw_aux_variants.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// tile t of `per` floats: |values| < m_t = 2^-e_t (e_t hashed, 0..emax; tile 0: e = 0), element 0 of the tile = m_t exactly
__global__ void fill_kernel(float* p, int64_t per, int64_t ntiles, unsigned seed, int emax, unsigned* tmax) {
    const int64_t t = blockIdx.x;
    const int e = t == 0 ? 0 : (int)(mix32(seed * 977u + (unsigned)t) % (unsigned)(emax + 1));
    const float m = ldexpf(1.f, -e);
    for (int64_t i = threadIdx.x; i < per; i += blockDim.x) {
        const unsigned h = mix32(seed + (unsigned)(t * per + i) * 2654435761u);
        const float u = ((float)(h >> 8) * (1.f / 16777216.f)) * 2.f - 1.f;           // (-1, 1)
        // heavy tail inside the tile too: a cube keeps most values small
        p[t * per + i] = i == 0 ? m : m * u * u * u;
    }
    if (threadIdx.x == 0 && tmax) tmax[t] = __builtin_bit_cast(unsigned, m);
}
__global__ void cmp_kernel(const unsigned* __restrict__ cur, const unsigned* __restrict__ ref, int64_t n, unsigned* out) {
    unsigned c = 0; int64_t first = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (cur[i] != ref[i]) { ++c; if (first < 0) first = i; }
    if (c) { atomicAdd(out, c); atomicMin(reinterpret_cast<unsigned long long*>(out + 2), (unsigned long long)first); }
}

int main(int argc, char** argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 2000;
    const int g_aux = argc > 2 ? atoi(argv[2]) : 256;
    const int64_t nt = argc > 3 ? atoll(argv[3]) : 4096;
    const int naux = argc > 4 ? atoi(argv[4]) : 3;
    const int G = 256;
    using namespace dh;
    float *A0, *A1, *B8a, *B8b, *Bx0, *Bx1;
    unsigned *absmax, *tmax;
    CK(hipMalloc(&A0, nt * TILE_F * 4)); CK(hipMalloc(&A1, nt * TILE_F * 4));
    CK(hipMalloc(&B8a, nt * TILE_F * 4)); CK(hipMalloc(&B8b, nt * TILE_F * 4));
    CK(hipMalloc(&Bx0, nt * AUXT_F * 4)); CK(hipMalloc(&Bx1, nt * AUXT_F * 4));
    CK(hipMalloc(&absmax, ABSMAX_FLOATS * 4)); CK(hipMalloc(&tmax, 4 * nt * 4));
    // classes: 0 = A0 (heavy, tmax 0), 1 = A1 (tame), 2 = Bx1 (heavy, tmax 1), 3 = B8b (heavy, tmax 2); Bx0 / B8a: the constant scale
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, A0, (int64_t)TILE_F, nt, 1u, 20, tmax + 0 * nt);
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, A1, (int64_t)TILE_F, nt, 2u, 3, (unsigned*)nullptr);
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, Bx0, (int64_t)AUXT_F, nt, 3u, 0, (unsigned*)nullptr);
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, Bx1, (int64_t)AUXT_F, nt, 4u, 20, tmax + 1 * nt);
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, B8a, (int64_t)TILE_F, nt, 5u, 0, (unsigned*)nullptr);
    hipLaunchKernelGGL(fill_kernel, dim3(nt), dim3(256), 0, 0, B8b, (int64_t)TILE_F, nt, 6u, 20, tmax + 2 * nt);
    {
        std::vector<unsigned> h(ABSMAX_FLOATS, 0u);
        const float one = 1.f;
        unsigned ob; memcpy(&ob, &one, 4);
        for (int c = 0; c < 4; ++c) h[c * ABSMAX_STRIDE] = ob;       // every class has its tile 0 at maximum 1.0
        h[ABSMAX_TAG * ABSMAX_STRIDE] = ABSMAX_TAG_F16;
        CK(hipMemcpy(absmax, h.data(), ABSMAX_FLOATS * 4, hipMemcpyHostToDevice));
    }
    // jobs: naux aux jobs (two operand pairs, as jobs 0 / 8 of the product), then main jobs (two pairs, as jobs 1..7)
    DwJobs J{};
    const int nmain = g_aux < G ? 5 : 0;
    int64_t off = 0;
    for (int j = 0; j < naux + nmain; ++j) {
        DwJob& q = J.j[j];
        const bool aux = j < naux;
        q.nb = aux ? 2 : 8;
        q.off = off; off += (int64_t)8 * q.nb * 1024;
        q.A1 = A0; q.B1 = aux ? Bx0 : B8a; q.A2 = A1; q.B2 = aux ? Bx1 : B8b;
        q.ca[0] = 0; q.ha[0] = 0; q.cb[0] = -1; q.hb[0] = -1;
        q.ca[1] = 1; q.ha[1] = -1; q.cb[1] = aux ? 2 : 3; q.hb[1] = aux ? 1 : 2;
    }
    J.n = naux + nmain;
    const int64_t gstride = off;
    DwGroups Gp{};
    if (nmain) { Gp.n = 2; Gp.job0[0] = 0; Gp.job0[1] = naux; Gp.job0[2] = J.n; Gp.wg0[0] = 0; Gp.wg0[1] = g_aux; Gp.wg0[2] = G; }
    else { Gp.n = 1; Gp.job0[0] = 0; Gp.job0[1] = J.n; Gp.wg0[0] = 0; Gp.wg0[1] = G; }
    float *slabs, *ref, *prev;
    unsigned *out, *out2;
    CK(hipMalloc(&slabs, G * gstride * 4)); CK(hipMalloc(&ref, G * gstride * 4)); CK(hipMalloc(&prev, G * gstride * 4));
    CK(hipMalloc(&out, 16)); CK(hipMalloc(&out2, 16));
    CK(hipMemset(slabs, 0, G * gstride * 4));
    CK(hipDeviceSynchronize());
    auto launch = [&]() {
        hipLaunchKernelGGL(dw_f16x2_variant_kernel, dim3(G), dim3(512), 0, 0, J, Gp, nt, slabs, gstride, (const unsigned*)absmax, (const unsigned*)tmax);
    };
    launch();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref, slabs, G * gstride * 4, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(prev, slabs, G * gstride * 4, hipMemcpyDeviceToDevice));
    // sanity: the reference is finite and non-zero
    {
        std::vector<float> h(gstride);
        CK(hipMemcpy(h.data(), ref, gstride * 4, hipMemcpyDeviceToHost));
        double s = 0; int bad = 0;
        for (float v : h) { if (!std::isfinite(v)) ++bad; s += std::fabs(v); }
        fprintf(stderr, "reference slab 0: sum|x| = %.6g, non-finite %d, gstride %lld, jobs %d aux + %d main, aux workgroups %d\n", s, bad,
                (long long)gstride, naux, nmain, nmain ? g_aux : G);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int bad_launches = 0, shown = 0, bad_vs_prev = 0;
    long long bad_words = 0;
    std::set<int> lanes_seen, regs_seen, ntile_seen, job_seen;
    float ms_total = 0.f;
    for (int rep = 0; rep < launches; ++rep) {
        unsigned long long init[2] = {0ull, ~0ull};
        CK(hipMemcpyAsync(out, init, 16, hipMemcpyHostToDevice, 0));
        CK(hipMemcpyAsync(out2, init, 16, hipMemcpyHostToDevice, 0));
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(cmp_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)slabs, (const unsigned*)ref, G * gstride, out);
        hipLaunchKernelGGL(cmp_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned*)slabs, (const unsigned*)prev, G * gstride, out2);
        unsigned long long res[2], res2[2];
        CK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(res2, out2, 16, hipMemcpyDeviceToHost));
        if (res2[0] & 0xffffffffu) { ++bad_vs_prev; CK(hipMemcpy(prev, slabs, G * gstride * 4, hipMemcpyDeviceToDevice)); }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_total += ms;
        const unsigned nd = (unsigned)(res[0] & 0xffffffffu);
        if (nd) {
            ++bad_launches; bad_words += nd;
            const int64_t first = (int64_t)res[1];
            const int wg = (int)(first / gstride);
            std::vector<float> hc(gstride), hr(gstride);
            CK(hipMemcpy(hc.data(), slabs + (int64_t)wg * gstride, gstride * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hr.data(), ref + (int64_t)wg * gstride, gstride * 4, hipMemcpyDeviceToHost));
            double maxd = 0, maxr = 0;
            std::set<int> ln, rg, ntl, otl, jb;
            for (int64_t i = 0; i < gstride; ++i) {
                maxr = std::fmax(maxr, std::fabs(hr[i]));
                if (memcmp(&hc[i], &hr[i], 4)) {
                    maxd = std::fmax(maxd, std::fabs((double)hc[i] - hr[i]));
                    int j = 0; while (j + 1 < J.n && i >= J.j[j + 1].off) ++j;
                    const int64_t w = i - J.j[j].off;
                    const int tile = (int)(w / 1024), r = (int)(w % 1024) / 64, lane = (int)(w % 64);
                    jb.insert(j); otl.insert(tile / J.j[j].nb); ntl.insert(tile % J.j[j].nb); rg.insert(r); ln.insert(lane);
                    lanes_seen.insert(lane); regs_seen.insert(r); ntile_seen.insert(tile % J.j[j].nb); job_seen.insert(j);
                }
            }
            if (shown < 8) {
                ++shown;
                fprintf(stderr, "launch %d: %u words differ; first workgroup %d: jobs {", rep, nd, wg);
                for (int v : jb) fprintf(stderr, "%d ", v);
                fprintf(stderr, "} out tiles {"); for (int v : otl) fprintf(stderr, "%d ", v);
                fprintf(stderr, "} n tiles {"); for (int v : ntl) fprintf(stderr, "%d ", v);
                fprintf(stderr, "} regs %zu lanes {", rg.size()); for (int v : ln) fprintf(stderr, "%d ", v);
                fprintf(stderr, "} max diff / max |ref| = %.3g\n", maxd / (maxr > 0 ? maxr : 1));
            }
        }
    }
    printf("{\"variant\": \"%s\", \"launches\": %d, \"differing\": %d, \"differing_from_previous\": %d, \"words\": %lld, \"ms_per_launch\": %.4f, \"aux_workgroups\": %d, \"ntiles\": %lld, "
           "\"aux_jobs\": %d, \"main_jobs\": %d, \"lanes\": [",
#ifdef VARIANT_NAME
           VARIANT_NAME,
#else
           "default",
#endif
           launches, bad_launches, bad_vs_prev, bad_words, ms_total / launches, nmain ? g_aux : G, (long long)nt, naux, nmain);
    { bool f = true; for (int v : lanes_seen) { printf("%s%d", f ? "" : ",", v); f = false; } }
    printf("], \"n_tiles\": ["); { bool f = true; for (int v : ntile_seen) { printf("%s%d", f ? "" : ",", v); f = false; } }
    printf("], \"jobs\": ["); { bool f = true; for (int v : job_seen) { printf("%s%d", f ? "" : ",", v); f = false; } }
    printf("]}\n");
    return 0;
}
