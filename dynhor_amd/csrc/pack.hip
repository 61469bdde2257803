// Weight preparation: weight-norm row scales, then MFMA-operand packing of every linear (layout.h).
// Runs once per optimiser step (weights change every iteration); ~5 MB written, L2-resident afterwards.
#include "tile16h.h"
#include "kernels.h"

namespace dh {

// one wave per row: rowscale = g / ||v||_2
__global__ __launch_bounds__(256) void rowscale_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.y;                    // 0..8 sdf, 9..13 colour
    const bool is_sdf = job < N_SDF;
    const int l = is_sdf ? job : job - N_SDF;
    int out, in;
    int64_t goff, voff;
    if (is_sdf) { out = SDF_DIMS[l].out; in = SDF_DIMS[l].in; goff = sdf_off(l).g; voff = sdf_off(l).v; }
    else        { out = COL_DIMS[l].out; in = COL_DIMS[l].in; goff = col_off(l).g; voff = col_off(l).v; }
    // a wave per row, 16 rows per workgroup; max |W| of the linear (the two-piece fp16 packs scale by a power of two derived from it:
    // tile16h.h) is reduced over the workgroup's rows first: ONE atomicMax per workgroup (one per row -- 2,313 atomics on 14
    // addresses -- made this kernel 40 us instead of 6)
    __shared__ float wmax[4];
    float rmax = 0.f;
    for (int rr = 0; rr < 4; ++rr) {
        const int row = (blockIdx.x * 4 + rr) * 4 + wave;
        if (row >= out) break;
        const float* v = params + voff + (int64_t)row * in;
        float s = 0.f, vmax = 0.f;
        for (int k = lane; k < in; k += 64) { s += v[k] * v[k]; vmax = fmaxf(vmax, fabsf(v[k])); }
        for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); vmax = fmaxf(vmax, __shfl_xor(vmax, off)); }
        const float inv = 1.f / sqrtf(s);
        const float gi = params[goff + row] * inv;
        if (lane == 0) {
            const int64_t ro = PACK.rowscale + (is_sdf ? (int64_t)l * 260 : (int64_t)N_SDF * 260 + (int64_t)l * 256);
            packed[ro + row] = gi;
            packed[ro + row + (PACK.invnorm - PACK.rowscale)] = inv;
        }
        rmax = fmaxf(rmax, fabsf(gi) * vmax);
    }
    if (lane == 0) wmax[wave] = rmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(packed + PACKH.wabs) + job, __builtin_bit_cast(unsigned, m));
    }
}

struct PackJob {
    int64_t dst;        // float offset into packed
    int64_t voff;       // float offset of weight_v in params
    int64_t rsoff;      // float offset of the rowscale block in packed
    int ldv;            // in-dim of the linear
    int nkg;            // k-groups
    int nt;             // n-tiles (8 or 2)
    int rev;            // 0: M[k=in][n=out]   1: M[k=out][n=in]   2 (pack16_kernel only): as 0 with the k slots of a chunk permuted for
                        //    chain_t.hip: slot s of lane half h <-> k = 16 kc + 8 (s/4) + 4 h + (s%4)
    int row_off;        // output-row offset (lin8: 1)
    int col_off;        // input-col offset
    int out_valid;      // valid output rows (after row_off)
    int in_valid;       // valid input cols (after col_off)
    float scale;
    int li;             // linear index (sdf 0..8, colour 9..13): the two-piece fp16 packs read PACKH.wabs[li]
};

struct PackJobs { PackJob j[44]; int n; };

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ params, float* __restrict__ packed, PackJobs jobs) {
    const PackJob J = jobs.j[blockIdx.y];
    const int64_t total4 = (int64_t)J.nkg * J.nt * 64;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const int lane = i & 63;
        const int nt = (i >> 6) % J.nt;
        const int kg = (i >> 6) / J.nt;
        const int n = nt * 32 + (lane & 31);
        f32x4 v;
        DH_UNROLL for (int s = 0; s < 4; ++s) {
            const int k = kg * 8 + 4 * (lane >> 5) + s;
            const int o = J.rev ? k : n, c = J.rev ? n : k;
            float x = 0.f;
            if (o < J.out_valid && c < J.in_valid) {
                const int row = o + J.row_off;
                x = J.scale * packed[J.rsoff + row] * params[J.voff + (int64_t)row * J.ldv + J.col_off + c];
            }
            v[s] = x;
        }
        reinterpret_cast<f32x4*>(packed + J.dst)[i] = v;
    }
}

// biases, lin8 row 0, colour lin4 (small vectors)
__global__ __launch_bounds__(256) void pack_small_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    const int c = threadIdx.x;   // 0..255
    const int job = blockIdx.x;
    if (job < N_SDF) {
        const int l = job;
        float b = 0.f;
        if (l < 8) { if (c < SDF_DIMS[l].out) b = params[sdf_off(l).bias + c]; }
        else b = params[sdf_off(8).bias + 1 + c];
        packed[PACK.sdf_bias[l] + c] = b;
    } else if (job == N_SDF) {
        const int64_t rs = PACK.rowscale + 8 * 260;
        packed[PACK.sdf_w8row0 + c] = packed[rs + 0] * params[sdf_off(8).v + c];
        if (c == 0) packed[PACK.sdf_b8_0] = params[sdf_off(8).bias];
    } else if (job < N_SDF + 1 + 4) {
        const int l = job - N_SDF - 1;
        packed[PACK.col_bias[l] + c] = params[col_off(l).bias + c];
    } else {
        const int64_t rs = PACK.rowscale + (int64_t)N_SDF * 260 + 4 * 256;
        for (int r = 0; r < 3; ++r) packed[PACK.col_w4 + r * 256 + c] = packed[rs + r] * params[col_off(4).v + r * 256 + c];
        if (c < 3) packed[PACK.col_b4 + c] = params[col_off(4).bias + c];
    }
}

// split-bf16 B operands (tile16.h): same job description as pack_kernel with nkg = number of 16-deep k-chunks
__global__ __launch_bounds__(256) void pack16_kernel(const float* __restrict__ params, float* __restrict__ packed, PackJobs jobs) {
    const PackJob J = jobs.j[blockIdx.y];
    const int64_t total = (int64_t)J.nkg * J.nt * 64;
    bf16x8* dst = reinterpret_cast<bf16x8*>(packed + J.dst);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = i & 63;
        const int nt = (i >> 6) % J.nt;
        const int kc = (i >> 6) / J.nt;
        const int n = nt * 32 + (lane & 31);
        Bf3 w;
        DH_UNROLL for (int s = 0; s < 8; ++s) {
            const int k = kc * 16 + (J.rev == 2 ? 8 * (s >> 2) + 4 * (lane >> 5) + (s & 3) : 8 * (lane >> 5) + s);
            const int o = J.rev == 1 ? k : n, c = J.rev == 1 ? n : k;
            float x = 0.f;
            if (o < J.out_valid && c < J.in_valid) {
                const int row = o + J.row_off;
                x = J.scale * packed[J.rsoff + row] * params[J.voff + (int64_t)row * J.ldv + J.col_off + c];
            }
            __bf16 h1, h2, h3;
            split_f32(x, h1, h2, h3);
            w.p[0][s] = h1; w.p[1][s] = h2; w.p[2][s] = h3;
        }
        DH_UNROLL for (int p = 0; p < 3; ++p) dst[(((int64_t)kc * J.nt + nt) * 3 + p) * 64 + lane] = w.p[p];
    }
}

// two-piece fp16 B operands (tile16h.h): pack16_kernel's geometry, values scaled by the linear's power of two
__global__ __launch_bounds__(256) void packh_kernel(const float* __restrict__ params, float* __restrict__ packed, PackJobs jobs) {
    const PackJob J = jobs.j[blockIdx.y];
    const int64_t total = (int64_t)J.nkg * J.nt * 64;
    u32x4* dst = reinterpret_cast<u32x4*>(packed + J.dst);
    const float sw = wscale_from_bits(reinterpret_cast<const unsigned*>(packed + PACKH.wabs)[J.li]);
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = i & 63;
        const int nt = (i >> 6) % J.nt;
        const int kc = (i >> 6) / J.nt;
        const int n = nt * 32 + (lane & 31);
        float x[8];
        DH_UNROLL for (int s = 0; s < 8; ++s) {
            const int k = kc * 16 + (J.rev == 2 ? 8 * (s >> 2) + 4 * (lane >> 5) + (s & 3) : 8 * (lane >> 5) + s);
            const int o = J.rev == 1 ? k : n, c = J.rev == 1 ? n : k;
            x[s] = 0.f;
            if (o < J.out_valid && c < J.in_valid) {
                const int row = o + J.row_off;
                x[s] = sw * (J.scale * packed[J.rsoff + row] * params[J.voff + (int64_t)row * J.ldv + J.col_off + c]);
            }
        }
        const H2 w = split2(f32x4{x[0], x[1], x[2], x[3]}, f32x4{x[4], x[5], x[6], x[7]});
        DH_UNROLL for (int p = 0; p < 2; ++p) dst[(((int64_t)kc * J.nt + nt) * 2 + p) * 64 + lane] = w.p[p];
    }
}

// chain_t.hip's bias rows, lin8 row 0 and lin8's bias rows 1..256 (layout.h PACKT.bias10)
__global__ __launch_bounds__(256) void packt_small_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    const int c = threadIdx.x, l = blockIdx.x;
    float v;
    if (l < 8) v = c < SDF_DIMS[l].out ? params[sdf_off(l).bias + c] : 0.f;
    else if (l == 8) v = packed[PACK.rowscale + 8 * 260] * params[sdf_off(8).v + c];
    else v = params[sdf_off(8).bias + 1 + c];
    packed[PACKT.bias10 + l * 256 + c] = v;
}

// chain_t.hip's H2 table (layout.h PACKH.bias11): 16 x bias of lin0..lin7, lin8's effective row 0, lin8's bias rows 1..256,
// and 1 / S_w of lin0..lin8
__global__ __launch_bounds__(256) void packth_small_kernel(const float* __restrict__ params, float* __restrict__ packed) {
    const int c = threadIdx.x, l = blockIdx.x;
    float v;
    if (l < 8) v = c < SDF_DIMS[l].out ? H2_XS * params[sdf_off(l).bias + c] : 0.f;
    else if (l == 8) v = packed[PACK.rowscale + 8 * 260] * params[sdf_off(8).v + c];
    else if (l == 9) v = params[sdf_off(8).bias + 1 + c];
    else v = c < N_SDF ? winv_from_bits(reinterpret_cast<const unsigned*>(packed + PACKH.wabs)[c]) : 0.f;
    packed[PACKH.bias11 + l * 256 + c] = v;
}

// the split packs' job list: OFF = Pack16Off (three bf16 pieces) or PackHOff (two fp16 pieces); ts = float offset of the
// register-resident chain's stream, NP = pieces per value
template <class OFF>
static PackJobs build_jobs_split(const OFF& P, int64_t ts, int NP) {
    PackJobs J{};
    int n = 0;
    auto add = [&](int64_t dst, int64_t voff, int64_t rsoff, int ldv, int nkc, int nt, int rev, int row_off, int col_off,
                   int out_valid, int in_valid, float scale, int li) {
        J.j[n++] = PackJob{dst, voff, rsoff, ldv, nkc, nt, rev, row_off, col_off, out_valid, in_valid, scale, li};
    };
    for (int l = 0; l < N_SDF; ++l) {
        const int64_t rs = PACK.rowscale + (int64_t)l * 260;
        const int64_t v = sdf_off(l).v;
        const int in = SDF_DIMS[l].in, out = SDF_DIMS[l].out;
        if (l == 0) {
            add(P.sdf_fwd_aux[0], v, rs, in, 3, 8, 0, 0, 0, out, EMB, 1.f, l);
            add(P.sdf_rev_aux[0], v, rs, in, 16, 2, 1, 0, 0, out, EMB, 1.f, l);
        } else if (l == 4) {
            add(P.sdf_fwd_main[4], v, rs, in, 14, 8, 0, 0, 0, out, SKIP_OUT, INV_SQRT2, l);
            add(P.sdf_fwd_aux[4], v, rs, in, 3, 8, 0, 0, SKIP_OUT, out, EMB, INV_SQRT2, l);
            add(P.sdf_rev_main[4], v, rs, in, 16, 8, 1, 0, 0, out, SKIP_OUT, INV_SQRT2, l);
            add(P.sdf_rev_aux[4], v, rs, in, 16, 2, 1, 0, SKIP_OUT, out, EMB, INV_SQRT2, l);
        } else if (l == 8) {
            add(P.sdf_fwd_main[8], v, rs, in, 16, 8, 0, 1, 0, 256, 256, 1.f, l);
            add(P.sdf_rev_main[8], v, rs, in, 16, 8, 1, 1, 0, 256, 256, 1.f, l);
        } else {
            add(P.sdf_fwd_main[l], v, rs, in, 16, 8, 0, 0, 0, out, in, 1.f, l);
            add(P.sdf_rev_main[l], v, rs, in, 16, 8, 1, 0, 0, out, in, 1.f, l);
        }
    }
    // the register-resident chain's stream: lin0 .. lin7 in stage order, then lin8's rows 1..256
    auto addt = [&](int l, int nstage, int col_off, int in_valid, float scale, int row_off = 0) {
        add(ts, sdf_off(l).v, PACK.rowscale + (int64_t)l * 260, SDF_DIMS[l].in, nstage, 8, 2, row_off, col_off, SDF_DIMS[l].out - row_off, in_valid, scale, l);
        ts += (int64_t)nstage * 8 * NP * 64 * 4;
    };
    addt(0, 3, 0, EMB, 1.f);
    for (int l = 1; l <= 3; ++l) addt(l, 16, 0, 256, 1.f);
    addt(4, 14, 0, SKIP_OUT, INV_SQRT2);
    addt(4, 3, SKIP_OUT, EMB, INV_SQRT2);
    for (int l = 5; l <= 7; ++l) addt(l, 16, 0, 256, 1.f);
    addt(8, 16, 0, 256, 1.f, 1);
    for (int l = 0; l < 4; ++l) {
        const int64_t rs = PACK.rowscale + (int64_t)N_SDF * 260 + (int64_t)l * 256;
        const int64_t v = col_off(l).v;
        const int in = COL_DIMS[l].in;
        if (l == 0) {
            add(P.col_fwd_main[0], v, rs, in, 16, 8, 0, 0, CAUX, 256, 256, 1.f, N_SDF + l);
            add(P.col_fwd_aux0, v, rs, in, 3, 8, 0, 0, 0, 256, CAUX, 1.f, N_SDF + l);
            add(P.col_rev_main[0], v, rs, in, 16, 8, 1, 0, CAUX, 256, 256, 1.f, N_SDF + l);
            add(P.col_rev_aux0, v, rs, in, 16, 2, 1, 0, 0, 256, CAUX, 1.f, N_SDF + l);
        } else {
            add(P.col_fwd_main[l], v, rs, in, 16, 8, 0, 0, 0, 256, 256, 1.f, N_SDF + l);
            add(P.col_rev_main[l], v, rs, in, 16, 8, 1, 0, 0, 256, 256, 1.f, N_SDF + l);
        }
    }
    J.n = n;
    return J;
}

static PackJobs build_jobs() {
    PackJobs J{};
    int n = 0;
    auto add = [&](int64_t dst, int64_t voff, int64_t rsoff, int ldv, int nkg, int nt, int rev, int row_off, int col_off,
                   int out_valid, int in_valid, float scale) {
        J.j[n++] = PackJob{dst, voff, rsoff, ldv, nkg, nt, rev, row_off, col_off, out_valid, in_valid, scale, 0};
    };
    for (int l = 0; l < N_SDF; ++l) {
        const int64_t rs = PACK.rowscale + (int64_t)l * 260;
        const int64_t v = sdf_off(l).v;
        const int in = SDF_DIMS[l].in, out = SDF_DIMS[l].out;
        if (l == 0) {
            add(PACK.sdf_fwd_aux[0], v, rs, in, 5, 8, 0, 0, 0, out, EMB, 1.f);
            add(PACK.sdf_rev_aux[0], v, rs, in, 32, 2, 1, 0, 0, out, EMB, 1.f);
        } else if (l == 4) {
            add(PACK.sdf_fwd_main[4], v, rs, in, 28, 8, 0, 0, 0, out, SKIP_OUT, INV_SQRT2);
            add(PACK.sdf_fwd_aux[4], v, rs, in, 5, 8, 0, 0, SKIP_OUT, out, EMB, INV_SQRT2);
            add(PACK.sdf_rev_main[4], v, rs, in, 32, 8, 1, 0, 0, out, SKIP_OUT, INV_SQRT2);
            add(PACK.sdf_rev_aux[4], v, rs, in, 32, 2, 1, 0, SKIP_OUT, out, EMB, INV_SQRT2);
        } else if (l == 8) {
            add(PACK.sdf_fwd_main[8], v, rs, in, 32, 8, 0, 1, 0, 256, 256, 1.f);
            add(PACK.sdf_rev_main[8], v, rs, in, 32, 8, 1, 1, 0, 256, 256, 1.f);
        } else {
            add(PACK.sdf_fwd_main[l], v, rs, in, 32, 8, 0, 0, 0, out, in, 1.f);
            add(PACK.sdf_rev_main[l], v, rs, in, 32, 8, 1, 0, 0, out, in, 1.f);
        }
    }
    for (int l = 0; l < 4; ++l) {
        const int64_t rs = PACK.rowscale + (int64_t)N_SDF * 260 + (int64_t)l * 256;
        const int64_t v = col_off(l).v;
        const int in = COL_DIMS[l].in;
        if (l == 0) {
            add(PACK.col_fwd_main[0], v, rs, in, 32, 8, 0, 0, CAUX, 256, 256, 1.f);
            add(PACK.col_fwd_aux0, v, rs, in, 5, 8, 0, 0, 0, 256, CAUX, 1.f);
            add(PACK.col_rev_main[0], v, rs, in, 32, 8, 1, 0, CAUX, 256, 256, 1.f);
            add(PACK.col_rev_aux0, v, rs, in, 32, 2, 1, 0, 0, 256, CAUX, 1.f);
        } else {
            add(PACK.col_fwd_main[l], v, rs, in, 32, 8, 0, 0, 0, 256, 256, 1.f);
            add(PACK.col_rev_main[l], v, rs, in, 32, 8, 1, 0, 0, 256, 256, 1.f);
        }
    }
    J.n = n;
    return J;
}

// arith_mask: bit a set = pack the operands of arithmetic a (kernels.h ARITH_*); the row scales, biases and the small fp32 vectors
// every arithmetic reads are always written
int launch_pack_weights(const float* params, float* packed, int arith_mask, hipStream_t stream) {
    (void)hipMemsetAsync(packed + PACKH.wabs, 0, 16 * sizeof(unsigned), stream);
    hipLaunchKernelGGL(rowscale_kernel, dim3(17, N_SDF + N_COL), dim3(256), 0, stream, params, packed);
    hipLaunchKernelGGL(pack_small_kernel, dim3(N_SDF + 1 + 4 + 1), dim3(256), 0, stream, params, packed);
    if (arith_mask & (1 << ARITH_FP32)) {
        static const PackJobs jobs = build_jobs();
        hipLaunchKernelGGL(pack_kernel, dim3(16, jobs.n), dim3(256), 0, stream, params, packed, jobs);
    }
    if (arith_mask & (1 << ARITH_BF16)) {
        static const PackJobs jobs16 = build_jobs_split(PACK16, PACKT.stream, 3);
        hipLaunchKernelGGL(pack16_kernel, dim3(8, jobs16.n), dim3(256), 0, stream, params, packed, jobs16);
        hipLaunchKernelGGL(packt_small_kernel, dim3(10), dim3(256), 0, stream, params, packed);
    }
    if (arith_mask & (1 << ARITH_F16)) {
        static const PackJobs jobsh = build_jobs_split(PACKH, PACKH.stream, 2);
        hipLaunchKernelGGL(packh_kernel, dim3(8, jobsh.n), dim3(256), 0, stream, params, packed, jobsh);
        hipLaunchKernelGGL(packth_small_kernel, dim3(11), dim3(256), 0, stream, params, packed);
    }
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace dh
