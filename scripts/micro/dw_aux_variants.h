// Development variants of the weight-gradient kernel's aux body (dw.hip dw_body_aux_h) -- the irreproducibility hunt of round 5
// (DESIGN.md section 4 "Reproducibility", profiles/r05_dw_aux_hazard_table*.json).  NOT product code: until round 6 these switches
// lived inside csrc/dw.hip; the product file now holds the shipping body only and this header holds the body WITH its switches
// (-DDW_AUX_V_SCALAR_MUL, _PK_OPSEL_HI_DWORD, _PK_OPSEL_LO_DWORD, _PK_PLAIN, _VOLATILE_PIECES, _NO_MFMA, _LGKM_AFTER_A, _NOP_AFTER_A,
// _LGKM_BEFORE_LOAD, -DDW_AUX_THREE_PIECE) and a kernel identical to dw_f16x2_kernel that calls it.  Several variants are wrong by
// construction (that is what they are for).  Include AFTER csrc/dw.hip.
#pragma once
namespace dh {

// The aux jobs (64-wide B operand: jobs 0, 8, 11) in the two-piece arithmetic: wave w owns output rows [32w, 32w+32) x 64 columns
// (two accumulators), waves 0 / 1 publish the two B tiles.  Two raw register sets alternate with the two piece buffers.
__device__ __forceinline__ void dw_body_aux_v(const DwJob& J, const DwScales& sc, int64_t t0, int64_t t1, float* __restrict__ out,
                                              int wave, int lane, char* lds) {
    constexpr int KQ = MT * 4;
    f32x16 acc[2];
    DH_UNROLL for (int j = 0; j < 2; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int T = (int)(t1 - t0);
    const int npairs = J.A2 ? 2 : 1;
    const int NP1 = T * (KQ / 2);
    const int NP = npairs * NP1;
    const bool has_b = wave < 2;
    struct Raw { f32x4 a0, a1, b0, b1; float sa, sb; };
    int ld_pair = 0, ld_kp = 0;
    int64_t ld_tile = t0;
    auto load = [&](Raw& r) {
        const float* A = ld_pair ? J.A2 : J.A1;
        const float* Bm = ld_pair ? J.B2 : J.B1;
        {
            const unsigned sxb = pow2_scale_bits((ld_pair ? sc.xt1 : sc.xt0)[ld_tile], H2_AT);
            const float sx = __builtin_bit_cast(float, sxb), sy = (ld_pair ? sc.prod1 : sc.prod0) * __builtin_bit_cast(float, pow2_inv_bits(sxb));
            const bool ha = ld_pair ? sc.heavy_a1 : sc.heavy_a0;
            r.sa = ha ? sx : sy;
            r.sb = ha ? sy : sx;
        }
        const int kq = 2 * ld_kp, m = kq >> 2, r4 = kq & 3;
        const f32x4* ga = reinterpret_cast<const f32x4*>(A + ld_tile * TILE_F) + ((((wave >> 1) * MT + m) * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
        r.a0 = __builtin_nontemporal_load(ga); r.a1 = __builtin_nontemporal_load(ga + 64);
        if (has_b) {
            const f32x4* gb = reinterpret_cast<const f32x4*>(Bm + ld_tile * AUXT_F) + ((m * 2 + (wave & 1)) * 4 + r4) * 64 + lane;
            r.b0 = __builtin_nontemporal_load(gb); r.b1 = __builtin_nontemporal_load(gb + 64);
        }
        if (++ld_kp == KQ / 2) {
            ld_kp = 0;
            if (++ld_tile == t1) { ld_tile = t0; if (++ld_pair == npairs) ld_pair = 0; }
        }
    };
    auto piece = [&](int par, int tile) {
        H2 f;
        const char* base = lds + par * DWH_BUF + tile * DWH_TILE + lane * 16;
#ifdef DW_AUX_V_VOLATILE_PIECES
        DH_UNROLL for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const volatile u32x4*>(base + p * 1024);
#else
        DH_UNROLL for (int p = 0; p < 2; ++p) f.p[p] = *reinterpret_cast<const u32x4*>(base + p * 1024);
#endif
        return f;
    };
    auto scaled_split = [&](const f32x4& x0, const f32x4& x1, float s) {
#if defined(DW_AUX_V_SCALAR_MUL)
        // the scale applied by eight single v_mul_f32 (inline asm: the compiler cannot pair them into v_pk_mul_f32)
        f32x4 m0, m1;
        DH_UNROLL for (int i = 0; i < 4; ++i) {
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0[i]) : "v"(x0[i]), "v"(s));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1[i]) : "v"(x1[i]), "v"(s));
        }
        return split2(m0, m1);
#elif defined(DW_AUX_V_PK_OPSEL_HI_DWORD) || defined(DW_AUX_V_PK_OPSEL_LO_DWORD)
        // packed multiplies that broadcast the scale out of ONE dword of an aligned pair through op_sel, the other dword holding a
        // constant: HI_DWORD = the form hipcc generates in the failing body (op_sel:[0,1]: both results read src1's high dword),
        // LO_DWORD = its mirror (op_sel_hi:[1,0]: both results read src1's low dword)
        f32x2 ss;
#ifdef DW_AUX_V_PK_OPSEL_HI_DWORD
        ss[0] = 0.f; ss[1] = s;
#else
        ss[0] = s; ss[1] = 0.f;
#endif
        asm volatile("" : "+v"(ss));
        f32x4 m0, m1;
        DH_UNROLL for (int i = 0; i < 2; ++i) {
            f32x2 a, b, ra, rb;
            a[0] = x0[2 * i]; a[1] = x0[2 * i + 1]; b[0] = x1[2 * i]; b[1] = x1[2 * i + 1];
#ifdef DW_AUX_V_PK_OPSEL_HI_DWORD
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(ra) : "v"(a), "v"(ss));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(rb) : "v"(b), "v"(ss));
#else
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(ra) : "v"(a), "v"(ss));
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(rb) : "v"(b), "v"(ss));
#endif
            m0[2 * i] = ra[0]; m0[2 * i + 1] = ra[1]; m1[2 * i] = rb[0]; m1[2 * i + 1] = rb[1];
        }
        return split2(m0, m1);
#elif defined(DW_AUX_V_PK_PLAIN)
        // packed multiplies by a scale held in BOTH halves of an aligned pair: no op_sel
        f32x2 ss; ss[0] = s; ss[1] = s;
        asm volatile("" : "+v"(ss));
        f32x4 m0, m1;
        DH_UNROLL for (int i = 0; i < 2; ++i) {
            f32x2 a, b, ra, rb;
            a[0] = x0[2 * i]; a[1] = x0[2 * i + 1]; b[0] = x1[2 * i]; b[1] = x1[2 * i + 1];
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(ra) : "v"(a), "v"(ss));
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(rb) : "v"(b), "v"(ss));
            m0[2 * i] = ra[0]; m0[2 * i + 1] = ra[1]; m1[2 * i] = rb[0]; m1[2 * i + 1] = rb[1];
        }
        return split2(m0, m1);
#else
        return split2(x0 * s, x1 * s);
#endif
    };
    auto publish_a = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        const H2 pa = scaled_split(r.a0, r.a1, r.sa);
        DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + wave * DWH_TILE + p * 1024) = pa.p[p];
    };
    auto publish_b = [&](const Raw& r, int par) {
        char* base = lds + par * DWH_BUF + lane * 16;
        if (has_b) {
            const H2 pb = scaled_split(r.b0, r.b1, r.sb);
            DH_UNROLL for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(base + (8 + (wave & 1)) * DWH_TILE + p * 1024) = pb.p[p];
        }
    };
    if (NP > 0) {
        Raw r0, r1;
        load(r0);
        load(r1);
        __builtin_amdgcn_sched_barrier(0);
        publish_a(r0, 0);
        publish_b(r0, 0);
        load(r0);
        __syncthreads();
        auto step = [&](int par, Raw& nxt) {
            const H2 a = piece(par, wave), b0 = piece(par, 8), b1 = piece(par, 9);
            __builtin_amdgcn_sched_barrier(0);
            publish_a(nxt, par ^ 1);
#ifdef DW_AUX_V_NO_MFMA
            // no matrix instruction in the step: the "accumulators" take a cheap function of the same pieces on the vector ALU
            DH_UNROLL for (int r = 0; r < 16; ++r)
                acc[0][r] += __builtin_bit_cast(float, ((a.p[r & 1][(r >> 1) & 3] ^ b0.p[(r >> 3) & 1][(r >> 1) & 3]) & 0x007fffffu) | 0x3f800000u);
#else
            acc[0] = mfma3(a, b0, acc[0]);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if defined(DW_AUX_V_LGKM_AFTER_A)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(DW_AUX_V_NOP_AFTER_A)
            asm volatile("s_nop 15\n s_nop 15" ::: "memory");
#endif
            publish_b(nxt, par ^ 1);
#ifdef DW_AUX_V_NO_MFMA
            DH_UNROLL for (int r = 0; r < 16; ++r)
                acc[1][r] += __builtin_bit_cast(float, ((a.p[r & 1][(r >> 1) & 3] ^ b1.p[(r >> 3) & 1][(r >> 1) & 3]) & 0x007fffffu) | 0x3f800000u);
#else
            acc[1] = mfma3(a, b1, acc[1]);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if defined(DW_AUX_V_LGKM_BEFORE_LOAD)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            load(nxt);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        };
        const float ratio = npairs == 2 ? sc.prod1 / sc.prod0 : 1.f;
        auto rescale = [&](int p) {
            if (npairs == 2 && p == NP1) {
                DH_UNROLL for (int j = 0; j < 2; ++j) DH_UNROLL for (int r = 0; r < 16; ++r) acc[j][r] *= ratio;
            }
        };
        int p = 0;
        for (; p + 1 < NP; p += 2) {
            rescale(p); step(0, r1);
            rescale(p + 1); step(1, r0);
        }
        if (p < NP) { rescale(p); step(0, r1); }
    }
    const float inv = sc.poison / (npairs == 2 ? sc.prod1 : sc.prod0);
    DH_UNROLL for (int j = 0; j < 2; ++j) {
        float* o = out + ((int64_t)wave * 2 + j) * 1024 + lane;
        DH_UNROLL for (int r = 0; r < 16; ++r) o[r * 64] = acc[j][r] * inv;
    }
}


__global__ __launch_bounds__(512, 1) void dw_f16x2_variant_kernel(DwJobs jobs, DwGroups groups, int64_t ntiles, float* __restrict__ slabs,
                                                                  int64_t gstride, const unsigned* __restrict__ absmax,
                                                                  const unsigned* __restrict__ tmax) {
    __shared__ __attribute__((aligned(16))) char pieces[2 * (DWP_BUF > DWH_BUF ? DWP_BUF : DWH_BUF)];
    const int g = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int k = dw_group_of(groups, g), gl = g - groups.wg0[k], gc = groups.wg0[k + 1] - groups.wg0[k];
    const int64_t t0 = ntiles * gl / gc, t1 = ntiles * (gl + 1) / gc;
    float* base = slabs + (int64_t)g * gstride;
    for (int job = groups.job0[k]; job < groups.job0[k + 1]; ++job) {
        const DwJob J = jobs.j[job];
        const DwScales sc = dw_job_scales(J, absmax, tmax, ntiles);
        if (J.nb == 8) dw_body_pieces_h<8>(J, sc, t0, t1, base + J.off, wave, lane, pieces);
#ifdef DW_AUX_THREE_PIECE          // round 4's shipping form: the aux jobs on the three-piece bf16 body
        else dw_body_pieces<2>(J, t0, t1, base + J.off, wave, lane, pieces);
#else
        else dw_body_aux_v(J, sc, t0, t1, base + J.off, wave, lane, pieces);
#endif
    }
}

}  // namespace dh
