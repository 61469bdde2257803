// extern "C" boundary (include/dynhor_hip.h).  Argument checking + launch dispatch only.
#include "../../include/dynhor_hip.h"
#include "kernels.h"
#include "hash_layout.h"
#include "layout.h"
#include "workspace.h"

using namespace dh;

namespace dh {
// the library's only process-global state (include/dynhor_hip.h "Conventions")
static int g_arith = DH_ARITH_SPLIT_F16;
static int g_hash_scatter = 0;
int hash_scatter_mode() { return __atomic_load_n(&g_hash_scatter, __ATOMIC_RELAXED); }
}  // namespace dh

namespace {
inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
inline bool bad_arith(int a) { return a != DH_ARITH_SPLIT_BF16 && a != DH_ARITH_FP32_MFMA && a != DH_ARITH_SPLIT_F16; }
// stages with a tile-PAIR form (chain_pair.hip) accept DH_CHAIN_FORM_* flags above the arithmetic byte, with DH_ARITH_SPLIT_F16 only
inline bool bad_arith_form(int a) {
    const int form = a & ~0xff;
    return bad_arith(a & 0xff) || (form != 0 && form != DH_CHAIN_FORM_TILE && form != DH_CHAIN_FORM_PAIR) ||
           (form != 0 && (a & 0xff) != DH_ARITH_SPLIT_F16);
}
inline int cur_arith() { return __atomic_load_n(&dh::g_arith, __ATOMIC_RELAXED); }
#ifndef DH_GRID_DIV
#define DH_GRID_DIV 1                    // development macro: 2 = ONE workgroup per CU (what a chain's phases cost without a co-resident partner)
#endif
constexpr int DEFAULT_GRID = 256 * 2 / DH_GRID_DIV;   // persistent workgroups: all that are co-resident (two 64-point tiles per CU)
}  // namespace

extern "C" {

int dh_version(void) { return 3; }

int dh_set_arithmetic(int mode) {
    if (bad_arith(mode)) return DH_ERR_BAD_ARG;
    __atomic_store_n(&dh::g_arith, mode, __ATOMIC_RELAXED);
    return DH_OK;
}
int dh_get_arithmetic(void) { return __atomic_load_n(&dh::g_arith, __ATOMIC_RELAXED); }

int dh_hash_set_scatter_mode(int mode) {
    if (mode < 0 || mode > 2) return DH_ERR_BAD_ARG;
    __atomic_store_n(&dh::g_hash_scatter, mode, __ATOMIC_RELAXED);
    return DH_OK;
}

const char* dh_strerror(int status) {
    switch (status) {
        case DH_OK: return "ok";
        case DH_ERR_BAD_ARG: return "bad argument (null / negative size / misaligned pointer)";
        case DH_ERR_UNSUPPORTED: return "unsupported configuration";
        case DH_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
        default: return "unknown dynhor_hip status";
    }
}

int64_t dh_num_params(void) { return N_PARAMS; }
int64_t dh_packed_floats(void) { return PACKH.total; }

int dh_packed_section(int section, int64_t* offset_floats, int64_t* n_floats) {
    if (!offset_floats || !n_floats) return DH_ERR_BAD_ARG;
    switch (section) {
        case 0: *offset_floats = PACKT.stream; *n_floats = PACKT_STREAM_FLOATS; break;
        case 1: *offset_floats = PACKT.bias10; *n_floats = 10 * 256; break;
        case 2: *offset_floats = PACKH.stream; *n_floats = PACKTH_STREAM_FLOATS; break;
        case 3: *offset_floats = PACKH.bias11; *n_floats = 11 * 256; break;
        case 4: *offset_floats = PACKH.wabs; *n_floats = 16; break;
        default: return DH_ERR_BAD_ARG;
    }
    return DH_OK;
}

int dh_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim) {
    if (!bias_off || !g_off || !v_off || !out_dim || !in_dim) return DH_ERR_BAD_ARG;
    if (net == 0) {
        if (layer < 0 || layer >= N_SDF) return DH_ERR_BAD_ARG;
        const LinOff o = sdf_off(layer);
        *bias_off = o.bias; *g_off = o.g; *v_off = o.v; *out_dim = SDF_DIMS[layer].out; *in_dim = SDF_DIMS[layer].in;
    } else if (net == 1) {
        *bias_off = *g_off = *v_off = VARIANCE_OFF; *out_dim = 1; *in_dim = 1;
    } else if (net == 2) {
        if (layer < 0 || layer >= N_COL) return DH_ERR_BAD_ARG;
        const LinOff o = col_off(layer);
        *bias_off = o.bias; *g_off = o.g; *v_off = o.v; *out_dim = COL_DIMS[layer].out; *in_dim = COL_DIMS[layer].in;
    } else {
        return DH_ERR_BAD_ARG;
    }
    return DH_OK;
}

int dh_pack_weights(const float* params, float* packed, void* stream) {
    if (!params || !packed || misaligned16(packed)) return DH_ERR_BAD_ARG;
    return launch_pack_weights(params, packed, 7, static_cast<hipStream_t>(stream));
}
int dh_pack_weights_ex(int arithmetic, const float* params, float* packed, void* stream) {
    if (!params || !packed || misaligned16(packed) || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    return launch_pack_weights(params, packed, 1 << arithmetic, static_cast<hipStream_t>(stream));
}

int dh_sdf_nograd_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* sdf, void* stream) {
    if (npts < 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (npts == 0) return DH_OK;
    if (!packed || !pts || !sdf || misaligned16(packed)) return DH_ERR_BAD_ARG;
    return launch_sdf_nograd(packed, pts, npts, sdf, DEFAULT_GRID, arithmetic, static_cast<hipStream_t>(stream));
}

int dh_workspace_floats(int64_t npts, int64_t* infer_floats, int64_t* fwd_floats, int64_t* total_floats) {
    if (npts < 0 || !infer_floats || !fwd_floats || !total_floats) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(nullptr, npts);
    *infer_floats = w.infer_floats;
    *fwd_floats = w.fwd_floats;
    *total_floats = w.total_floats;
    return DH_OK;
}

int dh_range_words(int64_t* act_max_off, int64_t* tag_off, float* limit) {
    if (!act_max_off || !tag_off || !limit) return DH_ERR_BAD_ARG;
    // (absmax is the first block of the workspace: carve_workspace)
    *act_max_off = (int64_t)ABSMAX_ACT * ABSMAX_STRIDE;
    *tag_off = (int64_t)ABSMAX_TAG * ABSMAX_STRIDE;
    *limit = H2_RANGE / H2_XS;
    return DH_OK;
}

int dh_sdf_forward_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* ws, float* sdf, void* stream) {
    if (bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (npts <= 0) return npts == 0 ? DH_OK : DH_ERR_BAD_ARG;
    if (!packed || !pts || !ws || !sdf || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_sdf_fwd_train(packed, pts, npts, sdf, w.feat, w.act, w.eaux, w.absmax, DEFAULT_GRID, arithmetic, static_cast<hipStream_t>(stream));
}

int dh_sdf_gradient_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* ws, float* normals, int save,
                       void* stream) {
    if (bad_arith_form(arithmetic)) return DH_ERR_BAD_ARG;
    if (npts <= 0) return npts == 0 ? DH_OK : DH_ERR_BAD_ARG;
    if (!packed || !pts || !ws || !normals || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    if (save < 0 || save > 2) return DH_ERR_BAD_ARG;
    return launch_sdf_grad(packed, pts, npts, w.act, w.asave, normals, save, w.gesave, w.absmax, DEFAULT_GRID, arithmetic,
                           static_cast<hipStream_t>(stream));
}

int dh_color_forward_ex(int arithmetic, const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                        int64_t npts, float* ws, float* color, int save, void* stream) {
    if (npts < 0 || n_per_ray <= 0 || bad_arith_form(arithmetic)) return DH_ERR_BAD_ARG;
    if (npts == 0) return DH_OK;
    if (!packed || !pts || !dirs || !normals || !ws || !color || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_color_fwd(packed, pts, dirs, n_per_ray, normals, w.feat, npts, color, w.cact, w.caux, save, w.absmax, DEFAULT_GRID,
                            arithmetic, static_cast<hipStream_t>(stream));
}

int dh_mlp_forward_ex(int arithmetic, const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                      float* sdf, float* normals, float* color, void* stream) {
    int rc = dh_sdf_forward_ex(arithmetic, packed, pts, npts, ws, sdf, stream);
    if (rc) return rc;
    rc = dh_sdf_gradient_ex(arithmetic, packed, pts, npts, ws, normals, 1, stream);
    if (rc) return rc;
    return dh_color_forward_ex(arithmetic, packed, pts, dirs, n_per_ray, normals, npts, ws, color, 1, stream);
}

int dh_color_backward_ex(int arithmetic, const float* packed, const float* colors, const float* d_colors, int64_t npts, float* ws,
                         float* d_normals, void* stream) {
    if (npts <= 0 || bad_arith_form(arithmetic)) return DH_ERR_BAD_ARG;
    if (!packed || !colors || !d_colors || !ws || !d_normals || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_color_bwd(packed, colors, d_colors, npts, w.cact, w.czbar, w.featbar, d_normals, w.tpart, w.absmax, DEFAULT_GRID,
                            arithmetic, static_cast<hipStream_t>(stream));
}

int dh_sdf_tangent_ex(int arithmetic, const float* packed, const float* pts, const float* d_normals, int64_t npts, float* ws,
                      void* stream) {
    if (npts <= 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (!packed || !pts || !d_normals || !ws || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_sdf_tangent(packed, pts, d_normals, npts, w.act, w.asave, w.t0aux, w.tsave, w.rsave, w.tpart, w.absmax, DEFAULT_GRID,
                              arithmetic, static_cast<hipStream_t>(stream));
}

int dh_sdf_backward_ex(int arithmetic, const float* packed, const float* d_sdf, int64_t npts, float* ws, void* stream) {
    if (npts <= 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (!packed || !d_sdf || !ws || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_sdf_bwd(packed, d_sdf, npts, w.act, w.rsave, w.featbar, w.zbar, w.tpart, w.absmax, DEFAULT_GRID, arithmetic,
                          static_cast<hipStream_t>(stream));
}

int dh_color_backward_rays_ex(int arithmetic, const float* packed, const float* colors, const float* d_colors, const float* dirs,
                              int n_per_ray, int64_t npts, float* ws, float* d_normals, float* d_pts, float* d_dirs_pts, void* stream) {
    if (npts <= 0 || n_per_ray <= 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (!packed || !colors || !d_colors || !dirs || !ws || !d_normals || !d_pts || !d_dirs_pts || misaligned16(packed) || misaligned16(ws))
        return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_color_bwd_rays(packed, colors, d_colors, dirs, n_per_ray, npts, w.cact, w.czbar, w.featbar, d_normals, w.tpart,
                                 d_pts, d_dirs_pts, w.absmax, DEFAULT_GRID, arithmetic, static_cast<hipStream_t>(stream));
}

int dh_sdf_backward_rays_ex(int arithmetic, const float* packed, const float* d_sdf, const float* pts, const float* d_normals,
                            int64_t npts, float* ws, float* d_pts, void* stream) {
    if (npts <= 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (!packed || !d_sdf || !pts || !d_normals || !ws || !d_pts || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_sdf_bwd_rays(packed, d_sdf, pts, d_normals, npts, w.act, w.rsave, w.featbar, w.gesave, w.zbar, w.tpart, d_pts,
                               w.absmax, DEFAULT_GRID, arithmetic, static_cast<hipStream_t>(stream));
}

int dh_weight_grads_gemm_ex(int arithmetic, int64_t npts, float* ws, void* stream) {
    if (npts <= 0 || bad_arith(arithmetic)) return DH_ERR_BAD_ARG;
    if (!ws || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_weight_grads_gemm(w, w.slabs, DW_G, arithmetic, static_cast<hipStream_t>(stream));
}

int dh_weight_grads_fold(const float* packed, const float* params, int64_t npts, float* ws, float* grad_flat, void* stream) {
    if (npts <= 0) return DH_ERR_BAD_ARG;
    if (!packed || !params || !ws || !grad_flat || misaligned16(packed) || misaligned16(ws)) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(ws, npts);
    return launch_weight_grads_fold(w, w.slabs, w.tred, DW_G, DW_NS, params, packed, grad_flat, static_cast<hipStream_t>(stream));
}

int dh_mlp_backward_ex(int arithmetic, const float* packed, const float* params, const float* pts, int64_t npts, float* ws,
                       const float* colors, const float* d_sdf, float* d_normals, const float* d_colors, float* grad_flat,
                       void* stream) {
    int rc = dh_color_backward_ex(arithmetic, packed, colors, d_colors, npts, ws, d_normals, stream);
    if (rc) return rc;
    rc = dh_sdf_tangent_ex(arithmetic, packed, pts, d_normals, npts, ws, stream);
    if (rc) return rc;
    rc = dh_sdf_backward_ex(arithmetic, packed, d_sdf, npts, ws, stream);
    if (rc) return rc;
    rc = dh_weight_grads_gemm_ex(arithmetic, npts, ws, stream);
    if (rc) return rc;
    return dh_weight_grads_fold(packed, params, npts, ws, grad_flat, stream);
}

// the entry points without an arithmetic argument: the process default (dh_set_arithmetic)
int dh_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, void* stream) {
    return dh_sdf_nograd_ex(cur_arith(), packed, pts, npts, sdf, stream);
}
int dh_sdf_forward(const float* packed, const float* pts, int64_t npts, float* ws, float* sdf, void* stream) {
    return dh_sdf_forward_ex(cur_arith(), packed, pts, npts, ws, sdf, stream);
}
int dh_sdf_gradient(const float* packed, const float* pts, int64_t npts, float* ws, float* normals, int save, void* stream) {
    return dh_sdf_gradient_ex(cur_arith(), packed, pts, npts, ws, normals, save, stream);
}
int dh_color_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                     int64_t npts, float* ws, float* color, int save, void* stream) {
    return dh_color_forward_ex(cur_arith(), packed, pts, dirs, n_per_ray, normals, npts, ws, color, save, stream);
}
int dh_mlp_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                   float* sdf, float* normals, float* color, void* stream) {
    return dh_mlp_forward_ex(cur_arith(), packed, pts, dirs, n_per_ray, npts, ws, sdf, normals, color, stream);
}
int dh_color_backward(const float* packed, const float* colors, const float* d_colors, int64_t npts, float* ws,
                      float* d_normals, void* stream) {
    return dh_color_backward_ex(cur_arith(), packed, colors, d_colors, npts, ws, d_normals, stream);
}
int dh_sdf_tangent(const float* packed, const float* pts, const float* d_normals, int64_t npts, float* ws, void* stream) {
    return dh_sdf_tangent_ex(cur_arith(), packed, pts, d_normals, npts, ws, stream);
}
int dh_sdf_backward(const float* packed, const float* d_sdf, int64_t npts, float* ws, void* stream) {
    return dh_sdf_backward_ex(cur_arith(), packed, d_sdf, npts, ws, stream);
}
int dh_color_backward_rays(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                           int64_t npts, float* ws, float* d_normals, float* d_pts, float* d_dirs_pts, void* stream) {
    return dh_color_backward_rays_ex(cur_arith(), packed, colors, d_colors, dirs, n_per_ray, npts, ws, d_normals, d_pts, d_dirs_pts, stream);
}
int dh_sdf_backward_rays(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                         float* ws, float* d_pts, void* stream) {
    return dh_sdf_backward_rays_ex(cur_arith(), packed, d_sdf, pts, d_normals, npts, ws, d_pts, stream);
}
int dh_weight_grads_gemm(int64_t npts, float* ws, void* stream) { return dh_weight_grads_gemm_ex(cur_arith(), npts, ws, stream); }
int dh_mlp_backward(const float* packed, const float* params, const float* pts, int64_t npts, float* ws,
                    const float* colors, const float* d_sdf, float* d_normals, const float* d_colors, float* grad_flat,
                    void* stream) {
    return dh_mlp_backward_ex(cur_arith(), packed, params, pts, npts, ws, colors, d_sdf, d_normals, d_colors, grad_flat, stream);
}

int dh_gen_rays(const uint8_t* rgb, const int8_t* label, const uint8_t* normal, const float* R, const float* T,
                const float* Kinv, int H, int W, int n_frames, int frame, const int64_t* px, const int64_t* py, int64_t B,
                float* rays, float* near, float* far, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || frame < 0 || frame >= n_frames) return DH_ERR_BAD_ARG;
    if (B == 0) return DH_OK;
    if (!rgb || !label || !normal || !R || !T || !Kinv || !px || !py || !rays || !near || !far) return DH_ERR_BAD_ARG;
    return launch_gen_rays(rgb, label, normal, R, T, Kinv, H, W, frame, px, py, B, rays, near, far, static_cast<hipStream_t>(stream));
}

int dh_coarse_samples(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* t_rand,
                      int64_t B, int n_samples, float* z, float* pts, void* stream) {
    if (B < 0 || n_samples <= 0) return DH_ERR_BAD_ARG;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !near || !far || !z || !pts) return DH_ERR_BAD_ARG;
    return launch_coarse_samples(rays_o, rays_d, near, far, t_rand, B, n_samples, z, pts, static_cast<hipStream_t>(stream));
}

int dh_upsample_step(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t B, int n_cur,
                     int n_new, float inv_s, float* z_new, float* pts_new, void* stream) {
    if (B < 0 || n_cur < 2 || n_new <= 0) return DH_ERR_BAD_ARG;
    if (n_cur > 128 || n_new > 64) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !z || !sdf || !z_new || !pts_new) return DH_ERR_BAD_ARG;
    return launch_upsample(rays_o, rays_d, z, sdf, B, n_cur, n_new, inv_s, z_new, pts_new, static_cast<hipStream_t>(stream));
}

int dh_merge_samples(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int64_t B, int n_cur,
                     int n_new, float* z_out, float* sdf_out, void* stream) {
    if (B < 0 || n_cur <= 0 || n_new <= 0) return DH_ERR_BAD_ARG;
    if (n_cur > 128 || n_new > 64) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!z || !z_new || !z_out) return DH_ERR_BAD_ARG;
    if (sdf_out && (!sdf || !sdf_new)) return DH_ERR_BAD_ARG;
    return launch_merge(z, z_new, sdf, sdf_new, B, n_cur, n_new, z_out, sdf_out, static_cast<hipStream_t>(stream));
}

int dh_midpoints(const float* rays_o, const float* rays_d, const float* z, int64_t B, int n, float sample_dist, float* pts,
                 void* stream) {
    if (B < 0 || n <= 0) return DH_ERR_BAD_ARG;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !z || !pts) return DH_ERR_BAD_ARG;
    return launch_midpoints(rays_o, rays_d, z, B, n, sample_dist, pts, static_cast<hipStream_t>(stream));
}

int dh_render_scan_fwd(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                       const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                       const float* background_rgb, int64_t B, int n, float* weights, float* color, float* weight_sum,
                       float* weight_max, float* cdf, float* inside_sphere, float* eik_partial, float* normal_map,
                       void* stream) {
    if (B < 0 || n <= 0) return DH_ERR_BAD_ARG;
    if (n > 128) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !z || !sdf || !normals || !colors || !inv_s || !weights || !color || !weight_sum ||
        !weight_max || !cdf || !inside_sphere || !eik_partial) return DH_ERR_BAD_ARG;
    return launch_render_fwd(rays_o, rays_d, z, sdf, normals, colors, inv_s, cos_anneal_ratio, sample_dist, background_rgb,
                             B, n, weights, color, weight_sum, weight_max, cdf, inside_sphere, eik_partial, normal_map,
                             nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int dh_render_scan_bwd(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                       const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                       const float* background_rgb, int64_t B, int n, const float* d_color, const float* d_weight_sum,
                       const float* d_weights, const float* d_gradients, const float* d_normal_map, const float* eik_coef,
                       float* d_sdf, float* d_normals, float* d_colors, float* d_inv_s, void* stream) {
    if (B < 0 || n <= 0) return DH_ERR_BAD_ARG;
    if (n > 128) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !z || !sdf || !normals || !colors || !inv_s || !d_color || !eik_coef || !d_sdf ||
        !d_normals || !d_colors || !d_inv_s) return DH_ERR_BAD_ARG;
    return launch_render_bwd(rays_o, rays_d, z, sdf, normals, colors, inv_s, cos_anneal_ratio, sample_dist, background_rgb,
                             B, n, d_color, d_weight_sum, d_weights, d_gradients, d_normal_map, eik_coef, d_sdf, d_normals, d_colors,
                             d_inv_s, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int dh_render_scan_bwd_rays(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                            const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                            const float* background_rgb, int64_t B, int n, const float* d_color, const float* d_weight_sum,
                            const float* d_weights, const float* d_gradients, const float* d_normal_map, const float* eik_coef,
                            float* d_sdf, float* d_normals, float* d_colors, float* d_inv_s, float* d_rays_d, void* stream) {
    if (B < 0 || n <= 0) return DH_ERR_BAD_ARG;
    if (n > 128) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !z || !sdf || !normals || !colors || !inv_s || !d_color || !eik_coef || !d_sdf ||
        !d_normals || !d_colors || !d_inv_s || !d_rays_d) return DH_ERR_BAD_ARG;
    return launch_render_bwd(rays_o, rays_d, z, sdf, normals, colors, inv_s, cos_anneal_ratio, sample_dist, background_rgb,
                             B, n, d_color, d_weight_sum, d_weights, d_gradients, d_normal_map, eik_coef, d_sdf, d_normals, d_colors,
                             d_inv_s, d_rays_d, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int dh_march_count(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* u,
                   const uint8_t* occupancy, int res, float radius, float step, float half_step, int max_samples, int64_t B,
                   int32_t* cnt, void* stream) {
    if (B < 0 || res <= 0 || !(radius > 0.f) || !(step > 0.f) || max_samples <= 0) return DH_ERR_BAD_ARG;
    if (max_samples > 1024) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !near || !far || !occupancy || !cnt) return DH_ERR_BAD_ARG;
    return launch_march_count(rays_o, rays_d, near, far, u, occupancy, res, radius, step, half_step, max_samples, B, cnt,
                              static_cast<hipStream_t>(stream));
}

int dh_march_emit(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* u,
                  const uint8_t* occupancy, int res, float radius, float step, float half_step, int max_samples, int64_t B,
                  const int64_t* off, const int32_t* keep, float* t_start, float* pts, float* dirs_pts, int32_t* ray_idx,
                  void* stream) {
    if (B < 0 || res <= 0 || !(radius > 0.f) || !(step > 0.f) || max_samples <= 0) return DH_ERR_BAD_ARG;
    if (max_samples > 1024) return DH_ERR_UNSUPPORTED;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !near || !far || !occupancy || !off || !t_start || !pts || !dirs_pts || !ray_idx) return DH_ERR_BAD_ARG;
    return launch_march_emit(rays_o, rays_d, near, far, u, occupancy, res, radius, step, half_step, max_samples, B, off, keep,
                             t_start, pts, dirs_pts, ray_idx, static_cast<hipStream_t>(stream));
}

int dh_render_scan_fwd_packed(const float* rays_o, const float* rays_d, const float* t_start, const float* sdf, const float* normals,
                              const float* colors, const float* inv_s, float cos_anneal_ratio, float step,
                              const float* background_rgb, int64_t B, const int64_t* seg_off, const int32_t* seg_cnt,
                              float* weights, float* color, float* weight_sum, float* weight_max, float* cdf,
                              float* inside_sphere, float* eik_partial, float* normal_map, void* stream) {
    if (B < 0) return DH_ERR_BAD_ARG;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !t_start || !sdf || !normals || !colors || !inv_s || !seg_off || !seg_cnt || !weights || !color ||
        !weight_sum || !weight_max || !cdf || !inside_sphere || !eik_partial) return DH_ERR_BAD_ARG;
    return launch_render_fwd(rays_o, rays_d, t_start, sdf, normals, colors, inv_s, cos_anneal_ratio, step, background_rgb, B, 0,
                             weights, color, weight_sum, weight_max, cdf, inside_sphere, eik_partial, normal_map, seg_off, seg_cnt,
                             static_cast<hipStream_t>(stream));
}

int dh_render_scan_bwd_packed(const float* rays_o, const float* rays_d, const float* t_start, const float* sdf, const float* normals,
                              const float* colors, const float* inv_s, float cos_anneal_ratio, float step,
                              const float* background_rgb, int64_t B, const int64_t* seg_off, const int32_t* seg_cnt,
                              const float* d_color, const float* d_weight_sum, const float* d_weights, const float* d_gradients,
                              const float* d_normal_map, const float* eik_coef, float* d_sdf, float* d_normals, float* d_colors,
                              float* d_inv_s, void* stream) {
    if (B < 0) return DH_ERR_BAD_ARG;
    if (B == 0) return DH_OK;
    if (!rays_o || !rays_d || !t_start || !sdf || !normals || !colors || !inv_s || !seg_off || !seg_cnt || !d_color || !eik_coef ||
        !d_sdf || !d_normals || !d_colors || !d_inv_s) return DH_ERR_BAD_ARG;
    return launch_render_bwd(rays_o, rays_d, t_start, sdf, normals, colors, inv_s, cos_anneal_ratio, step, background_rgb, B, 0,
                             d_color, d_weight_sum, d_weights, d_gradients, d_normal_map, eik_coef, d_sdf, d_normals, d_colors,
                             d_inv_s, nullptr, seg_off, seg_cnt, static_cast<hipStream_t>(stream));
}

int dh_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, float grad_scale, void* stream) {
    if (n < 0 || step < 1) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!params || !grad || !exp_avg || !exp_avg_sq) return DH_ERR_BAD_ARG;
    return launch_adam(params, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, grad_scale,
                       static_cast<hipStream_t>(stream));
}

int dh_neus_loss(const float* color, const float* weight_sum, const float* normal_map, const float* eik_partial,
                 const float* rays, const float* R, int64_t B, float igr_weight, float mask_weight, float normal_weight,
                 float* stats, float* d_color, float* d_weight_sum, float* d_normal_map, float* eik_coef, void* stream) {
    if (B <= 0) return DH_ERR_BAD_ARG;
    if (!color || !weight_sum || !eik_partial || !rays || !stats || !d_color || !d_weight_sum || !eik_coef)
        return DH_ERR_BAD_ARG;
    if (normal_weight > 0.f && (!normal_map || !R || !d_normal_map)) return DH_ERR_BAD_ARG;
    return launch_loss(color, weight_sum, normal_map, eik_partial, rays, R, B, igr_weight, mask_weight, normal_weight, stats,
                       d_color, d_weight_sum, d_normal_map, eik_coef, static_cast<hipStream_t>(stream));
}

int dh_corr_loss(const float* rays_o, const float* rays_d, const float* z, const float* weights, const float* corr,
                 const float* R_all, const float* T_all, int n_frames, const float* K, int64_t B, int n, float sample_dist,
                 float delta_px, float corr_weight, float* stats, float* residual_px, float* d_weights, float* pose_adjoints,
                 void* stream) {
    if (B <= 0 || n <= 0 || n_frames <= 0 || !(delta_px > 0.f)) return DH_ERR_BAD_ARG;
    if (!rays_o || !rays_d || !z || !weights || !corr || !R_all || !T_all || !K || !stats || !residual_px || !d_weights)
        return DH_ERR_BAD_ARG;
    return launch_corr_loss(rays_o, rays_d, z, weights, corr, R_all, T_all, n_frames, K, B, n, sample_dist, delta_px, corr_weight,
                            stats, residual_px, d_weights, pose_adjoints, static_cast<hipStream_t>(stream));
}

int64_t dh_hashgrid_entries(void) { return hashgrid_entries(); }

int dh_hashgrid_level(int level, float* scale, uint32_t* resolution, uint32_t* offset, uint32_t* dense) {
    if (!scale || !resolution || !offset || !dense) return DH_ERR_BAD_ARG;
    return hashgrid_level(level, scale, resolution, offset, dense) ? DH_ERR_BAD_ARG : DH_OK;
}

int dh_hashgrid_encode(const float* table, const float* x01, int64_t n, float* out, void* stream) {
    if (n < 0) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!table || !x01 || !out || (reinterpret_cast<uintptr_t>(table) & 7u) || (reinterpret_cast<uintptr_t>(out) & 7u))
        return DH_ERR_BAD_ARG;
    return launch_hashgrid_fwd(table, x01, n, out, static_cast<hipStream_t>(stream));
}

int dh_hashgrid_encode_backward(const float* x01, const float* d_out, int64_t n, float* d_table, void* stream) {
    if (n < 0) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!x01 || !d_out || !d_table || (reinterpret_cast<uintptr_t>(d_out) & 7u)) return DH_ERR_BAD_ARG;
    return launch_hashgrid_bwd(x01, d_out, n, d_table, static_cast<hipStream_t>(stream));
}

int64_t dh_hash_num_params(void) { return hash_num_params(); }
int64_t dh_hash_packed_floats(void) { return HP_TOTAL; }

int dh_hash_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim) {
    if (!bias_off || !g_off || !v_off || !out_dim || !in_dim) return DH_ERR_BAD_ARG;
    const HashParamOff P = make_hash_param_off(hashgrid_entries());
    if (net == 0 && layer == 0) { *bias_off = P.g0_b; *g_off = P.g0_g; *v_off = P.g0_v; *out_dim = 64; *in_dim = HM_GIN; }
    else if (net == 0 && layer == 1) { *bias_off = P.g1_b; *g_off = P.g1_g; *v_off = P.g1_v; *out_dim = HM_GOUT; *in_dim = 64; }
    else if (net == 1 && layer == 0) { *bias_off = *g_off = *v_off = P.variance; *out_dim = 1; *in_dim = 1; }
    else if (net == 2 && layer == 0) { *bias_off = P.c0_b; *g_off = P.c0_g; *v_off = P.c0_v; *out_dim = 64; *in_dim = HM_CIN; }
    else if (net == 2 && layer == 1) { *bias_off = P.c1_b; *g_off = P.c1_g; *v_off = P.c1_v; *out_dim = 64; *in_dim = 64; }
    else if (net == 2 && layer == 2) { *bias_off = P.c2_b; *g_off = P.c2_g; *v_off = P.c2_v; *out_dim = 3; *in_dim = 64; }
    else if (net == 3 && layer == 0) { *bias_off = *g_off = *v_off = P.table; *out_dim = (int)hashgrid_entries(); *in_dim = 2; }
    else return DH_ERR_BAD_ARG;
    return DH_OK;
}

int dh_hash_pack_weights(const float* params, float* packed, void* stream) {
    if (!params || !packed) return DH_ERR_BAD_ARG;
    return launch_hash_pack(params, packed, static_cast<hipStream_t>(stream));
}

int dh_hash_workspace_floats(int64_t npts, int64_t* infer_floats, int64_t* total_floats) {
    if (npts < 0 || !infer_floats || !total_floats) return DH_ERR_BAD_ARG;
    *infer_floats = hash_infer_workspace_floats(npts);
    *total_floats = hash_workspace_floats(npts);
    return DH_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int dh_hash_sdf_nograd(const float* params, const float* packed, const float* pts, int64_t n, float radius, float* sdf,
                       void* stream) {
    if (n < 0 || !(radius > 0.f)) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!params || !packed || !pts || !sdf || !al16(params) || !al16(packed)) return DH_ERR_BAD_ARG;
    return launch_hash_sdf_nograd(params, packed, pts, n, radius, sdf, static_cast<hipStream_t>(stream));
}

int dh_hash_geo_forward(const float* params, const float* packed, const float* pts, int64_t n, float radius, float eps,
                        float* ws, int save, float* sdf, float* feature, float* gradient, const int64_t* n_active, void* stream) {
    if (n < 0 || !(radius > 0.f) || !(eps > 0.f) || (n_active && n % 8)) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!params || !packed || !pts || !ws || !sdf || !feature || !gradient || !al16(params) || !al16(packed) || !al16(ws))
        return DH_ERR_BAD_ARG;
    return launch_hash_geo_fwd(params, packed, pts, n, radius, eps, ws, save, sdf, feature, gradient, n_active,
                               static_cast<hipStream_t>(stream));
}

int dh_hash_color_forward(const float* packed, const float* feature, const float* normals, const float* dirs,
                          int n_per_ray, int64_t n, float* color, const int64_t* n_active, void* stream) {
    if (n < 0 || n_per_ray <= 0 || n % n_per_ray) return DH_ERR_BAD_ARG;
    if (n == 0) return DH_OK;
    if (!packed || !feature || !normals || !dirs || !color || !al16(packed)) return DH_ERR_BAD_ARG;
    return launch_sh_color_fwd(packed, feature, normals, dirs, n_per_ray, n, color, n_active, static_cast<hipStream_t>(stream));
}

int dh_hash_color_backward(const float* packed, const float* feature, const float* normals, const float* dirs,
                           const float* d_color, int n_per_ray, int64_t n, float* ws, float* d_feature, float* d_normals,
                           const int64_t* n_active, void* stream) {
    if (n <= 0 || n_per_ray <= 0 || n % n_per_ray || (n_active && n % 8)) return DH_ERR_BAD_ARG;
    if (!packed || !feature || !normals || !dirs || !d_color || !ws || !d_feature || !d_normals || !al16(packed) || !al16(ws))
        return DH_ERR_BAD_ARG;
    return launch_sh_color_bwd(packed, feature, normals, dirs, d_color, n_per_ray, n, ws, d_feature, d_normals, n_active,
                               static_cast<hipStream_t>(stream));
}

int dh_hash_geo_backward(const float* params, const float* packed, const float* pts, const float* d_sdf,
                         const float* d_feature, const float* d_normals, int64_t n, float radius, float eps, float* ws,
                         const int64_t* n_active, void* stream) {
    if (n <= 0 || !(radius > 0.f) || !(eps > 0.f) || (n_active && n % 8)) return DH_ERR_BAD_ARG;
    if (!params || !packed || !pts || !d_sdf || !d_feature || !d_normals || !ws || !al16(params) || !al16(packed) || !al16(ws))
        return DH_ERR_BAD_ARG;
    return launch_hash_geo_bwd(params, packed, pts, d_sdf, d_feature, d_normals, n, radius, eps, ws, n_active,
                               static_cast<hipStream_t>(stream));
}

int dh_hash_weight_grads(const float* params, const float* packed, int64_t n, float* ws, float* grad, const int64_t* n_active,
                         void* stream) {
    if (n <= 0 || !params || !packed || !ws || !grad || !al16(ws) || (n_active && n % 8)) return DH_ERR_BAD_ARG;
    if (hash_scatter_mode() != 0) return DH_ERR_BAD_ARG;        // the merge ablations exist for the float-atomic form only (header)
    return launch_hash_weight_grads(params, packed, n, ws, grad, n_active, 7, static_cast<hipStream_t>(stream));
}

int dh_hash_weight_grads_parts(const float* params, const float* packed, int64_t n, float* ws, float* grad, const int64_t* n_active,
                               int parts, void* stream) {
    if (n <= 0 || !params || !packed || !ws || !grad || !al16(ws) || (n_active && n % 8) || parts < 1 || parts > 7 || parts == 4 || parts == 6) return DH_ERR_BAD_ARG;
    if ((parts & 4) && hash_scatter_mode() != 0) return DH_ERR_BAD_ARG;      // (as above)
    return launch_hash_weight_grads(params, packed, n, ws, grad, n_active, parts, static_cast<hipStream_t>(stream));
}

}  // extern "C"
