// Internal launch-function declarations (one per kernel family).  All enqueue on the caller's stream,
// allocate nothing, and return 0 / negative dh error codes (include/dynhor_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "layout.h"

namespace dh {

// arithmetic of the MLP GEMMs (include/dynhor_hip.h dh_arithmetic): passed to every launch function; the entry points without an
// arithmetic argument pass the process default (dh_set_arithmetic)
enum : int { ARITH_BF16 = 0, ARITH_FP32 = 1, ARITH_F16 = 2 };
int hash_scatter_mode();

int launch_pack_weights(const float* params, float* packed, int arith_mask, hipStream_t stream);

// MLP chains (kernels_mlp.hip).  npts is padded by the caller to a multiple of 128 for saved buffers.
int launch_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, int grid, int arith, hipStream_t stream);
int launch_sdf_nograd_t(const float* packed, const float* pts, int64_t npts, float* sdf, bool h2, hipStream_t stream);   // chain_t.hip
int launch_sdf_fwd_train_t(const float* packed, const float* pts, int64_t npts, float* sdf, float* feat, float* act, float* eaux,
                           unsigned* absmax, bool h2, hipStream_t stream);

// (absmax: workspace.h -- the two-piece fp16 kernels post the per-launch maxima of their saved-tile classes there)
int launch_sdf_fwd_train(const float* packed, const float* pts, int64_t npts, float* sdf, float* feat, float* act,
                         float* eaux, float* absmax, int grid, int arith, hipStream_t stream);
int launch_sdf_grad(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals,
                    int save, float* gesave, float* absmax, int grid, int arith, hipStream_t stream);
int launch_color_fwd(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                     const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, float* absmax, int grid,
                     int arith, hipStream_t stream);
// kernels_mlp_h.hip (two-piece fp16)
int launch_sdf_grad_h(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals,
                      int save, float* gesave, unsigned* absmax, int grid, hipStream_t stream);
int launch_color_fwd_h(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                       const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, unsigned* absmax,
                       int grid, hipStream_t stream);
int launch_color_bwd_h(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                       int64_t npts, const float* cact, float* czbar, float* featbar, float* d_normals, float* tpart,
                       float* d_pts, float* d_dirs_pts, unsigned* absmax, unsigned* tmax, int grid, hipStream_t st);
int launch_sdf_tangent_h(const float* packed, const float* pts, const float* d_normals, int64_t npts, const float* act,
                         const float* asave, float* t0aux, float* tsave, float* rsave, float* tpart, unsigned* absmax, unsigned* tmax,
                         int grid, hipStream_t st);
int launch_sdf_bwd_h(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                     const float* act, const float* rsave, const float* featbar, const float* gesave, float* zbar, float* tpart,
                     float* d_pts, unsigned* absmax, unsigned* tmax, int grid, hipStream_t st);

// chain_pair.hip (two-piece fp16, tile-PAIR form: one workgroup per CU, weights held in registers across two tiles; round 6).
// The launchers above pick it for launches of at least 2 x #CUs tiles unless the arithmetic word carries a form flag (CHAIN_FORM_*).
enum : int { CHAIN_FORM_AUTO = 0, CHAIN_FORM_TILE = 1, CHAIN_FORM_PAIR = 2 };
bool pair_form_available();
int pair_form_cus();
// the form a SPLIT_F16 stage launch of npts points takes: forced by the flag; otherwise PAIR where the stage's pair form is the faster one
// on the bench's launch (auto_pair: measured same-process A/Bs, profiles/r06_ab_chain_forms.json -- colour forward yes; input gradient and
// colour backward no: their epilogues load a saved tile, and with one wave per SIMD nothing hides that latency) and every CU gets a pair
inline bool use_pair_form(int form, int64_t npts, bool auto_pair) {
    if (form == CHAIN_FORM_PAIR) return true;
    if (form == CHAIN_FORM_TILE || !auto_pair) return false;
    const int cus = pair_form_cus();
    return cus > 0 && (npts + TM - 1) / TM >= 2 * (int64_t)cus;
}
int launch_color_fwd_p(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                       const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, unsigned* absmax,
                       hipStream_t stream);

int launch_sdf_grad_p(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals, int save,
                      float* gesave, unsigned* absmax, hipStream_t stream);

int launch_color_bwd_p(const float* packed, const float* colors, const float* d_colors, int64_t npts, const float* cact, float* czbar,
                       float* featbar, float* d_normals, float* tpart, unsigned* absmax, unsigned* tmax, hipStream_t stream);

// per-ray kernels (kernels_ray.hip)
int launch_gen_rays(const uint8_t* rgb, const int8_t* label, const uint8_t* normal, const float* R, const float* T,
                    const float* Kinv, int H, int W, int frame, const int64_t* px, const int64_t* py, int64_t B,
                    float* rays, float* near, float* far, hipStream_t st);
int launch_coarse_samples(const float* o, const float* d, const float* near, const float* far, const float* t_rand,
                          int64_t B, int n, float* z, float* pts, hipStream_t st);
int launch_upsample(const float* o, const float* d, const float* z, const float* sdf, int64_t B, int n, int n_new,
                    float inv_s, float* z_new, float* pts_new, hipStream_t st);
int launch_merge(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int64_t B, int n, int n_new,
                 float* z_out, float* sdf_out, hipStream_t st);
int launch_midpoints(const float* o, const float* d, const float* z, int64_t B, int n, float sample_dist, float* pts,
                     hipStream_t st);
int launch_render_fwd(const float* o, const float* d, const float* z, const float* sdf, const float* normals,
                      const float* colors, const float* inv_s, float car, float sample_dist, const float* bg, int64_t B,
                      int n, float* weights, float* color, float* wsum, float* wmax, float* cdf, float* inside, float* eik,
                      float* nmap, const int64_t* seg_off, const int32_t* seg_cnt, hipStream_t st);
int launch_render_bwd(const float* o, const float* d, const float* z, const float* sdf, const float* normals,
                      const float* colors, const float* inv_s, float car, float sample_dist, const float* bg, int64_t B,
                      int n, const float* d_color, const float* d_wsum, const float* d_weights, const float* d_gradients,
                      const float* d_nmap, const float* eik_coef, float* d_sdf, float* d_normals, float* d_colors,
                      float* d_inv_s, float* d_rays_d, const int64_t* seg_off, const int32_t* seg_cnt, hipStream_t st);
// occupancy-grid marching (march.hip)
int launch_march_count(const float* o, const float* d, const float* near, const float* far, const float* u, const uint8_t* occ,
                       int res, float radius, float step, float half_step, int max_samples, int64_t B, int32_t* cnt,
                       hipStream_t st);
int launch_march_emit(const float* o, const float* d, const float* near, const float* far, const float* u, const uint8_t* occ,
                      int res, float radius, float step, float half_step, int max_samples, int64_t B, const int64_t* off,
                      const int32_t* keep, float* t_start, float* pts, float* dirs_pts, int32_t* ray_idx, hipStream_t st);
int launch_loss(const float* color, const float* wsum, const float* nmap, const float* eik, const float* rays,
                const float* R, int64_t B, float igr_w, float mask_w, float normal_w, float* stats, float* d_color,
                float* d_wsum, float* d_nmap, float* eik_coef, hipStream_t st);

int launch_corr_loss(const float* rays_o, const float* rays_d, const float* z, const float* weights, const float* corr,
                     const float* R_all, const float* T_all, int n_frames, const float* K, int64_t B, int n, float sample_dist,
                     float delta_px, float corr_w, float* stats, float* residual_px, float* d_weights, float* pose_adj,
                     hipStream_t st);

// backward chains (kernels_mlp_bwd.hip) and weight gradients (dw.hip)
int launch_color_bwd(const float* packed, const float* colors, const float* d_colors, int64_t npts, const float* cact,
                     float* czbar, float* featbar, float* d_normals, float* tpart, float* absmax, int grid, int arith, hipStream_t st);
// pose-refinement variants (split-bf16 arithmetic only): additionally the adjoints w.r.t. the sample points / view directions
int launch_color_bwd_rays(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                          int64_t npts, const float* cact, float* czbar, float* featbar, float* d_normals, float* tpart,
                          float* d_pts, float* d_dirs_pts, float* absmax, int grid, int arith, hipStream_t st);
int launch_sdf_bwd_rays(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                        const float* act, const float* rsave, const float* featbar, const float* gesave, float* zbar,
                        float* tpart, float* d_pts, float* absmax, int grid, int arith, hipStream_t st);
int launch_sdf_tangent(const float* packed, const float* pts, const float* d_normals, int64_t npts, const float* act,
                       const float* asave, float* t0aux, float* tsave, float* rsave, float* tpart, float* absmax, int grid, int arith,
                       hipStream_t st);
int launch_sdf_bwd(const float* packed, const float* d_sdf, int64_t npts, const float* act, const float* rsave,
                   const float* featbar, float* zbar, float* tpart, float* absmax, int grid, int arith, hipStream_t st);
struct Workspace;
int launch_weight_grads_gemm(const Workspace& w, float* slabs, int G, int arith, hipStream_t st);
int launch_weight_grads_fold(const Workspace& w, float* slabs, float* tred, int G, int nS, const float* params,
                             const float* packed, float* grad, hipStream_t st);

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int64_t step, float grad_scale, hipStream_t st);

// multiresolution hash-grid encoding (hashgrid.hip)
int64_t hashgrid_entries();
int hashgrid_level(int l, float* scale, uint32_t* res, uint32_t* offset, uint32_t* dense);
int launch_hashgrid_fwd(const float* table, const float* x01, int64_t n, float* out, hipStream_t st);
int launch_hashgrid_bwd(const float* x01, const float* d_out, int64_t n, float* d_table, hipStream_t st);

// hash-grid model family (hash_mlp.hip): fused encoding + small MLPs, forward and backward
int64_t hash_num_params();
int64_t hash_workspace_floats(int64_t n);
int launch_hash_pack(const float* params, float* hp, hipStream_t st);
int launch_hash_sdf_nograd(const float* params, const float* hp, const float* pts, int64_t n, float radius, float* sdf,
                           hipStream_t st);
int64_t hash_infer_workspace_floats(int64_t n);
int launch_hash_geo_fwd(const float* params, const float* hp, const float* pts, int64_t n, float radius, float eps,
                        float* ws, int save, float* sdf, float* feat, float* grad, const int64_t* n_act, hipStream_t st);
int launch_sh_color_fwd(const float* hp, const float* feat, const float* normals, const float* dirs, int n_per_ray,
                        int64_t n, float* color, const int64_t* n_act, hipStream_t st);
int launch_sh_color_bwd(const float* hp, const float* feat, const float* normals, const float* dirs, const float* d_color,
                        int n_per_ray, int64_t n, float* ws, float* d_feat, float* d_normals, const int64_t* n_act, hipStream_t st);
int launch_hash_geo_bwd(const float* params, const float* hp, const float* pts, const float* d_sdf, const float* d_feat,
                        const float* d_grad, int64_t n, float radius, float eps, float* ws, const int64_t* n_act, hipStream_t st);
int launch_hash_weight_grads(const float* params, const float* hp, int64_t n, float* ws, float* grad, const int64_t* n_act,
                             int parts, hipStream_t st);

}  // namespace dh
