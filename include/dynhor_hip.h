/* dynhor_hip.h -- C ABI of libdynhor_hip.so: the MI355X (gfx950) implementation of the NeuS reconstruction
 * hot path named by BASELINE.json.north_star.
 *
 * The reference (EAST-J/Dynhor @ 2025-09-05) exposes NO plugin/FFI surface for this path -- the path itself is
 * unreleased (reference README.md:7-11, 55-58).  Each entry point therefore cites the
 * upstream-NeuS Python method it implements and the reference file:line that constrains
 * its inputs, and INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes; every pointer is a DEVICE pointer unless named host_*; fp32 row-major.
 *   - the caller allocates everything (torch tensors); the library never allocates, frees or retains pointers.
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, no implicit synchronisation.
 *   - return 0 on success or a negative dh_status; never throws, never exits.  Re-entrant: the library keeps no
 *     pointer and no per-call state.  Its ONLY process-global state is the two DEFAULT words set by dh_set_arithmetic() and
 *     dh_hash_set_scatter_mode() below (plain ints; no environment variable is read anywhere).  Every MLP stage also has
 *     an `_ex` entry point that takes the arithmetic as its first argument and reads no global at all: two host threads
 *     (or two renderers) can run different arithmetics side by side through those.
 *   - arithmetic: every buffer that crosses this boundary is fp32.  Inside, the GEMMs form each fp32 product from
 *     low-precision pieces on the matrix cores with fp32 accumulation, to fp32 accuracy:
 *     DH_ARITH_SPLIT_F16 (default since round 4) two fp16 pieces per operand and three MFMA products, the operands
 *     scaled by powers of two; DH_ARITH_SPLIT_BF16 three bf16 pieces and six products.
 */
#ifndef DYNHOR_HIP_H
#define DYNHOR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    DH_OK = 0,
    DH_ERR_BAD_ARG = -1,        /* null pointer / negative size / misaligned buffer */
    DH_ERR_UNSUPPORTED = -2,    /* configuration outside the fixed NeuS architecture */
    DH_ERR_LAUNCH = -3          /* hipGetLastError() != hipSuccess after enqueue */
} dh_status;

int dh_version(void);
const char* dh_strerror(int status);

/* Arithmetic of the MLP GEMMs.  DH_ARITH_SPLIT_F16 (default): two-piece fp16 split, three products (csrc/tile16h.h).
 * DH_ARITH_SPLIT_BF16: three-piece bf16 split, six products (csrc/tile16.h; the default of rounds 1-3).
 * DH_ARITH_FP32_MFMA: the native v_mfma_f32_32x32x2_f32 twin of every kernel (the independent, exact-fp32 arithmetic that
 * tests/test_gpu_arithmetic_modes.py checks the other two against).  All three read and write the same buffers (packed
 * weights -- dh_pack_weights writes every layout --, workspace, outputs).  dh_set_arithmetic sets the DEFAULT used by the
 * entry points that take no arithmetic argument (for launches enqueued after the call); the `_ex` entry points below
 * ignore it.
 * ONE ARITHMETIC PER STEP: every stage of one training step on one workspace, dh_sdf_forward through dh_weight_grads_gemm, must
 * run in the SAME arithmetic.  The buffers are shared, but the workspace's scale tables (absmax / tmax, csrc/workspace.h) are
 * cleared by dh_sdf_forward and filled by the SPLIT_F16 stages only: a SPLIT_F16 dh_weight_grads_gemm behind a
 * bf16 / fp32 backward (or behind a dh_set_arithmetic between the un-suffixed stage calls of one step) would read stale scale
 * words.  Guard: every training forward clears the table and only the SPLIT_F16 one then writes a tag word into it
 * (csrc/workspace.h ABSMAX_TAG); a SPLIT_F16 dh_weight_grads_gemm that finds no tag writes NaN gradients (loud) instead of
 * gradients scaled by stale words (silently wrong). */
typedef enum { DH_ARITH_SPLIT_BF16 = 0, DH_ARITH_FP32_MFMA = 1, DH_ARITH_SPLIT_F16 = 2 } dh_arithmetic;
/* Chain FORM of a DH_ARITH_SPLIT_F16 stage (round 6; csrc/pair16h.h).  The stages listed below exist in two kernel forms that write
 * bit-identical results: the TILE form (csrc/kernels_mlp_h.hip: one 64-point tile per workgroup, two workgroups per CU, weights
 * streamed from L2 per tile) and the PAIR form (csrc/chain_pair.hip: one workgroup per CU owns two tiles, holds a layer's weight slice
 * in registers across both and runs one tile's epilogue under the other's MFMAs: half the L2 -> CU weight bytes per point).  OR one of
 * these flags into the `arithmetic` argument of the stage's `_ex` entry point to force a form (tests, A/Bs); without a flag the stage
 * takes the form that measured faster on the bench's launch: dh_color_forward_ex the PAIR form for launches of at least 2 x #CUs tiles
 * (32,768 points on MI355X; a pair workgroup occupies a whole CU), dh_sdf_gradient_ex and dh_color_backward_ex the TILE form (their pair
 * forms are 5-20 % slower: DESIGN.md section 3).  The PAIR form returns DH_ERR_UNSUPPORTED on a device without 160 KB of LDS per CU.
 * Stages with a PAIR form: dh_sdf_gradient_ex, dh_color_forward_ex, dh_color_backward_ex (not its pose-refinement form
 * dh_color_backward_rays_ex).  Every other entry point rejects the flags (DH_ERR_BAD_ARG). */
enum { DH_CHAIN_FORM_TILE = 0x100, DH_CHAIN_FORM_PAIR = 0x200 };
int dh_set_arithmetic(int mode);
int dh_get_arithmetic(void);

/* ---- parameter vector / packed weights -----------------------------------------------------------------
 * One flat fp32 vector of dh_num_params() == 802,491 values in state_dict order (SURVEY.md §5 checkpoint row:
 * lin{l}.bias, lin{l}.weight_g, lin{l}.weight_v for sdf_network_fine lin0..8, variance, color_network_fine
 * lin0..4).  dh_param_layout: net 0 = SDFNetwork, 1 = SingleVarianceNetwork (layer ignored; only v_off), 2 =
 * RenderingNetwork. */
int64_t dh_num_params(void);
int64_t dh_packed_floats(void);
int dh_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim);

/* Where a section of the packed buffer lies (float offset, float count) -- for tests and tools that decode it: 0 = the
 * register-resident chains' bf16x3 weight stream (132 stages of 24 KB), 1 = their 10 bias rows, 2 = the two-piece fp16 stream
 * (132 stages of 16 KB), 3 = its 11-row table (16 x bias of lin0..7, lin8 row 0, lin8 bias rows 1..256, 1 / S_w of lin0..8),
 * 4 = max |W| per linear (16 u32: sdf lin0..8, colour lin0..4), from which the fp16 weight scales are derived. */
int dh_packed_section(int section, int64_t* offset_floats, int64_t* n_floats);

/* weight-norm (W = g v/||v||, upstream nn.utils.weight_norm) + MFMA-operand packing; once per optimiser step. */
int dh_pack_weights(const float* params, float* packed, void* stream);

/* SDFNetwork.sdf(pts) under no_grad (upstream NeuSRenderer.render up-sampling loop).
 * pts [npts,3] -> sdf [npts]. */
int dh_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, void* stream);

/* Workspace size (floats) for a render call over npts fine sample points: infer_floats suffices for a forward-only
 * render (save = 0 below), fwd_floats is what a training forward writes (saved activations), total_floats additionally
 * covers the backward pass. */
int dh_workspace_floats(int64_t npts, int64_t* infer_floats, int64_t* fwd_floats, int64_t* total_floats);

/* Range watch of the DH_ARITH_SPLIT_F16 arithmetic.  Its register-resident SDF forward chain carries the softplus activations at
 * the CONSTANT scale 16 in fp16 (csrc/chain_t.hip): an activation beyond *limit = 65504 / 16 = 4094 overflows the hi piece and the
 * results downstream of it are NaN (every other operand class of the MLPs is scaled dynamically, per tile or per launch, and has no
 * such limit; the embedding input shares the constant scale but is a point of the unit sphere).  dh_sdf_gradient(_ex) -- which reads
 * every activation tile the forward saved -- therefore posts the largest activation of the launch into the workspace: one fp32 word
 * at float offset *act_max_off of ws, valid once dh_sdf_gradient of the step has run.  A caller that can afford a device read (the
 * Runner: at report iterations) compares it with *limit and switches to DH_ARITH_SPLIT_BF16 / raises.  *tag_off: the word that holds
 * 0x00F16F16 after a SPLIT_F16 forward (see "ONE ARITHMETIC PER STEP" above).  The no-grad chain has no workspace: beyond the limit
 * its outputs are NaN (never finite garbage). */
int dh_range_words(int64_t* act_max_off, int64_t* tag_off, float* limit);

/* The MLP part of upstream NeuSRenderer.render_core (App. A.7) on npts points (point i belongs to ray
 * i / n_per_ray): sdf_network(pts) -> sdf [npts], feature (kept in ws); sdf_network.gradient(pts) -> normals
 * [npts,3]; color_network(pts, normals, dirs, feature) -> color [npts,3] (post-sigmoid).  Saves what
 * dh_mlp_backward needs into ws. */
int dh_mlp_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                   float* sdf, float* normals, float* color, void* stream);

/* The three stages of dh_mlp_forward as separate single-kernel launches (same ws); save = 0 skips the stores that only
 * the backward pass needs (forward-only rendering: validate_image). */
int dh_sdf_forward(const float* packed, const float* pts, int64_t npts, float* ws, float* sdf, void* stream);
int dh_sdf_gradient(const float* packed, const float* pts, int64_t npts, float* ws, float* normals, int save, void* stream);
int dh_color_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                     int64_t npts, float* ws, float* color, int save, void* stream);

/* Adjoint of dh_mlp_forward (autograd of upstream render_core's network calls, incl. the second-order path through
 * sdf_network.gradient's create_graph=True): given d_sdf [npts], d_normals [npts,3] (updated in place with the colour
 * network's contribution) and d_colors [npts,3] (wrt the post-sigmoid colour), writes the gradient of every network
 * parameter (weight-norm folded: bias, weight_g, weight_v) into grad_flat [dh_num_params()] (the variance entry is
 * left untouched).  ws must be the workspace dh_mlp_forward filled; packed/params the weights it used. */
int dh_mlp_backward(const float* packed, const float* params, const float* pts, int64_t npts, float* ws,
                    const float* colors, const float* d_sdf, float* d_normals, const float* d_colors, float* grad_flat,
                    void* stream);

/* The four stages of dh_mlp_backward as separate launches (same ws, in this order): colour-network backward
 * (adds its d/d normal into d_normals), the forward-mode tangent chain of the second-order path, the SDF backward
 * chain, the split-K weight-gradient GEMMs (one kernel), and their reduction + weight-norm fold into grad_flat. */
int dh_color_backward(const float* packed, const float* colors, const float* d_colors, int64_t npts, float* ws,
                      float* d_normals, void* stream);
int dh_sdf_tangent(const float* packed, const float* pts, const float* d_normals, int64_t npts, float* ws, void* stream);
int dh_sdf_backward(const float* packed, const float* d_sdf, int64_t npts, float* ws, void* stream);
/* Pose refinement (per-frame object poses as trainable cameras; reference precedent for the 6-D
 * rotation + translation parameters and their optimiser: ObjTracker/utils/geometry.py:7-25, jointopt.py:125-141).  The same two
 * stages as above, additionally producing d loss / d sample point (d_pts [npts,3]) and, for the colour network's view embedding,
 * d loss / d ray direction per point (d_dirs_pts [npts,3]).  Call order: dh_sdf_gradient(save = 2) in the forward (keeps the
 * embedding-gradient vector in ws), then dh_color_backward_rays (WRITES d_pts, d_dirs_pts) -> dh_sdf_tangent ->
 * dh_sdf_backward_rays (ACCUMULATES onto d_pts, incl. the second-order path) -> weight-gradient stages as usual.  The caller
 * reduces per ray: d_rays_o = sum_k d_pts, d_rays_d = sum_k (mid_k d_pts + d_dirs_pts) + dh_render_scan_bwd_rays' d_rays_d;
 * sample depths are treated as constants.  Both arithmetic modes (round 3: the fp32-MFMA twins have the same variants). */
int dh_color_backward_rays(const float* packed, const float* colors, const float* d_colors, const float* dirs, int n_per_ray,
                           int64_t npts, float* ws, float* d_normals, float* d_pts, float* d_dirs_pts, void* stream);
int dh_sdf_backward_rays(const float* packed, const float* d_sdf, const float* pts, const float* d_normals, int64_t npts,
                         float* ws, float* d_pts, void* stream);
int dh_weight_grads_gemm(int64_t npts, float* ws, void* stream);
int dh_weight_grads_fold(const float* packed, const float* params, int64_t npts, float* ws, float* grad_flat, void* stream);

/* The same MLP stages with the arithmetic passed explicitly (a dh_arithmetic value; DH_ERR_BAD_ARG otherwise): no launch
 * depends on process-global state (SURVEY.md section 8b "re-entrant; no global mutable state").  Arguments after the first
 * are those of the entry point of the same name above.  (dh_pack_weights and dh_weight_grads_fold do not depend on the
 * arithmetic.) */
int dh_sdf_nograd_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* sdf, void* stream);
int dh_mlp_forward_ex(int arithmetic, const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                      float* sdf, float* normals, float* color, void* stream);
int dh_sdf_forward_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* ws, float* sdf, void* stream);
int dh_sdf_gradient_ex(int arithmetic, const float* packed, const float* pts, int64_t npts, float* ws, float* normals, int save,
                       void* stream);
int dh_color_forward_ex(int arithmetic, const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                        int64_t npts, float* ws, float* color, int save, void* stream);
int dh_mlp_backward_ex(int arithmetic, const float* packed, const float* params, const float* pts, int64_t npts, float* ws,
                       const float* colors, const float* d_sdf, float* d_normals, const float* d_colors, float* grad_flat,
                       void* stream);
int dh_color_backward_ex(int arithmetic, const float* packed, const float* colors, const float* d_colors, int64_t npts, float* ws,
                         float* d_normals, void* stream);
int dh_sdf_tangent_ex(int arithmetic, const float* packed, const float* pts, const float* d_normals, int64_t npts, float* ws,
                      void* stream);
int dh_sdf_backward_ex(int arithmetic, const float* packed, const float* d_sdf, int64_t npts, float* ws, void* stream);
int dh_color_backward_rays_ex(int arithmetic, const float* packed, const float* colors, const float* d_colors, const float* dirs,
                              int n_per_ray, int64_t npts, float* ws, float* d_normals, float* d_pts, float* d_dirs_pts, void* stream);
int dh_sdf_backward_rays_ex(int arithmetic, const float* packed, const float* d_sdf, const float* pts, const float* d_normals,
                            int64_t npts, float* ws, float* d_pts, void* stream);
int dh_weight_grads_gemm_ex(int arithmetic, int64_t npts, float* ws, void* stream);
/* dh_pack_weights for ONE arithmetic: the row scales, biases and small fp32 vectors every arithmetic reads plus that arithmetic's
 * MFMA operands only (a training step then packs one operand set instead of three). */
int dh_pack_weights_ex(int arithmetic, const float* params, float* packed, void* stream);

/* ---- per-ray stages ---------------------------------------------------------------------------------------
 * Mask-conditioned ray generation = upstream Dataset.gen_random_rays_at + near_far_from_sphere (App. A.8) under the
 * reference's hand-off conventions: K per ObjTracker/run.py:119-123 (Kinv = its inverse, row-major [9]); pose
 * x_cam = R x_obj + T per run.py:166 / vis.py:52 (R [F,9] row-major, T [F,3]); label map 1 object / 0 background /
 * -1 hand per run.py:66 and utils/maskutils.py:24-28; obj = label>0, keep = label>=0 per pose_initializtion.py:60-61.
 * Frames stay resident in HBM: rgb u8 [F,H,W,3], label i8 [F,H,W], normal u8 [F,H,W,3] (n = u8/255*2-1, camera frame).
 * rays [B,14] = o(3) d(3) rgb(3) obj(1) keep(1) mono_normal(3); near/far [B]. */
int dh_gen_rays(const uint8_t* rgb, const int8_t* label, const uint8_t* normal, const float* R, const float* T,
                const float* Kinv, int H, int W, int n_frames, int frame, const int64_t* px, const int64_t* py, int64_t B,
                float* rays, float* near, float* far, void* stream);

/* coarse z = near + (far-near) linspace(0,1,n) + (t_rand-0.5)*2/n (t_rand [B] or NULL) and pts = o + d z
 * (upstream NeuSRenderer.render, App. A.5). */
int dh_coarse_samples(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* t_rand,
                      int64_t B, int n_samples, float* z, float* pts, void* stream);

/* upstream NeuSRenderer.up_sample + sample_pdf(det=True) (App. A.6): z,sdf [B,n_cur] -> z_new [B,n_new] and
 * pts_new [B*n_new,3].  n_cur <= 128, n_new <= 64. */
int dh_upsample_step(const float* rays_o, const float* rays_d, const float* z, const float* sdf, int64_t B, int n_cur,
                     int n_new, float inv_s, float* z_new, float* pts_new, void* stream);

/* upstream NeuSRenderer.cat_z_vals (App. A.6): stable sorted merge; sdf gathered alongside unless sdf_out is NULL
 * (the `last` step). */
int dh_merge_samples(const float* z, const float* z_new, const float* sdf, const float* sdf_new, int64_t B, int n_cur,
                     int n_new, float* z_out, float* sdf_out, void* stream);

/* section mid-points of render_core (App. A.7): pts [B*n,3] = o + d (z + dists/2). */
int dh_midpoints(const float* rays_o, const float* rays_d, const float* z, int64_t B, int n, float sample_dist, float* pts,
                 void* stream);

/* render_core tail (App. A.7): alpha from (sdf, normals, inv_s[0], cos_anneal), transmittance scan, compositing.
 * weights/cdf/inside_sphere [B,n]; color [B,3]; weight_sum/weight_max [B]; eik_partial [B,2] = per-ray
 * (sum relax*(|n|-1)^2, sum relax); normal_map [B,3] = sum_j w_j n_j (object frame) or NULL.  n <= 128.
 * background_rgb [3] or NULL. */
int dh_render_scan_fwd(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                       const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                       const float* background_rgb, int64_t B, int n, float* weights, float* color, float* weight_sum,
                       float* weight_max, float* cdf, float* inside_sphere, float* eik_partial, float* normal_map,
                       void* stream);

/* adjoint of dh_render_scan_fwd.  d_weight_sum, d_weights [B,n], d_gradients [B*n,3], d_normal_map [B,3] may be NULL; eik_coef[0] =
 * d(loss)/d(gradient_error) / (sum relax + 1e-5).  d_inv_s [B] holds per-ray partial sums. */
int dh_render_scan_bwd(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                       const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                       const float* background_rgb, int64_t B, int n, const float* d_color, const float* d_weight_sum,
                       const float* d_weights, const float* d_gradients, const float* d_normal_map, const float* eik_coef,
                       float* d_sdf, float* d_normals, float* d_colors, float* d_inv_s, void* stream);
/* The same, additionally d_rays_d [B,3] = d loss / d rays_d through true_cos = d . n (pose refinement). */
int dh_render_scan_bwd_rays(const float* rays_o, const float* rays_d, const float* z, const float* sdf, const float* normals,
                            const float* colors, const float* inv_s, float cos_anneal_ratio, float sample_dist,
                            const float* background_rgb, int64_t B, int n, const float* d_color, const float* d_weight_sum,
                            const float* d_weights, const float* d_gradients, const float* d_normal_map, const float* eik_coef,
                            float* d_sdf, float* d_normals, float* d_colors, float* d_inv_s, float* d_rays_d, void* stream);

/* Loss stack of the training step (upstream Runner.train, App. A.8, with Dynhor's hand gating): rays [B,14] as
 * written by dh_gen_rays; m = obj*keep (the keep-mask gating precedent: reference ObjTracker/utils/losses.py:69-71,
 * pose_initializtion.py:60-65,148-150).
 *   colour L1 over m / (sum m + 1e-5); eikonal = sum eik_partial[:,0] / (sum eik_partial[:,1] + 1e-5);
 *   mask BCE(clip(weight_sum,1e-3,1-1e-3), obj) over keep / (sum keep + 1e-5);
 *   normal (if normal_weight > 0): L1 + (1 - cos) between normalize(R normal_map) and the monocular normal over m.
 * stats[8] = loss, colour, eikonal, mask, normal, psnr, sum m, sum keep.  Also writes the adjoints d_color [B,3],
 * d_weight_sum [B], d_normal_map [B,3] and eik_coef[0] = igr_weight / (sum relax + 1e-5) for dh_render_scan_bwd. */
int dh_neus_loss(const float* color, const float* weight_sum, const float* normal_map, const float* eik_partial,
                 const float* rays, const float* R, int64_t B, float igr_weight, float mask_weight, float normal_weight,
                 float* stats, float* d_color, float* d_weight_sum, float* d_normal_map, float* eik_coef, void* stream);

/* Dense-correspondence reprojection term of the full loss stack (BASELINE.json configs[4]; reference README.md:43 names the
 * input folder `correspondence_infos` "obtained using DKM for reconstruction and outlier-voting" and nothing else, so the form
 * is this build's specification -- oracle/neus_oracle.py:correspondence_loss, DESIGN.md section 9, parity unpinned):
 *   corr [B,4] = (u_j, v_j, certainty, frame_j) per ray, certainty 0 = no match; poses x_cam = R x_obj + T of all n_frames
 *   frames (reference ObjTracker/run.py:166), K [3,3] row-major (run.py:119-123).
 *   t^ = sum_k weights[r,k] m_k (m = mid-point depths of z as in dh_render_scan_fwd), x = o + t^ d, pi = K (R_j x + T_j),
 *   s = |pi - (u_j,v_j)| / fx, rho = Huber(s; delta_px / fx), L = sum c v rho / (sum c v + 1e-5), v = [depth in camera j > 1e-3].
 * stats[4] = L, sum c v, certainty-weighted mean residual in pixels, corr_weight * L.  residual_px [B] feeds the outlier voting
 * (host side): 0 for rays without a match (certainty 0), +inf for a match that cannot be evaluated (partner frame out of range,
 * point behind the partner camera: a gross failure that must vote as an outlier).  d_weights [B,n] = d (corr_weight * L) / d
 * weights, written for EVERY ray (zeros where there is no match): pass it to dh_render_scan_bwd as d_weights.
 * pose_adjoints (may be null; [B,7], pose refinement together with this term): per ray d (corr_weight L) / d x [3] (x = o + t^ d
 * with t^ held fixed: the dependence of t^ on the geometry is what d_weights carries), d (corr_weight L) / d y [3] (y = R_j x +
 * T_j: the partner frame's pose gradient is sum_rays d_y x^T and sum_rays d_y) and t^. */
int dh_corr_loss(const float* rays_o, const float* rays_d, const float* z, const float* weights, const float* corr,
                 const float* R_all, const float* T_all, int n_frames, const float* K, int64_t B, int n, float sample_dist,
                 float delta_px, float corr_weight, float* stats, float* residual_px, float* d_weights, float* pose_adjoints,
                 void* stream);

/* ---- occupancy-grid ray marching, packed variable-length rays (BASELINE.json configs[3]) -------
 * The sampler of the instant-nsr-pl variant the reference names as its direction (README.md:11,13; code on an unmounted
 * branch; it calls nerfacc's OccupancyGrid / ray_marching).  Specification: oracle/occgrid_oracle.py (parity unpinned).
 *   occupancy: res^3 bytes (non-zero = occupied) over the cube [-radius, radius]^3, cell (ix,iy,iz) at (ix*res + iy)*res + iz.
 *   Step k of ray r is [t_k, t_k + step], t_k = near + (k + u[r]) step (u: one stratified offset per ray, null = 0.5); it is a
 *   sample iff t_k + step <= far and the cell of its mid-point is occupied; at most max_samples (<= 1024) per ray, front to back.
 *   half_step = (float)(0.5 * step) as the caller rounds it (kept separate so host and device agree bit for bit).
 * dh_march_count -> cnt [B]; the caller forms off = exclusive prefix sum (int64) and N = sum cnt, then dh_march_emit writes
 * t_start [N], the mid-point positions pts [N,3], the ray direction per sample dirs_pts [N,3] (pass it as `dirs` with
 * n_per_ray = 1 to the colour stages) and ray_idx [N].  No atomics: the packed order is a pure function of the inputs.
 * dh_march_emit's `keep` (device, [B], may be null): ray r emits only its first min(keep[r], max_samples) samples -- the caller
 * may lower the counts on the device (e.g. a per-ray cap chosen so that the total fits a fixed capacity) between the two calls
 * without reading them back. */
int dh_march_count(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* u,
                   const uint8_t* occupancy, int res, float radius, float step, float half_step, int max_samples, int64_t B,
                   int32_t* cnt, void* stream);
int dh_march_emit(const float* rays_o, const float* rays_d, const float* near, const float* far, const float* u,
                  const uint8_t* occupancy, int res, float radius, float step, float half_step, int max_samples, int64_t B,
                  const int64_t* off, const int32_t* keep, float* t_start, float* pts, float* dirs_pts, int32_t* ray_idx,
                  void* stream);
/* dh_render_scan_fwd / _bwd over packed rays: ray r owns samples [seg_off[r], seg_off[r] + seg_cnt[r]) of the packed arrays
 * (seg_cnt <= 1024: the wave takes 128 samples per trip and carries the transmittance), every interval is `step` long and
 * t_start holds the interval starts; per-sample outputs are packed too. */
int dh_render_scan_fwd_packed(const float* rays_o, const float* rays_d, const float* t_start, const float* sdf, const float* normals,
                              const float* colors, const float* inv_s, float cos_anneal_ratio, float step,
                              const float* background_rgb, int64_t B, const int64_t* seg_off, const int32_t* seg_cnt,
                              float* weights, float* color, float* weight_sum, float* weight_max, float* cdf,
                              float* inside_sphere, float* eik_partial, float* normal_map, void* stream);
int dh_render_scan_bwd_packed(const float* rays_o, const float* rays_d, const float* t_start, const float* sdf, const float* normals,
                              const float* colors, const float* inv_s, float cos_anneal_ratio, float step,
                              const float* background_rgb, int64_t B, const int64_t* seg_off, const int32_t* seg_cnt,
                              const float* d_color, const float* d_weight_sum, const float* d_weights, const float* d_gradients,
                              const float* d_normal_map, const float* eik_coef, float* d_sdf, float* d_normals, float* d_colors,
                              float* d_inv_s, void* stream);

/* ---- optimiser -----------------------------------------------------------------------------------------
 * torch.optim.Adam step (upstream Runner uses Adam, App. A.8; the reference's own optimisers are Adam too:
 * ObjTracker/pose_initializtion.py:346, jointopt.py:135-141) fused over the flat vector; step counts from 1;
 * grad_scale multiplies the gradient first (1/world_size after a sum all-reduce). */
int dh_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int64_t step, float grad_scale, void* stream);

/* ---- multiresolution hash-grid encoding (BASELINE.json configs[3]) -------------------------
 * The instant-nsr-pl variant the reference names as its direction (README.md:11,13; code on an unmounted branch).
 * Fixed geometry: 16 levels x 2 features, 2^19 entries per hashed level, resolutions 16 .. 2049; dense indexing on the
 * levels whose grid fits the table.  table: [dh_hashgrid_entries(), 2] fp32; x01 [n,3] in [0,1]; out [n,32].
 * Backward ACCUMULATES into d_table with float atomics (caller zeroes it). */
int64_t dh_hashgrid_entries(void);
int dh_hashgrid_level(int level, float* scale, uint32_t* resolution, uint32_t* offset, uint32_t* dense);
int dh_hashgrid_encode(const float* table, const float* x01, int64_t n, float* out, void* stream);
int dh_hashgrid_encode_backward(const float* x01, const float* d_out, int64_t n, float* d_table, void* stream);

/* ---- hash-grid model family: fused encoding + small MLPs (BASELINE.json configs[3]) ---------
 * geometry: [2 x01 - 1 (3), hash encoding (32)] -> 64 softplus(100) -> 13 (out[0] = sdf, all 13 = feature), x01 =
 * (x + radius) / (2 radius); normals by central finite differences with step eps (instant-nsr-pl 'finite_difference').
 * colour: [feature(13), SH degree-4 of the view dir (16), normal(3)] -> 64 relu -> 64 relu -> 3 sigmoid.
 * All five linears are weight-normed.  Flat parameter vector: the table first, then per linear bias | g | v (see
 * dh_hash_param_layout), `variance` between the two networks.  `packed` (dh_hash_packed_floats) holds the effective
 * weights; refresh it with dh_hash_pack_weights whenever params change.
 * dh_hash_param_layout: net 0 = geometry (layers 0..1), 1 = variance, 2 = colour (layers 0..2), 3 = table
 * (v_off = offset, out_dim = entries, in_dim = 2). */
int64_t dh_hash_num_params(void);
int64_t dh_hash_packed_floats(void);
int dh_hash_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim);
int dh_hash_pack_weights(const float* params, float* packed, void* stream);
/* workspace sizes (floats): *infer for dh_hash_geo_forward(save = 0), *total for a forward that a backward follows */
int dh_hash_workspace_floats(int64_t npts, int64_t* infer_floats, int64_t* total_floats);
/* sdf only (hierarchical up-sampling): pts [n,3] -> sdf [n] */
int dh_hash_sdf_nograd(const float* params, const float* packed, const float* pts, int64_t n, float radius, float* sdf,
                       void* stream);
/* sdf [n], feature [n,13], finite-difference gradient [n,3].  ws: caller-owned workspace (save = 0: *infer floats; save =
 * 1: *total floats, and the encodings of the 7 evaluations stay in it for the three backward calls below).
 * n_active (every stage of this family that takes it; device pointer to ONE int64, may be null): packed rays know their sample
 * count only on the device.  The caller then sizes buffers, workspace and n for a CAPACITY (a multiple of 8) and passes the
 * device-resident count: rows >= *n_active are neither read nor written and contribute nothing to any gradient, and no
 * device -> host read is needed.  All stages of one forward / backward must be given the same n and n_active. */
int dh_hash_geo_forward(const float* params, const float* packed, const float* pts, int64_t n, float radius, float eps,
                        float* ws, int save, float* sdf, float* feature, float* gradient, const int64_t* n_active, void* stream);
/* colour [n,3]; dirs [n / n_per_ray, 3] */
int dh_hash_color_forward(const float* packed, const float* feature, const float* normals, const float* dirs,
                          int n_per_ray, int64_t n, float* color, const int64_t* n_active, void* stream);
/* Adjoint, three calls in this order on the workspace dh_hash_geo_forward(save = 1) filled:
 *   colour   : d_color [n,3] -> d_feature [n,13] (written) and d_normals [n,3] (ACCUMULATED onto the caller's values)
 *   geometry : d_sdf [n], d_feature, d_normals (= cotangent of the finite-difference gradient)
 *   weights  : every parameter gradient -> grad [dh_hash_num_params()] (table part zeroed then scattered with float
 *              atomics: order-dependent in the last bits, unlike the NeuS fp32 path); the variance slot is left untouched */
int dh_hash_color_backward(const float* packed, const float* feature, const float* normals, const float* dirs,
                           const float* d_color, int n_per_ray, int64_t n, float* ws, float* d_feature, float* d_normals,
                           const int64_t* n_active, void* stream);
int dh_hash_geo_backward(const float* params, const float* packed, const float* pts, const float* d_sdf,
                         const float* d_feature, const float* d_normals, int64_t n, float radius, float eps, float* ws,
                         const int64_t* n_active, void* stream);
int dh_hash_weight_grads(const float* params, const float* packed, int64_t n, float* ws, float* grad, const int64_t* n_active,
                         void* stream);
/* The same in two parts, for data-parallel callers: parts & 1 writes the table gradient (the leading dh_hashgrid_entries() x 2 floats
 * of grad: 49 MB), parts & 2 the five small linears.  Calling the table part, starting its all-reduce on a side stream, then the
 * linears, hides the large collective behind the small weight-gradient GEMMs (dynhor_amd/hash_fields.py).
 * parts & 4 selects how the table scatter adds.  Set (5, 7; dh_hash_weight_grads = 7): every contribution is converted to 2^-48 fixed
 * point and added by an INTEGER atomic to an int64 accumulator in the workspace (dh_hash_workspace_floats counts it), converted to
 * float once -- integer addition is associative, so the result is bit-identical from launch to launch; resolution 3.6e-15.  Range: one
 * contribution below 64, a sum exact up to +-16,384 (256 same-signed contributions at the limit).  A non-finite contribution or one
 * beyond 64 turns the WHOLE table gradient into NaN; an entry whose accumulator ends at |sum| >= 16,384 is NaN itself (the guard band
 * covers every true sum up to 3 x that; a finite wrong value would need more than 768 same-signed contributions at the
 * limit on one entry).  Same speed as the float form on MI355X (both are bound by the memory side's atomic request rate).  Clear (1,
 * 3): float atomics, whose sums depend on the order in which the memory side sees the requests (last-bit differences from launch to
 * launch: the only such sums in the library; kept for comparison).  The merge ablations of dh_hash_set_scatter_mode apply to the float
 * form only: with bit 4 set and a scatter mode other than 0 selected the call returns DH_ERR_BAD_ARG instead of ignoring the mode. */
int dh_hash_weight_grads_parts(const float* params, const float* packed, int64_t n, float* ws, float* grad, const int64_t* n_active,
                               int parts, void* stream);
/* Diagnosis only (scripts/psnr_parity.py ablations): how the FLOAT-atomic table scatter (dh_hash_weight_grads_parts with parts 1 or
 * 3) merges table-gradient adds before they reach memory.  0 (default, shipping) = 7-evaluation blending + ray-run merging + quad-lane
 * packing; 1 = no ray-run merging; 2 = neither (one atomic per evaluation corner, tcnn's scheme).  Same sums up to float-atomic
 * ordering.  The fixed-point form (parts bit 4, dh_hash_weight_grads) always merges: it refuses to run while a mode other than 0 is
 * selected (DH_ERR_BAD_ARG). */
int dh_hash_set_scatter_mode(int mode);

#ifdef __cplusplus
}
#endif
#endif /* DYNHOR_HIP_H */
