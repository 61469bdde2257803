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
 *   - return 0 on success or a negative dh_status; never throws, never exits.  Re-entrant, no global state.
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

/* ---- parameter vector / packed weights -----------------------------------------------------------------
 * One flat fp32 vector of dh_num_params() == 802,491 values in state_dict order (SURVEY.md §5 checkpoint row:
 * lin{l}.bias, lin{l}.weight_g, lin{l}.weight_v for sdf_network_fine lin0..8, variance, color_network_fine
 * lin0..4).  dh_param_layout: net 0 = SDFNetwork, 1 = SingleVarianceNetwork (layer ignored; only v_off), 2 =
 * RenderingNetwork. */
int64_t dh_num_params(void);
int64_t dh_packed_floats(void);
int dh_param_layout(int net, int layer, int64_t* bias_off, int64_t* g_off, int64_t* v_off, int* out_dim, int* in_dim);

/* weight-norm (W = g v/||v||, upstream nn.utils.weight_norm) + MFMA-operand packing; once per optimiser step. */
int dh_pack_weights(const float* params, float* packed, void* stream);

/* SDFNetwork.sdf(pts) under no_grad (upstream NeuSRenderer.render up-sampling loop).
 * pts [npts,3] -> sdf [npts]. */
int dh_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, void* stream);

/* Workspace size (floats) for a render call over npts fine sample points: fwd_floats is what the forward pass
 * writes (saved activations), total_floats additionally covers the backward pass. */
int dh_workspace_floats(int64_t npts, int64_t* fwd_floats, int64_t* total_floats);

/* The MLP part of upstream NeuSRenderer.render_core (App. A.7) on npts points (point i belongs to ray
 * i / n_per_ray): sdf_network(pts) -> sdf [npts], feature (kept in ws); sdf_network.gradient(pts) -> normals
 * [npts,3]; color_network(pts, normals, dirs, feature) -> color [npts,3] (post-sigmoid).  Saves what
 * dh_mlp_backward needs into ws. */
int dh_mlp_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                   float* sdf, float* normals, float* color, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYNHOR_HIP_H */
