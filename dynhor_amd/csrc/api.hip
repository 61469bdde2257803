// extern "C" boundary (include/dynhor_hip.h).  Argument checking + launch dispatch only.
#include "../../include/dynhor_hip.h"
#include "kernels.h"
#include "layout.h"
#include "workspace.h"

using namespace dh;

namespace {
inline bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
constexpr int DEFAULT_GRID = 256;   // one persistent workgroup per CU
}  // namespace

extern "C" {

int dh_version(void) { return 1; }

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
int64_t dh_packed_floats(void) { return PACK.total; }

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
    return launch_pack_weights(params, packed, static_cast<hipStream_t>(stream));
}

int dh_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, void* stream) {
    if (npts < 0) return DH_ERR_BAD_ARG;
    if (npts == 0) return DH_OK;
    if (!packed || !pts || !sdf || misaligned16(packed)) return DH_ERR_BAD_ARG;
    return launch_sdf_nograd(packed, pts, npts, sdf, DEFAULT_GRID, static_cast<hipStream_t>(stream));
}

int dh_workspace_floats(int64_t npts, int64_t* fwd_floats, int64_t* total_floats) {
    if (npts < 0 || !fwd_floats || !total_floats) return DH_ERR_BAD_ARG;
    const Workspace w = carve_workspace(nullptr, npts);
    *fwd_floats = w.fwd_floats;
    *total_floats = w.total_floats;
    return DH_OK;
}

int dh_mlp_forward(const float* packed, const float* pts, const float* dirs, int n_per_ray, int64_t npts, float* ws,
                   float* sdf, float* normals, float* color, void* stream) {
    if (npts < 0 || n_per_ray <= 0) return DH_ERR_BAD_ARG;
    if (npts == 0) return DH_OK;
    if (!packed || !pts || !dirs || !ws || !sdf || !normals || !color || misaligned16(packed) || misaligned16(ws))
        return DH_ERR_BAD_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Workspace w = carve_workspace(ws, npts);
    int rc = launch_sdf_fwd_train(packed, pts, npts, sdf, w.feat, w.act, w.eaux, DEFAULT_GRID, st);
    if (rc) return rc;
    rc = launch_sdf_grad(packed, pts, npts, w.act, w.asave, normals, DEFAULT_GRID, st);
    if (rc) return rc;
    return launch_color_fwd(packed, pts, dirs, n_per_ray, normals, w.feat, npts, color, w.cact, w.caux, 1, DEFAULT_GRID, st);
}

}  // extern "C"
