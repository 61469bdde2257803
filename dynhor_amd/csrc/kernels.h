// Internal launch-function declarations (one per kernel family).  All enqueue on the caller's stream,
// allocate nothing, and return 0 / negative dh error codes (include/dynhor_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dh {

int launch_pack_weights(const float* params, float* packed, hipStream_t stream);

// MLP chains (kernels_mlp.hip).  npts is padded by the caller to a multiple of 128 for saved buffers.
int launch_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, int grid, hipStream_t stream);

int launch_sdf_fwd_train(const float* packed, const float* pts, int64_t npts, float* sdf, float* feat, float* act,
                         float* eaux, int grid, hipStream_t stream);
int launch_sdf_grad(const float* packed, const float* pts, int64_t npts, const float* act, float* asave, float* normals,
                    int grid, hipStream_t stream);
int launch_color_fwd(const float* packed, const float* pts, const float* dirs, int n_per_ray, const float* normals,
                     const float* feat, int64_t npts, float* color, float* cact, float* caux, int save, int grid,
                     hipStream_t stream);

}  // namespace dh
