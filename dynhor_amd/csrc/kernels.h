// Internal launch-function declarations (one per kernel family).  All enqueue on the caller's stream,
// allocate nothing, and return 0 / negative dh error codes (include/dynhor_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dh {

int launch_pack_weights(const float* params, float* packed, hipStream_t stream);

// MLP chains (kernels_mlp.hip).  npts is padded by the caller to a multiple of 128 for saved buffers.
int launch_sdf_nograd(const float* packed, const float* pts, int64_t npts, float* sdf, int grid, hipStream_t stream);

}  // namespace dh
