"""ctypes binding of libdynhor_hip.so (include/dynhor_hip.h).  Fails loudly when the library is missing."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdynhor_hip.so")
_LIB = None

_vp, _i64, _i32, _f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float

# name -> (restype, argtypes); must list every symbol include/dynhor_hip.h declares (tests check this)
SIGNATURES = {
    "dh_version": (_i32, []),
    "dh_strerror": (ctypes.c_char_p, [_i32]),
    "dh_set_arithmetic": (_i32, [_i32]),
    "dh_get_arithmetic": (_i32, []),
    "dh_hash_set_scatter_mode": (_i32, [_i32]),
    "dh_num_params": (_i64, []),
    "dh_packed_floats": (_i64, []),
    "dh_param_layout": (_i32, [_i32, _i32, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                               ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "dh_packed_section": (_i32, [_i32, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dh_pack_weights": (_i32, [_vp, _vp, _vp]),
    "dh_pack_weights_ex": (_i32, [_i32, _vp, _vp, _vp]),
    "dh_sdf_nograd": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "dh_workspace_floats": (_i32, [_i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dh_range_words": (_i32, [ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_f32)]),
    "dh_mlp_forward": (_i32, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dh_sdf_forward": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_sdf_gradient": (_i32, [_vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dh_color_forward": (_i32, [_vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dh_color_backward": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_sdf_tangent": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "dh_sdf_backward": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "dh_color_backward_rays": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dh_sdf_backward_rays": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_weight_grads_gemm": (_i32, [_i64, _vp, _vp]),
    "dh_weight_grads_fold": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_mlp_backward": (_i32, [_vp, _vp, _vp, _i64] + [_vp] * 7),
    # the same stages with the arithmetic passed explicitly (no process-global state)
    "dh_sdf_nograd_ex": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp]),
    "dh_mlp_forward_ex": (_i32, [_i32, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dh_sdf_forward_ex": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_sdf_gradient_ex": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dh_color_forward_ex": (_i32, [_i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _i32, _vp]),
    "dh_color_backward_ex": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_sdf_tangent_ex": (_i32, [_i32, _vp, _vp, _vp, _i64, _vp, _vp]),
    "dh_sdf_backward_ex": (_i32, [_i32, _vp, _vp, _i64, _vp, _vp]),
    "dh_color_backward_rays_ex": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dh_sdf_backward_rays_ex": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "dh_weight_grads_gemm_ex": (_i32, [_i32, _i64, _vp, _vp]),
    "dh_mlp_backward_ex": (_i32, [_i32, _vp, _vp, _vp, _i64] + [_vp] * 7),
    "dh_adam_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _i64, _f32, _vp]),
    "dh_hashgrid_entries": (_i64, []),
    "dh_hashgrid_level": (_i32, [_i32, ctypes.POINTER(_f32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                 ctypes.POINTER(ctypes.c_uint32)]),
    "dh_hashgrid_encode": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "dh_hashgrid_encode_backward": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "dh_hash_num_params": (_i64, []),
    "dh_hash_packed_floats": (_i64, []),
    "dh_hash_param_layout": (_i32, [_i32, _i32, ctypes.POINTER(_i64), ctypes.POINTER(_i64), ctypes.POINTER(_i64),
                                    ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    "dh_hash_pack_weights": (_i32, [_vp, _vp, _vp]),
    "dh_hash_workspace_floats": (_i32, [_i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "dh_hash_sdf_nograd": (_i32, [_vp, _vp, _vp, _i64, _f32, _vp, _vp]),
    "dh_hash_geo_forward": (_i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "dh_hash_color_forward": (_i32, [_vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "dh_hash_color_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "dh_hash_geo_backward": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp, _vp]),
    "dh_hash_weight_grads": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dh_hash_weight_grads_parts": (_i32, [_vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp]),
    "dh_gen_rays": (_i32, [_vp] * 6 + [_i32, _i32, _i32, _i32, _vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "dh_coarse_samples": (_i32, [_vp] * 5 + [_i64, _i32, _vp, _vp, _vp]),
    "dh_upsample_step": (_i32, [_vp] * 4 + [_i64, _i32, _i32, _f32, _vp, _vp, _vp]),
    "dh_merge_samples": (_i32, [_vp] * 4 + [_i64, _i32, _i32, _vp, _vp, _vp]),
    "dh_midpoints": (_i32, [_vp] * 3 + [_i64, _i32, _f32, _vp, _vp]),
    "dh_render_scan_fwd": (_i32, [_vp] * 7 + [_f32, _f32, _vp, _i64, _i32] + [_vp] * 9),
    "dh_render_scan_bwd": (_i32, [_vp] * 7 + [_f32, _f32, _vp, _i64, _i32] + [_vp] * 11),
    "dh_render_scan_bwd_rays": (_i32, [_vp] * 7 + [_f32, _f32, _vp, _i64, _i32] + [_vp] * 12),
    "dh_march_count": (_i32, [_vp] * 6 + [_i32, _f32, _f32, _f32, _i32, _i64, _vp, _vp]),
    "dh_march_emit": (_i32, [_vp] * 6 + [_i32, _f32, _f32, _f32, _i32, _i64] + [_vp] * 7),
    "dh_render_scan_fwd_packed": (_i32, [_vp] * 7 + [_f32, _f32, _vp, _i64] + [_vp] * 11),
    "dh_render_scan_bwd_packed": (_i32, [_vp] * 7 + [_f32, _f32, _vp, _i64] + [_vp] * 13),
    "dh_neus_loss": (_i32, [_vp] * 6 + [_i64, _f32, _f32, _f32] + [_vp] * 6),
    "dh_corr_loss": (_i32, [_vp] * 7 + [_i32, _vp, _i64, _i32, _f32, _f32, _f32, _vp, _vp, _vp, _vp, _vp]),
}


class DynhorHipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DynhorHipError(
                f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
                "Build it with: python -c 'import __graft_entry__ as g; g.build()'")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


ARITH_SPLIT_BF16, ARITH_FP32_MFMA, ARITH_SPLIT_F16 = 0, 1, 2
ARITH_DEFAULT = ARITH_SPLIT_F16
ARITH_NAMES = {"split_f16": ARITH_SPLIT_F16, "split_bf16": ARITH_SPLIT_BF16, "fp32_mfma": ARITH_FP32_MFMA}

# stage (StageTimer / bench.py name) -> HIP kernel that runs it, per arithmetic mode.  bench.py's roofline block and
# scripts/make_traffic_json.py (rocprofv3 PMC -> HBM bytes per launch) both read THIS table, so a renamed or removed kernel
# cannot leave a stale entry behind.
STAGE_KERNELS = {
    ARITH_SPLIT_F16: {"weight_grads_gemm": "dw_f16x2_kernel", "sdf_forward": "sdf_fwd_train_h_kernel",
                      # (colour forward: the tile-PAIR form, csrc/chain_pair.hip, at the bench's launch size -- include/dynhor_hip.h DH_CHAIN_FORM_*;
                      #  launches below 2 x #CUs tiles run color_fwd_h_kernel)
                      "sdf_gradient": "sdf_grad_h_kernel", "color_forward": "color_fwd_p_kernel",
                      "color_backward": "color_bwd_h_kernel", "sdf_tangent": "sdf_tangent_h_kernel",
                      "sdf_backward": "sdf_bwd_h_kernel", "sdf_nograd_coarse": "sdf_nograd_h_kernel",
                      "sdf_nograd_fine": "sdf_nograd_h_kernel"},
    ARITH_SPLIT_BF16: {"weight_grads_gemm": "dw_bf16x3_kernel", "sdf_forward": "sdf_fwd_train_t_kernel",
                       "sdf_gradient": "sdf_grad_s_kernel", "color_forward": "color_fwd_s_kernel",
                       "color_backward": "color_bwd_s_kernel", "sdf_tangent": "sdf_tangent_s_kernel",
                       "sdf_backward": "sdf_bwd_s_kernel", "sdf_nograd_coarse": "sdf_nograd_t_kernel",
                       "sdf_nograd_fine": "sdf_nograd_t_kernel"},
    ARITH_FP32_MFMA: {"weight_grads_gemm": "dw_lds_kernel", "sdf_forward": "sdf_fwd_train_kernel",
                      "sdf_gradient": "sdf_grad_kernel", "color_forward": "color_fwd_kernel",
                      "color_backward": "color_bwd_kernel", "sdf_tangent": "sdf_tangent_kernel",
                      "sdf_backward": "sdf_bwd_kernel", "sdf_nograd_coarse": "sdf_nograd_kernel",
                      "sdf_nograd_fine": "sdf_nograd_kernel"},
}
# hash family: dh_hash_weight_grads_parts runs as two timed stages on two streams (hash_fields.HashNeuSRenderer._weight_grads): the table
# scatter (parts & 1) and the five small linears' weight gradients (parts & 2)
HASH_STAGE_KERNELS = {"hash_weight_grads": "hash_table_bwd_kernel", "hash_weight_grads_mlp": "small_dw_kernel"}
# every kernel a stage launches (scripts/make_traffic_json.py sums their counters)
HASH_STAGE_LAUNCHES = {"hash_weight_grads": ["hash_table_bwd_kernel", "hash_fix_to_float_kernel"],
                       "hash_weight_grads_mlp": ["small_dw_kernel", "small_dw_reduce_kernel", "hash_fold_kernel"]}


def set_arithmetic(mode: int):
    """dh_set_arithmetic: the DEFAULT arithmetic of the entry points without an arithmetic argument (2 = two-piece fp16 split,
    shipping; 0 = three-piece bf16 split; 1 = native fp32-MFMA twins).  Process-wide; the renderers pass theirs per call."""
    check(lib().dh_set_arithmetic(int(mode)))


def get_arithmetic() -> int:
    return int(lib().dh_get_arithmetic())


_ROCTX = False


def roctx():
    """ctypes handle of the ROCm marker library (roctxRangePushA / roctxRangePop), or None when it is not installed.  rocprofv3
    --marker-trace records the ranges; outside a profiler they cost two empty calls.  librocprofiler-sdk-roctx is the library
    rocprofv3 intercepts; libroctx64 is the older name of the same interface."""
    global _ROCTX
    if _ROCTX is False:
        _ROCTX = None
        for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
            try:
                L = ctypes.CDLL(name)
                L.roctxRangePushA.argtypes = [ctypes.c_char_p]
                L.roctxRangePushA.restype = ctypes.c_int
                L.roctxRangePop.restype = ctypes.c_int
                _ROCTX = L
                break
            except (OSError, AttributeError):
                continue
    return _ROCTX


def check(status: int):
    if status != 0:
        raise DynhorHipError(f"dynhor_hip status {status}: {lib().dh_strerror(status).decode()}")


def ptr(t: torch.Tensor):
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def param_layout(net: int, layer: int):
    b, g, v = _i64(), _i64(), _i64()
    o, i = _i32(), _i32()
    check(lib().dh_param_layout(net, layer, ctypes.byref(b), ctypes.byref(g), ctypes.byref(v),
                                ctypes.byref(o), ctypes.byref(i)))
    return b.value, g.value, v.value, o.value, i.value


def packed_section(section: int):
    """(float offset, float count) of a section of the packed weight buffer (dh_packed_section)."""
    o, n = _i64(), _i64()
    check(lib().dh_packed_section(int(section), ctypes.byref(o), ctypes.byref(n)))
    return o.value, n.value


def range_words():
    """(float offset of the activation maximum, of the arithmetic tag, limit) in a workspace."""
    a, t, lim = _i64(), _i64(), _f32()
    check(lib().dh_range_words(ctypes.byref(a), ctypes.byref(t), ctypes.byref(lim)))
    return a.value, t.value, lim.value


def workspace_floats(npts: int):
    """(forward-only, training-forward, total) workspace sizes in floats."""
    i, f, t = _i64(), _i64(), _i64()
    check(lib().dh_workspace_floats(npts, ctypes.byref(i), ctypes.byref(f), ctypes.byref(t)))
    return i.value, f.value, t.value
