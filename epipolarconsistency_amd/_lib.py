"""ctypes binding of libecc_hip.so (the C ABI in include/ecc_hip.h).

The HIP extension is the product: if it has not been built this module raises -- there is no
eager/CPU fallback anywhere in the package.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ECC_HIP_LIB: another build of the same library (scripts/sanitize.sh: host code under UndefinedBehaviorSanitizer)
LIB_PATH = os.environ.get("ECC_HIP_LIB") or os.path.join(_HERE, "libecc_hip.so")

ECC_OK = 0
FILTER_DERIVATIVE, FILTER_RAMP, FILTER_NONE = 0, 1, 2
POST_IDENTITY, POST_SQUARE_ROOT, POST_LOGARITHM = 0, 1, 2
RADON_EXACT, RADON_FMA = 0, 1
SAMPLING_AUTO, SAMPLING_POLYNOMIAL, SAMPLING_PER_SAMPLE, SAMPLING_REFERENCE = 0, 1, 2, 3


class EccError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libecc_hip error %d: %s" % (code, message))
        self.code = code


class PreprocessConfig(C.Structure):
    """struct ecc_preprocess_config (include/ecc_hip.h)."""
    _fields_ = [("process", C.c_int32), ("normalize", C.c_int32), ("bias", C.c_double), ("scale", C.c_double),
                ("apply_log", C.c_int32), ("gaussian_sigma", C.c_double), ("half_kernel_width", C.c_int32),
                ("flip_u", C.c_int32), ("flip_v", C.c_int32), ("zero", C.c_int32 * 4), ("feather", C.c_int32 * 4),
                ("n_blanks", C.c_int32), ("blanks", C.c_void_p)]


_lib = None

# name -> (restype, argtypes); every symbol declared in include/ecc_hip.h
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_pi, _pd, _pf = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_float)
SIGNATURES = {
    "ecc_last_error": (C.c_char_p, []),
    "ecc_version": (_i, []),
    "ecc_device_count": (_i, []),
    "ecc_ctx_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "ecc_ctx_destroy": (_i, [_vp]),
    "ecc_ctx_synchronize": (_i, [_vp]),
    "ecc_radon_compute": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_radon_compute_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_radon_compute_into": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ecc_radon_set_arithmetic": (_i, [_vp, _i]),
    "ecc_radon_get_arithmetic": (_i, [_vp, _pi]),
    "ecc_radon_compute_linear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ecc_dtr_from_device_linear": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_metric_evaluate_external": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, C.c_float, C.c_float, _i]),
    "ecc_dtr_from_host": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_dtr_readback": (_i, [_vp, _vp]),
    "ecc_dtr_info": (_i, [_vp, _pi, _pi, _pi, _pi, _pi, _pd, _pd]),
    "ecc_dtr_device_view": (_i, [_vp, C.POINTER(_vp), _pi, _pi]),
    "ecc_dtr_slab_floats": (_i64, [_i, _i]),
    "ecc_dtr_wrap_device": (_i, [_vp, _vp, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_dtr_destroy": (_i, [_vp]),
    "ecc_metric_create": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "ecc_metric_destroy": (_i, [_vp]),
    "ecc_metric_refresh_dtrs": (_i, [_vp, _i, _i]),
    "ecc_metric_set_projections": (_i, [_vp, _vp, _i]),
    "ecc_metric_debug_geometry": (_i, [_vp, _vp, _vp]),
    "ecc_metric_set_params": (_i, [_vp, _d, _d, _i]),
    "ecc_metric_set_sampling": (_i, [_vp, _i]),
    "ecc_metric_set_incremental": (_i, [_vp, _i]),
    "ecc_metric_set_record_reuse": (_i, [_vp, _i]),
    "ecc_metric_set_small_eval": (_i, [_vp, _i]),
    "ecc_metric_publish_scalar": (_i, [_vp, _vp]),
    "ecc_metric_wait_scalar": (_i, [_vp, _pd]),
    "ecc_metric_last_evaluated_pairs": (_i, [_vp, C.POINTER(_i64)]),
    "ecc_metric_device_bytes": (_i, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "ecc_metric_get_object_radius": (_i, [_vp, _pd]),
    "ecc_metric_evaluate_all": (_i, [_vp, _vp, _pd]),
    "ecc_metric_evaluate_poses": (_i, [_vp, _i, _vp, _i, _vp]),
    "ecc_metric_evaluate_poses_strided": (_i, [_vp, _i, _vp, _i, _i, _i, _vp]),
    "ecc_metric_evaluate_pose_deltas": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "ecc_metric_set_pose_batching": (_i, [_vp, _i]),
    "ecc_metric_last_batched_poses": (_i, [_vp, C.POINTER(_i64)]),
    "ecc_metric_evaluate_range": (_i, [_vp, _i64, _i64, _vp, _pd]),
    "ecc_metric_evaluate_range_async": (_i, [_vp, _i64, _i64, _vp, _vp]),
    "ecc_metric_evaluate_pairs": (_i, [_vp, _vp, _i, _vp, _pd]),
    "ecc_metric_debug_K01": (_i, [_vp, _i64, _i64, _vp]),
    "ecc_metric_debug_polynomials": (_i, [_vp, _i64, _i64, _vp]),
    "ecc_metric_pair_samples_bound": (_i, [_vp, _pi]),
    "ecc_metric_evaluate_for_image_pair": (_i, [_vp, _i, _i, _i, _pi, _vp, _vp, _vp, _vp, _vp, _vp, _pd]),
    "ecc_get_ij": (None, [_i64, _i, _pi, _pi]),
    "ecc_host_pinvT": (None, [_vp, _vp]),
    "ecc_host_source_position": (None, [_vp, _vp]),
    "ecc_host_object_radius": (_d, [_vp, _i, _i]),
    "ecc_direct_create": (_i, [_vp, _i, _vp, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_direct_destroy": (_i, [_vp]),
    "ecc_direct_update_images": (_i, [_vp]),
    "ecc_direct_set_projections": (_i, [_vp, _vp, _i]),
    "ecc_direct_set_params": (_i, [_vp, _d, _d, _i]),
    "ecc_direct_get_object_radius": (_i, [_vp, _pd]),
    "ecc_direct_evaluate": (_i, [_vp, _vp, _pd]),
    "ecc_direct_lines_bound": (_i, [_vp, _pi]),
    "ecc_direct_evaluate_for_image_pair": (_i, [_vp, _i, _i, _i, _pi, _vp, _vp, _vp, _vp, _pd]),
    "ecc_direct_evaluate_for_image_pair_kappas": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _pd]),
    "ecc_preprocess_defaults": (None, [_vp]),
    "ecc_preprocess": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "ecc_host_intrinsics": (None, [_vp, _pf, _pf, _pf]),
    "ecc_host_angular_range": (None, [_vp, _vp, _d, _pd, _pd]),
    "ecc_host_angular_step": (_d, [_vp, _vp, _i, _i]),
    "ecc_host_iso_center": (None, [_vp, _i, _vp]),
    "ecc_host_line_to_sample_dtr": (_i, [_vp, C.c_float]),
    "ecc_group_create": (_i, [_i, _vp, C.POINTER(_vp)]),
    "ecc_group_destroy": (_i, [_vp]),
    "ecc_group_size": (_i, [_vp]),
    "ecc_group_ctx": (_i, [_vp, _i, C.POINTER(_vp)]),
    "ecc_group_radon_compute_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "ecc_pair_shard": (None, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "ecc_pair_shards_balanced": (_i, [_vp, _i, _d, _i, _vp]),
    "ecc_metric_balanced_shards": (_i, [_vp, _i, _vp]),
    "ecc_comm_available": (_i, []),
    "ecc_comm_unique_id": (_i, [_vp]),
    "ecc_comm_create": (_i, [_vp, _vp, _i, _i, _vp]),
    "ecc_comm_destroy": (_i, [_vp]),
    "ecc_metric_evaluate_range_allreduce": (_i, [_vp, _vp, _i64, _i64, _vp]),
    "ecc_group_metric_rebalance": (_i, [_vp]),
    "ecc_group_metric_evaluate_poses": (_i, [_vp, _i, _vp, _i, _vp]),
    "ecc_group_metric_create": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "ecc_group_debug_force_replica": (_i, [_i]),
    "ecc_group_metric_destroy": (_i, [_vp]),
    "ecc_group_metric_set_projections": (_i, [_vp, _vp, _i]),
    "ecc_group_metric_set_params": (_i, [_vp, _d, _d, _i]),
    "ecc_group_metric_set_sampling": (_i, [_vp, _i]),
    "ecc_group_metric_set_incremental": (_i, [_vp, _i]),
    "ecc_group_metric_get_object_radius": (_i, [_vp, _pd]),
    "ecc_group_metric_evaluate_all": (_i, [_vp, _vp, _pd]),
    "ecc_group_metric_rank_metric": (_i, [_vp, _i, C.POINTER(_vp)]),
    "ecc_exchange_open": (_i, [C.c_char_p, _i, _i, C.POINTER(C.c_void_p)]),
    "ecc_exchange_sum": (_i, [_vp, _d, _pd]),
    "ecc_exchange_close": (_i, [_vp]),
    "ecc_debug_set_poly_tolerance": (_i, [_vp, C.c_float]),
    "ecc_debug_set_small_eval_bound": (_i, [_vp, _i64]),
    "ecc_debug_set_result_polling": (_i, [_i]),
    "ecc_debug_set_quad_copies": (_i, [_vp, _i]),
    "ecc_ctx_set_quad_copies": (_i, [_vp, _i]),
    "ecc_debug_small_stamps": (_i, [_vp, _i]),
    "ecc_debug_step_stamps": (_i, [_vp, _vp]),
    "ecc_ctx_enable_timing": (_i, [_vp, _i]),
    "ecc_ctx_last_kernel_ms": (_i, [_vp, _i, _pf]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -m epipolarconsistency_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime; when libecc_hip.so (linked against /opt/rocm)
        # initialises the GPU first, a later torch.cuda initialisation in the same process fails with "No HIP GPUs
        # are available" (seen on the MI355X boxes).  The other order works, so let torch go first when it is there.
        if not os.environ.get("ECC_NO_TORCH_PRELOAD"):
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.init()
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def check(code):
    if code != ECC_OK:
        raise EccError(code, lib().ecc_last_error().decode("utf-8", "replace"))
