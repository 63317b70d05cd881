"""ctypes binding of libcascadeclassifier_amd.so (the C ABI declared in include/cascadeclassifier_amd.h).

There is no fallback of any kind: if the shared library has not been built, importing the bindings raises; if no
HIP device is usable, every compute entry point returns CC_ERR_NO_DEVICE and the wrappers raise CascadeError.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCAMD_LIB selects another build of the same library (tuning experiments); the default is the in-tree build.
LIB_PATH = os.environ.get("CCAMD_LIB") or os.path.join(_HERE, "lib", "libcascadeclassifier_amd.so")

CC_OK = 0
CC_ERR_INVALID_ARG = -1
CC_ERR_NO_DEVICE = -2
CC_ERR_HIP = -3
CC_ERR_IO = -4
CC_ERR_PARSE = -5
CC_ERR_UNSUPPORTED = -6
CC_ERR_BUFFER_TOO_SMALL = -7
CC_ERR_OUT_OF_RANGE = -8

CC_FEATURE_HAAR, CC_FEATURE_LBP, CC_FEATURE_HOG = 0, 1, 2
CC_HAAR_BASIC, CC_HAAR_CORE, CC_HAAR_ALL = 0, 1, 2


class CascadeError(RuntimeError):
    """A C-ABI call failed; .status is the cc_status code (the reference throws cv::Exception here)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[cc_status {status}] {message}")
        self.status = status


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class CascadeInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("feature_type", "win_w", "win_h", "n_stages", "n_weak", "n_nodes", "n_leaves",
                                         "n_features", "max_cat_count", "subset_size", "max_nodes_per_tree", "has_tilted")]


class DetectParams(C.Structure):
    _fields_ = [("scale_factor", C.c_double), ("min_neighbors", C.c_int32), ("min_w", C.c_int32), ("min_h", C.c_int32),
                ("max_w", C.c_int32), ("max_h", C.c_int32)]


class ScaleInfo(C.Structure):
    _fields_ = [("scale", C.c_float), ("width", C.c_int32), ("height", C.c_int32), ("ystep", C.c_int32), ("nx", C.c_int32),
                ("ny", C.c_int32), ("win_w", C.c_int32), ("win_h", C.c_int32)]


class DetectorTimings(C.Structure):
    _fields_ = [("resize_ms", C.c_double), ("integral_ms", C.c_double), ("eval_ms", C.c_double), ("finalize_ms", C.c_double),
                ("resize_launches", C.c_int64), ("integral_launches", C.c_int64), ("eval_launches", C.c_int64),
                ("finalize_launches", C.c_int64), ("frames", C.c_int64), ("grid_windows", C.c_int64),
                ("integral_elems", C.c_int64), ("eval_step1_ms", C.c_double)]


class Split(C.Structure):
    _fields_ = [("found", C.c_int32), ("var_idx", C.c_int32), ("quality", C.c_float), ("ord_c", C.c_float),
                ("split_point", C.c_int32), ("subset", C.c_int32 * 8)]


class HaarFeatureC(C.Structure):
    _fields_ = [("tilted", C.c_int32), ("r", (C.c_int32 * 4) * 3), ("w", C.c_float * 3)]


# every symbol include/cascadeclassifier_amd.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _d = C.c_void_p, C.c_int, C.c_size_t, C.c_double
_pp = C.POINTER(C.c_void_p)
SIGNATURES = {
    "cc_last_error": (C.c_char_p, []),
    "cc_version": (_i, []),
    "cc_device_count": (_i, []),
    "cc_cascade_load_xml": (_i, [C.c_char_p, _pp]),
    "cc_cascade_load_xml_mem": (_i, [C.c_char_p, _sz, _pp]),
    "cc_cascade_destroy": (None, [_vp]),
    "cc_cascade_info_get": (_i, [_vp, C.POINTER(CascadeInfo)]),
    "cc_cascade_stages": (_i, [_vp, _pp, _pp, _pp]),
    "cc_cascade_stumps": (_i, [_vp, _pp, _pp, _pp, _pp, _pp]),
    "cc_cascade_features": (_i, [_vp, _pp, _pp, _pp]),
    "cc_cascade_save_xml": (_i, [_vp, C.c_char_p]),
    "cc_cascade_save_xml_legacy": (_i, [_vp, C.c_char_p]),
    "cc_vec_read": (_i, [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp, _i]),
    "cc_vec_write": (_i, [C.c_char_p, _vp, _i, _i, _i]),
    "cc_detector_create": (_i, [_vp, _i, _i, _pp]),
    "cc_detector_destroy": (None, [_vp]),
    "cc_detector_set_stream": (_i, [_vp, _vp]),
    "cc_detect_multiscale": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(DetectParams), _vp, _i, C.POINTER(_i)]),
    "cc_detect_batch": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _sz, C.POINTER(DetectParams), _vp, _i, _vp]),
    "cc_detect_batch_submit": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _sz, C.POINTER(DetectParams), _pp]),
    "cc_detect_batch_collect": (_i, [_vp, _vp, _vp, _i, _vp]),
    "cc_detect_batch_discard": (_i, [_vp, _vp]),
    "cc_detect_batch_device_only": (_i, [_vp, _vp, _i, _i, _i, _i, _sz, _sz, C.POINTER(DetectParams)]),
    "cc_detect_multiscale_levels": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(DetectParams), _vp, _vp, _vp, _i, C.POINTER(_i)]),
    "cc_detect_raw": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(DetectParams), _vp, _i, C.POINTER(_i)]),
    "cc_detect_debug_windows": (_i, [_vp, _vp, _i, _i, _sz, C.POINTER(DetectParams), _vp, _vp, _vp, C.c_int64,
                                     C.POINTER(C.c_int64)]),
    "cc_scale_plan": (_i, [_i, _i, _i, _i, C.POINTER(DetectParams), _vp, _i, C.POINTER(_i)]),
    "cc_detector_specialize": (_i, [_vp, _i]),
    "cc_detector_specialize_async": (_i, [_vp, _i]),
    "cc_detector_specialized_stages": (_i, [_vp]),
    "cc_cascade_compile_specialized": (_i, [_vp, _i, C.c_char_p, C.POINTER(C.c_size_t)]),
    "cc_detector_set_profiling": (_i, [_vp, _i]),
    "cc_detector_graph_active": (_i, [_vp]),
    "cc_detector_get_timings": (_i, [_vp, C.POINTER(DetectorTimings), _i]),
    "cc_integral_u8": (_i, [_i, _vp, _i, _i, _sz, _vp, _vp, _vp]),
    "cc_resize_linear_exact_u8": (_i, [_i, _vp, _i, _i, _sz, _vp, _i, _i, _sz]),
    "cc_debug_stream_dwords": (_i, [_i, _sz, _i, C.POINTER(C.c_uint32)]),
    "cc_debug_division_check": (_i, [_i, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cc_debug_vnf_check": (_i, [_i, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "cc_group_rectangles": (_i, [_vp, _i, _i, _d, _vp, _i, C.POINTER(_i)]),
    "cc_eval_create": (_i, [_i, _i, _i, _i, _i, _i, _pp]),
    "cc_eval_destroy": (None, [_vp]),
    "cc_eval_num_features": (_i, [_vp]),
    "cc_eval_max_cat_count": (_i, [_vp]),
    "cc_eval_feature_size": (_i, [_vp]),
    "cc_eval_feature_geometry": (_i, [_vp, _i, _vp, _vp, C.POINTER(_i)]),
    "cc_eval_set_image": (_i, [_vp, _vp, _sz, C.c_uint8, _i]),
    "cc_eval_set_images": (_i, [_vp, _vp, _i, _i, _vp]),
    "cc_eval_labels": (C.POINTER(C.c_float), [_vp]),
    "cc_eval_calc": (_i, [_vp, _i, _i, C.POINTER(C.c_float)]),
    "cc_eval_calc_list": (_i, [_vp, _vp, _i, _i, _vp]),
    "cc_eval_calc_batch": (_i, [_vp, _i, _i, _vp, _i, _vp, _i]),
    "cc_eval_calc_batch_device": (_i, [_vp, _i, _i, _vp, _i, _vp, _sz]),
    "cc_eval_calc_batch_sorted": (_i, [_vp, _i, _i, _i, _vp, _vp, _i]),
    "cc_eval_calc_custom_haar": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "cc_haar_feature_calc": (_i, [_i, _vp, _i, _i, _vp, _vp, _i, _i, _vp]),
    "cc_eval_get_sample": (_i, [_vp, _i, _vp, _vp, _vp]),
    "cc_eval_predict_cascade": (_i, [_vp, _vp, _vp, _i, _vp]),
    "cc_eval_last_kernel_ms": (_i, [_vp, C.POINTER(C.c_double)]),
    "cc_eval_presort": (_i, [_vp, _i]),
    "cc_eval_presort_range": (_i, [_vp, _i, _i, _i]),
    "cc_eval_find_best_split": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _d, _i, _i, C.POINTER(Split), _vp, _vp]),
    "cc_shard_range": (None, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "cc_comm_unique_id": (_i, [_vp]),
    "cc_comm_create": (_i, [_i, _i, _i, _vp, _pp]),
    "cc_comm_destroy": (None, [_vp]),
    "cc_comm_rank": (_i, [_vp]),
    "cc_comm_world": (_i, [_vp]),
    "cc_gather_detections": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "cc_gather_fetch": (_i, [_vp, _vp, _i, _vp, _i]),
    "cc_negminer_create": (_i, [_vp, _i, _pp]),
    "cc_negminer_destroy": (None, [_vp]),
    "cc_negminer_plan": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(_i), C.POINTER(C.c_int64)]),
    "cc_negminer_run": (_i, [_vp, _vp, _i, _i, _sz, _i, _i, _vp, C.c_int64, C.POINTER(C.c_int64), _vp, _vp, _i, C.POINTER(_i)]),
    "cc_negminer_run_batch": (_i, [_vp, _vp, _i, _i, _i, _sz, _i, _i, _vp, C.c_int64, C.POINTER(C.c_int64), _vp, _vp, _i, C.POINTER(_i)]),
}

_lib = None


def _share_torch_hip_runtime():
    """One process must hold ONE HIP runtime: PyTorch-ROCm wheels bundle their own libamdhip64.so.7, and a second copy
    (the system one our DT_NEEDED would pick) cannot open the GPU once the first has. Pre-load torch's copy (same
    SONAME) when a torch wheel is installed, so this library and a later `import torch` share it. Without torch the
    system runtime under /opt/rocm is used. CCAMD_HIP_RUNTIME=system forces the latter."""
    if os.environ.get("CCAMD_HIP_RUNTIME", "torch") == "system":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C cascadeclassifier_amd/csrc). There is no CPU fallback.")
        _share_torch_hip_runtime()
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status: int) -> int:
    if status != CC_OK:
        raise CascadeError(status, lib().cc_last_error().decode("utf-8", "replace"))
    return status
