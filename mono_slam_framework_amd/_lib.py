"""ctypes view of include/msf_abi.h.  Loading fails loudly when libmsf.so is missing: there is no CPU fallback."""
import ctypes as C
import os

import numpy as np

from . import build as _build

MSF_OK = 0
MSF_ERR_INVALID_ARG = -1
MSF_ERR_HIP = -2
MSF_ERR_UNSUPPORTED = -3
MSF_ERR_CAPACITY = -4
MSF_ERR_IO = -5

MSF_KIND_ORB = 0
MSF_KIND_LOFTR = 1

MSF_FLAG_BLUR_TIE_HALF_UP = 1
MSF_FLAG_PROFILE = 2
MSF_FLAG_KEEP_DEBUG = 4
MSF_FLAG_FAST_DENSE = 8
MSF_FLAG_NO_FRAME_CACHE = 16
MSF_FLAG_LEVEL_SIZE_MUL_INV = 32
MSF_FLAG_FAST_STREAM = 64
MSF_FLAG_LOFTR_F32 = 128
MSF_FLAG_BLUR_SUM256 = 256

(DBG_LEVEL_SIZES, DBG_LEVEL_PIXELS, DBG_FAST_CANDS, DBG_KEYPOINTS, DBG_DESCRIPTORS, DBG_STAGE1,
 DBG_LOFTR_CONF, DBG_LOFTR_FEAT, DBG_FAST_TAU, DBG_LOFTR_ACT, DBG_WALK_MODE) = range(11)

# every symbol include/msf_abi.h declares
ABI_SYMBOLS = ["msf_abi_version", "msf_default_config", "msf_create", "msf_destroy", "msf_set_threshold",
               "msf_last_error", "msf_match_pair", "msf_match_batch", "msf_match_batch_device",
               "msf_extract_device", "msf_match_slots_device", "msf_pack_matches_device", "msf_debug_get",
               "msf_stage_times", "msf_set_mappoints", "msf_count_mappoint_matches_device",
               "msf_store_frame", "msf_match_one_to_many", "msf_check_hypotheses",
               "msf_render_match_image", "msf_weights_info", "msf_convert_weights",
               "msf_frame_cache_stats", "msf_multi_create", "msf_multi_destroy", "msf_multi_device_count",
               "msf_multi_handle", "msf_multi_set_threshold", "msf_multi_last_error", "msf_multi_shard_range",
               "msf_multi_match_batch", "msf_multi_match_batch_device",
               "msf_gather_unique_id", "msf_gather_create", "msf_gather_destroy", "msf_gather_last_error",
               "msf_gather_plan", "msf_gather_matches_device"]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kind", C.c_int32), ("device", C.c_int32), ("threshold", C.c_float),
                ("image_width", C.c_int32), ("image_height", C.c_int32), ("max_batch_pairs", C.c_int32),
                ("flags", C.c_uint32), ("weights_path", C.c_char_p)]


class Image(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int64)]


MATCH_DTYPE = np.dtype([("x1", "<i4"), ("y1", "<i4"), ("x2", "<i4"), ("y2", "<i4")])
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<f4"), ("angle", "<f4"),
                     ("octave", "<i4"), ("lx", "<i4"), ("ly", "<i4"), ("fast_score", "<i4")])

_lib = None


def load():
    """Loads libmsf.so (after torch, if torch is importable, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (torch bundles its own libamdhip64.so.7; load it first)
    except Exception:  # pragma: no cover
        pass
    path = _build.lib_path()
    L = C.CDLL(path, mode=C.RTLD_GLOBAL if hasattr(C, "RTLD_GLOBAL") else 0)
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    L.msf_abi_version.restype = C.c_int
    L.msf_default_config.argtypes = [C.POINTER(Config), C.c_int]
    L.msf_default_config.restype = None
    L.msf_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.msf_destroy.argtypes = [vp]
    L.msf_destroy.restype = None
    L.msf_set_threshold.argtypes = [vp, f32]
    L.msf_last_error.argtypes = [vp]
    L.msf_last_error.restype = C.c_char_p
    L.msf_match_pair.argtypes = [vp, C.POINTER(Image), C.POINTER(Image), vp, i32, C.POINTER(i32)]
    L.msf_match_batch.argtypes = [vp, i32, C.POINTER(Image), C.POINTER(Image), vp, i32, vp]
    L.msf_match_batch_device.argtypes = [vp, i32, vp, vp, i64, i64, vp, i32, vp, vp]
    L.msf_extract_device.argtypes = [vp, i32, vp, i64, i64, i32, vp]
    L.msf_match_slots_device.argtypes = [vp, i32, vp, vp, vp, i32, vp, vp]
    L.msf_pack_matches_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp]
    L.msf_debug_get.argtypes = [vp, i32, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.msf_set_mappoints.argtypes = [vp, i32, vp, i32]
    L.msf_count_mappoint_matches_device.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.msf_store_frame.argtypes = [vp, i32, C.POINTER(Image)]
    L.msf_match_one_to_many.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32]
    L.msf_check_hypotheses.argtypes = [vp, i32, i32, vp, vp, i32, vp, f32, vp, C.POINTER(i32), vp]
    L.msf_render_match_image.argtypes = [vp, C.POINTER(Image), C.POINTER(Image), vp, i32, vp, vp, vp, i64]
    L.msf_weights_info.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(i32), C.POINTER(i64)]
    L.msf_convert_weights.argtypes = [C.c_char_p, C.c_char_p]
    L.msf_frame_cache_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(i32)]
    L.msf_stage_times.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(f32), i32]
    L.msf_multi_create.argtypes = [C.POINTER(Config), i32, C.POINTER(i32), C.POINTER(vp)]
    L.msf_multi_destroy.argtypes = [vp]
    L.msf_multi_destroy.restype = None
    L.msf_multi_device_count.argtypes = [vp]
    L.msf_multi_device_count.restype = i32
    L.msf_multi_handle.argtypes = [vp, i32]
    L.msf_multi_handle.restype = vp
    L.msf_multi_set_threshold.argtypes = [vp, f32]
    L.msf_multi_last_error.argtypes = [vp]
    L.msf_multi_last_error.restype = C.c_char_p
    L.msf_multi_shard_range.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.msf_multi_shard_range.restype = None
    L.msf_multi_match_batch.argtypes = [vp, i32, C.POINTER(Image), C.POINTER(Image), vp, i32, vp]
    L.msf_multi_match_batch_device.argtypes = [vp, C.POINTER(i32), C.POINTER(vp), C.POINTER(vp), i64, i64, C.POINTER(vp), i32,
                                               C.POINTER(vp)]
    L.msf_gather_unique_id.argtypes = [vp]
    L.msf_gather_create.argtypes = [i32, i32, i32, vp, i32, i64, C.POINTER(vp)]
    L.msf_gather_destroy.argtypes = [vp]
    L.msf_gather_destroy.restype = None
    L.msf_gather_last_error.argtypes = [vp]
    L.msf_gather_last_error.restype = C.c_char_p
    L.msf_gather_plan.argtypes = [i32, i32, vp, i64, vp, vp]
    L.msf_gather_matches_device.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    _lib = L
    return L


def default_weights_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "weights", "loftr_teacher.bin")
