"""ctypes loader of libpandora_amd.so - the C ABI declared in include/pandora_amd.h.

Fails loudly: there is no CPU fallback anywhere in this package.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpandora_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pandora_amd.h")
_LIB = None

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_i16_p = C.POINTER(C.c_int16)
c_i64_p = C.POINTER(C.c_int64)
c_int_p = C.POINTER(C.c_int)
vp = C.c_void_p

# name -> (restype, argtypes); must list every function of include/pandora_amd.h
SIGNATURES = {
    "pmx_last_error": (C.c_char_p, []),
    "pmx_device_count": (C.c_int, []),
    "pmx_create": (vp, [C.c_int]),
    "pmx_destroy": (None, [vp]),
    "pmx_sync": (C.c_int, [vp]),
    "pmx_set_images": (C.c_int, [vp, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int]),
    "pmx_set_images_fingerprinted": (C.c_int, [vp, c_float_p, c_float_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint64)]),
    "pmx_swap_images": (C.c_int, [vp]),
    "pmx_set_masks": (C.c_int, [vp, c_i16_p, c_i16_p, C.c_int, C.c_int]),
    "pmx_set_shifted_right": (C.c_int, [vp, C.c_int, C.POINTER(C.c_float)]),
    "pmx_set_disparity_grids": (C.c_int, [vp, c_double_p, c_double_p]),
    "pmx_set_lazy": (C.c_int, [vp, C.c_int]),
    "pmx_set_option": (C.c_int, [vp, C.c_char_p, C.c_char_p]),
    "pmx_get_option": (C.c_char_p, [vp, C.c_char_p]),
    "pmx_option_name": (C.c_char_p, [C.c_int]),
    "pmx_cv_alloc": (vp, [vp, C.c_int, C.c_int]),
    "pmx_cv_free": (None, [vp, vp]),
    "pmx_cv_fill_nan": (C.c_int, [vp, vp]),
    "pmx_cv_upload": (C.c_int, [vp, vp, c_float_p]),
    "pmx_cv_download": (C.c_int, [vp, vp, c_float_p]),
    "pmx_cv_download_rows": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "pmx_cv_dims": (C.c_int, [vp, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p]),
    "pmx_census": (C.c_int, [vp, vp, C.c_int]),
    "pmx_sad_ssd": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "pmx_zncc": (C.c_int, [vp, vp, C.c_int]),
    "pmx_cv_masked": (C.c_int, [vp, vp, C.c_int]),
    "pmx_nan_pixels": (C.c_int, [vp, vp, C.POINTER(C.c_uint8)]),
    "pmx_order_statistics": (C.c_int, [vp, c_float_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, c_float_p]),
    "pmx_host_minmax_i64": (C.c_int, [c_i64_p, C.c_size_t, c_i64_p, c_i64_p]),
    "pmx_host_fingerprint": (C.c_uint64, [vp, C.c_size_t]),
    "pmx_cv_mark_missing": (C.c_int, [vp, vp]),
    "pmx_cv_get_missing": (C.c_int, [vp, vp, C.POINTER(C.c_uint8)]),
    "pmx_compose_validity": (C.c_int, [vp, c_i64_p, C.c_int, vp, C.c_int]),
    "pmx_reverse_cost_volume": (vp, [vp, vp, C.c_int]),
    "pmx_cbca": (C.c_int, [vp, vp, C.c_int, C.c_float, C.c_int]),
    "pmx_cross_support": (C.c_int, [vp, C.c_int, C.c_int, C.c_float, C.c_int, c_i16_p]),
    "pmx_sgm": (C.c_int, [vp, vp, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int]),
    "pmx_sgm_p2maps": (C.c_int, [vp, vp, C.c_float, C.POINTER(C.c_float), C.c_int, C.c_float, C.c_int]),
    "pmx_debug_sgm_directions": (C.c_int, [vp, C.c_int]),
    "pmx_set_validity": (C.c_int, [vp, c_i64_p]),
    "pmx_map_snapshot": (vp, [vp, C.c_int]),
    "pmx_map_snapshot_alloc": (vp, [vp, C.c_int]),
    "pmx_maps_restore": (C.c_int, [vp, vp, vp]),
    "pmx_median_filter_maps": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "pmx_cross_checking_maps": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, vp]),
    "pmx_validity_frame_map": (C.c_int, [vp, vp, C.c_int]),
    "pmx_map_snapshot_read": (C.c_int, [vp, vp, vp]),
    "pmx_map_snapshot_free": (None, [vp, vp]),
    "pmx_wta": (C.c_int, [vp, vp, C.c_int, C.c_float]),
    "pmx_refine": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "pmx_refine_approximate": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "pmx_get_disparity": (C.c_int, [vp, c_float_p, c_i64_p, c_float_p]),
    "pmx_set_disparity": (C.c_int, [vp, c_float_p, c_i64_p]),
    "pmx_wta_minkey": (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    "pmx_wta_from_keys": (C.c_int, [vp, vp, C.c_double, C.c_int, C.c_float]),
    "pmx_cross_checking": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_double, C.POINTER(C.c_float)]),
    "pmx_reverse_disp_range": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pmx_cv_scale_pixels": (C.c_int, [vp, vp, C.POINTER(C.c_float)]),
    "pmx_interpolate_disparity": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int, C.c_int, c_int_p, C.c_int]),
    "pmx_median_filter_disparity": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int]),
    "pmx_denoise_disparity": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]),
    "pmx_bilateral_filter_disparity": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_double,
                                                 C.c_double]),
    "pmx_interpolate_nodata": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "pmx_disparity_range": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pmx_ambiguity": (C.c_int, [vp, vp, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int,
                                C.POINTER(C.c_float)]),
    "pmx_risk": (C.c_int, [vp, vp, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int,
                           C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pmx_interval_bounds": (C.c_int, [vp, vp, C.c_float, C.c_float, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_float),
                                      C.POINTER(C.c_float)]),
    "pmx_debug_path_costs": (C.c_int, [vp, vp, C.POINTER(C.c_uint8), C.c_size_t, c_int_p, c_int_p, c_int_p]),
    "pmx_debug_small_division": (C.c_int, [vp, C.POINTER(C.c_uint)]),
    "pmx_debug_fam_windows": (C.c_int, [vp, C.POINTER(C.c_uint), C.c_int]),
    "pmx_set_placement_trials": (C.c_int, [vp, C.c_int]),
    "pmx_measure_hbm": (C.c_int, [vp, C.c_size_t, c_double_p, c_double_p, c_double_p]),
    "pmx_release_caches": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "pmx_set_profiling": (C.c_int, [vp, C.c_int]),
    "pmx_reset_stage_times": (C.c_int, [vp]),
    "pmx_stage_time": (C.c_int, [vp, C.c_int, c_double_p, c_int_p]),
    "pmx_stream": (vp, [vp]),
    "pmx_host_alloc": (vp, [C.c_size_t]),
    "pmx_host_free": (None, [vp]),
    "pmx_cross_support_image": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    "pmx_cbca_slice": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp]),
    "pmx_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
    "pmx_comm_init": (C.c_int, [vp, vp, C.c_size_t, C.c_int, C.c_int]),
    "pmx_comm_destroy": (C.c_int, [vp]),
    "pmx_comm_info": (C.c_int, [vp, c_int_p, c_int_p]),
    "pmx_comm_count": (C.c_int, [vp, c_int_p]),
    "pmx_comm_allreduce": (C.c_int, [vp, C.c_int, C.c_int]),
    "pmx_comm_allreduce_scalars": (C.c_int, [vp, c_double_p, C.c_int]),
    "pmx_comm_allgather_rows": (C.c_int, [vp, C.c_int]),
    "pmx_comm_gather_rows": (C.c_int, [vp, C.c_int, C.c_int]),
    "pmx_shard_minkey": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "pmx_shard_from_keys": (C.c_int, [vp, C.c_double, C.c_int, C.c_float]),
    "pmx_shard_nan_pixels": (C.c_int, [vp, vp]),
    "pmx_shard_refine_pack": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "pmx_shard_refine_unpack": (C.c_int, [vp]),
    "pmx_tile_place": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "pmx_get_full_maps": (C.c_int, [vp, vp, vp, vp]),
    "pmx_set_full_rows": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "pmx_xbuf_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_size_t), c_int_p]),
    "pmx_xbuf_download": (C.c_int, [vp, C.c_int, vp]),
    "pmx_xbuf_upload": (C.c_int, [vp, C.c_int, vp]),
}

XBUFS = {"keys": (0, "uint64"), "nanpix": (1, "uint8"), "refine_pack": (2, "float32"), "refine_flags": (3, "int64"),
         "full_disp": (4, "float32"), "full_validity": (5, "int64"), "full_itp": (6, "float32"), "scalars": (7, "float64"), "full_validity16": (8, "uint16")}

STAGES = {
    "census_transform": 0, "census_cost": 1, "sad_ssd": 2, "zncc": 3, "mask": 4, "cbca_arms": 5, "cbca_h": 6,
    "cbca_v": 7, "sgm_path": 8, "sgm_final": 9, "wta": 10, "refine": 11, "reverse": 12, "minkey": 13, "sgm_fused": 14, "sgm_family": 15, "collective": 16, "sgm_span": 17,
}


def header_symbols():
    """Every pmx_* function name declared in include/pandora_amd.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pmx_[a-z_0-9]+)\s*\(", text)))


def lib():
    """Load libpandora_amd.so.  Raises RuntimeError (never falls back) when it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C pandora_amd/csrc`). pandora_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


class PmxError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = lib().pmx_last_error()
        raise PmxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
