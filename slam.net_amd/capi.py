"""ctypes binding of include/slamhip.h -- one Python function per C-ABI entry point, nothing else.

Loading fails loudly when libslamhip.so is missing (run ``python -m slam.net_amd.build`` or
``__graft_entry__.build()``); there is no CPU fallback in this package.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libslamhip.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "slamhip.h")

OK = 0
ERR_INVALID, ERR_HIP, ERR_NOMEM, ERR_STATE, ERR_RCCL, ERR_TIMEOUT = -1, -2, -3, -4, -5, -6
K_CS_PREP, K_CS_DISTANCE, K_CS_REDUCE, K_CS_HOLEMAP, K_CS_OBSTACLE, K_HS_MATCH, K_HS_UPDATE = range(7)

CELL_DTYPE = np.dtype([("update_index", np.int32), ("value", np.float32)])


class SlamhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slamhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def declared_symbols():
    """Every function name declared in include/slamhip.h."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slamhip_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("SLAMHIP_LIB") or SO_PATH            # (developer aid: an alternative build of the same ABI, for A/B timing on one box)
        if not os.path.exists(so):
            raise ImportError("libslamhip.so not built at %s -- build it with hipcc (python -m slam.net_amd.build); "
                              "slam.net_amd has no CPU fallback" % so)
        L = C.CDLL(so)
        _declare(L)
        _lib = L
    return _lib


def _declare(L):
    i32, u64, i64, f, vp, sz = C.c_int32, C.c_uint64, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
    P = C.POINTER
    fp, ip, u16p, i8p, u8p, u64p, vpp = P(f), P(i32), P(C.c_uint16), P(C.c_int8), P(C.c_uint8), P(u64), P(vp)
    sig = {
        "slamhip_version": (C.c_char_p, []),
        "slamhip_last_error": (C.c_char_p, []),
        "slamhip_device_count": (i32, [ip]),
        "slamhip_ctx_create": (i32, [i32, vpp]),
        "slamhip_ctx_destroy": (i32, [vp]),
        "slamhip_ctx_synchronize": (i32, [vp]),
        "slamhip_ctx_device": (i32, [vp, ip]),
        "slamhip_ctx_stream": (vp, [vp]),
        "slamhip_ctx_set_wait_timeout": (i32, [vp, i64]),
        "slamhip_ctx_poisoned": (i32, [vp, ip]),
        "slamhip_ctx_philox4x32_10": (i32, [vp, P(C.c_uint32), P(C.c_uint32), P(C.c_uint32)]),
        "slamhip_debug_flag_wait": (i32, [P(C.c_uint32), C.c_uint32, i64]),
        "slamhip_ctx_timing_enable": (i32, [vp, i32]),
        "slamhip_ctx_timing_reset": (i32, [vp]),
        "slamhip_ctx_timing_get": (i32, [vp, i32, P(C.c_double), P(i64)]),
        "slamhip_cs_create": (i32, [vp, f, i32, i32, vpp]),
        "slamhip_cs_destroy": (i32, [vp]),
        "slamhip_cs_info": (i32, [vp, ip, fp, ip, fp]),
        "slamhip_cs_reset": (i32, [vp, i32]),
        "slamhip_cs_holemap_upload": (i32, [vp, u16p, sz]),
        "slamhip_cs_holemap_download": (i32, [vp, u16p, sz]),
        "slamhip_cs_holemap_download_packed": (i32, [vp, u8p, sz]),
        "slamhip_cs_holemap_mirror": (i32, [vp, u16p, sz, ip]),
        "slamhip_cs_holemap_mirror_async": (i32, [vp, u16p, sz]),
        "slamhip_cs_holemap_mirror_wait": (i32, [vp, ip, P(i64)]),
        "slamhip_cs_holemap_mirror_release": (i32, [vp]),
        "slamhip_cs_obstaclemap_upload": (i32, [vp, i8p, sz]),
        "slamhip_cs_obstaclemap_download": (i32, [vp, i8p, sz]),
        "slamhip_cs_set_scan": (i32, [vp, fp, i32]),
        "slamhip_cs_distance_pxcs": (i32, [vp, fp, i32, ip, ip, ip]),
        "slamhip_cs_distance_poses": (i32, [vp, fp, i32, ip, ip, ip]),
        "slamhip_cs_set_offsets": (i32, [vp, fp, i32]),
        "slamhip_cs_generate_offsets": (i32, [vp, i32, f, f, u64, u64]),
        "slamhip_cs_generate_offsets_lattice": (i32, [vp, i32, f, f, u64, u64]),
        "slamhip_cs_offsets_download": (i32, [vp, fp, i32]),
        "slamhip_cs_search": (i32, [vp, fp, fp, ip, ip]),
        "slamhip_cs_search_shard": (i32, [vp, fp, i32, i32, u64p]),
        "slamhip_cs_search_shard_async": (i32, [vp, fp, i32, i32, vp]),
        "slamhip_cs_search_shard_enqueue": (i32, [vp, fp, i32, i32, C.POINTER(C.c_void_p)]),
        "slamhip_cs_key_read": (i32, [vp, vp, u64p]),
        "slamhip_cs_pose_from_key": (i32, [vp, fp, u64, fp, ip, ip]),
        "slamhip_cs_update_holemap": (i32, [vp, fp, f, i32]),
        "slamhip_cs_update_holemap_pxcs": (i32, [vp, fp, f, i32]),
        "slamhip_cs_update_obstaclemap": (i32, [vp, fp, i32]),
        "slamhip_cs_update_obstaclemap_pxcs": (i32, [vp, fp, i32]),
        "slamhip_cs_last_holemap_pixels": (i32, [vp, P(i64)]),
        "slamhip_cs_maps_checksum": (i32, [vp, P(u64)]),
        "slamhip_cs_search_and_update": (i32, [vp, fp, f, i32, i32, fp, ip, ip]),
        "slamhip_cs_scan_search_and_update": (i32, [vp, fp, i32, fp, f, i32, i32, fp, ip, ip]),
        "slamhip_cs_search_and_update_pxcs": (i32, [vp, fp, fp, fp, i32, f, i32, i32, ip, ip]),
        "slamhip_cs_update_maps_pxcs": (i32, [vp, fp, fp, f, i32, i32]),
        "slamhip_cs_selfcheck_failures": (i32, [vp, P(C.c_uint32)]),
        "slamhip_cs_prelaunch_stats": (i32, [vp, P(C.c_uint64)]),
        "slamhip_cs_plan_stats": (i32, [vp, P(C.c_uint64)]),
        "slamhip_cs_prepared_lists": (i32, [vp, P(C.c_uint64), P(C.c_uint64)]),
        "slamhip_csproc_create": (i32, [vp, f, i32, i32, fp, f, f, i32, i32, vpp]),
        "slamhip_csproc_destroy": (i32, [vp]),
        "slamhip_csproc_reset": (i32, [vp]),
        "slamhip_csproc_update": (i32, [vp, fp, ip, i32, fp]),
        "slamhip_csproc_get_pose": (i32, [vp, fp]),
        "slamhip_csproc_set_params": (i32, [vp, i32, f, i32, i32, i32]),
        "slamhip_csproc_set_seed": (i32, [vp, u64]),
        "slamhip_csproc_set_lattice": (i32, [vp, i32]),
        "slamhip_csproc_set_offsets": (i32, [vp, fp, i32]),
        "slamhip_csproc_cs": (i32, [vp, vpp]),
        "slamhip_scan_segments_to_cloud": (i32, [fp, ip, i32, fp, fp]),
        "slamhip_hs_create": (i32, [vp, f, i32, i32, i32, vpp]),
        "slamhip_hs_destroy": (i32, [vp]),
        "slamhip_hs_reset": (i32, [vp]),
        "slamhip_hs_level_info": (i32, [vp, i32, ip, ip, fp]),
        "slamhip_hs_set_factors": (i32, [vp, f, f]),
        "slamhip_hs_set_iterations": (i32, [vp, ip]),
        "slamhip_hs_cells_upload": (i32, [vp, i32, vp, sz]),
        "slamhip_hs_cells_download": (i32, [vp, i32, vp, sz]),
        "slamhip_hs_bitmap_download": (i32, [vp, i32, u8p, sz]),
        "slamhip_hs_map_extends": (i32, [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "slamhip_hs_checksum": (i32, [vp, i32, P(u64)]),
        "slamhip_hs_probability": (i32, [vp, i32, ip, i32, fp]),
        "slamhip_hs_set_scan": (i32, [vp, fp, i32, fp]),
        "slamhip_hs_match": (i32, [vp, fp, fp]),
        "slamhip_hs_match_level": (i32, [vp, i32, fp, i32, fp]),
        "slamhip_hs_match_batch": (i32, [vp, fp, i32, fp]),
        "slamhip_hs_hessian": (i32, [vp, i32, fp, fp, fp]),
        "slamhip_hs_update_by_scan": (i32, [vp, fp]),
        "slamhip_hsproc_create": (i32, [vp, f, i32, i32, fp, i32, vpp]),
        "slamhip_hsproc_destroy": (i32, [vp]),
        "slamhip_hsproc_reset": (i32, [vp]),
        "slamhip_hsproc_update": (i32, [vp, fp, i32, fp, fp, i32, ip]),
        "slamhip_hsproc_get": (i32, [vp, fp, fp, fp, fp]),
        "slamhip_hsproc_set_thresholds": (i32, [vp, f, f]),
        "slamhip_hsproc_hs": (i32, [vp, vpp]),
        "slamhip_group_create": (i32, [ip, i32, f, i32, i32, vpp]),
        "slamhip_group_destroy": (i32, [vp]),
        "slamhip_group_size": (i32, [vp, ip]),
        "slamhip_group_cs": (i32, [vp, i32, vpp]),
        "slamhip_group_reset": (i32, [vp, i32]),
        "slamhip_group_holemap_upload": (i32, [vp, u16p, sz]),
        "slamhip_group_set_scan": (i32, [vp, fp, i32]),
        "slamhip_group_set_offsets": (i32, [vp, fp, i32]),
        "slamhip_group_generate_offsets": (i32, [vp, i32, f, f, u64, u64]),
        "slamhip_group_search": (i32, [vp, fp, fp, ip, ip]),
        "slamhip_group_update_maps": (i32, [vp, fp, f, i32, i32]),
        "slamhip_group_search_and_update": (i32, [vp, fp, f, i32, i32, fp, ip, ip]),
        "slamhip_group_replicas_equal": (i32, [vp, P(i32)]),
        "slamhip_comm_probe": (i32, []),
        "slamhip_comm_unique_id": (i32, [u8p]),
        "slamhip_comm_create": (i32, [vp, u8p, i32, i32, vpp]),
        "slamhip_comm_destroy": (i32, [vp]),
        "slamhip_comm_info": (i32, [vp, ip, ip]),
        "slamhip_cs_search_allreduce_async": (i32, [vp, vp, fp, i32, i32, C.POINTER(C.c_void_p)]),
        "slamhip_comm_wait": (i32, [vp, C.POINTER(C.c_uint64)]),
        "slamhip_comm_set_batch": (i32, [vp, i32]),
        "slamhip_cs_search_allreduce": (i32, [vp, vp, fp, i32, i32, u64p]),
        "slamhip_cs_search_allreduce_and_update": (i32, [vp, vp, fp, i32, i32, f, i32, i32, fp, ip, ip]),
        "slamhip_comm_allreduce_probe": (i32, [vp, i32, fp]),
        "slamhip_comm_replicas_equal": (i32, [vp, vp, P(i32)]),
    }
    for name, (res, args) in sig.items():
        if os.environ.get("SLAMHIP_LIB") and not hasattr(L, name):
            continue                                               # (developer aid: an OLDER build of the library timed beside the current one)
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig


def check(rc):
    if rc != OK:
        raise SlamhipError(rc, lib().slamhip_last_error().decode(errors="replace"))


def call(name, *args):
    check(getattr(lib(), name)(*args))


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)
