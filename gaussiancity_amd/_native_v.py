"""ctypes binding of libgcv_hip.so (C ABI in include/gcv.h): point generation / visibility path.

The library is the product: there is NO Python/CPU fallback -- a missing library or a failing call
raises RuntimeError.
"""
import ctypes as C
import os

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libgcv_hip.so")

EXPORTED_SYMBOLS = (
    "gcv_abi_version", "gcv_last_error", "gcv_extrude_scratch_bytes", "gcv_extrude_count", "gcv_extrude_emit",
    "gcv_maps_to_volume", "gcv_occupancy_bytes", "gcv_points_to_volume", "gcv_build_occupancy", "gcv_bounds_scratch_bytes", "gcv_points_bounds", "gcv_rows_to_volume", "gcv_rows_erase_volume",
    "gcv_ray_voxel_intersection",
    "gcv_set_option", "gcv_get_stage_ms",
)
STAGE_NAMES = ("extrude_count", "extrude_emit", "volume_clear", "volume_scatter", "occupancy", "traversal")
ABI_VERSION = 4


class SegIns(C.Structure):
    """gcv_seg_ins == segInsMap (footprint_extruder.cpp:90-100,201)."""
    _fields_ = [(n, C.c_int16) for n in ("bldg_ins_min_id", "car_ins_min_id", "car_semantic_id",
                                         "bldg_facade_semantic_id", "roof_ins_offset")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libgcv_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C gaussiancity_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t
    L.gcv_abi_version.restype = C.c_int
    L.gcv_last_error.restype = C.c_char_p
    L.gcv_extrude_scratch_bytes.restype = sz
    L.gcv_extrude_scratch_bytes.argtypes = [i32, i32]
    L.gcv_extrude_count.restype = C.c_int
    L.gcv_extrude_count.argtypes = [i32, vp, C.POINTER(SegIns), i32, i32, vp, vp, vp, vp, vp, sz, C.POINTER(i64), vp]
    L.gcv_extrude_emit.restype = C.c_int
    L.gcv_extrude_emit.argtypes = [i32, vp, C.POINTER(SegIns), i32, i32, vp, vp, vp, vp, vp, sz, vp, i64, vp]
    L.gcv_maps_to_volume.restype = C.c_int
    L.gcv_maps_to_volume.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.gcv_occupancy_bytes.restype = sz
    L.gcv_occupancy_bytes.argtypes = [i32, i32, i32]
    L.gcv_points_to_volume.restype = C.c_int
    L.gcv_points_to_volume.argtypes = [i64, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.gcv_bounds_scratch_bytes.restype = sz
    L.gcv_bounds_scratch_bytes.argtypes = []
    L.gcv_points_bounds.restype = C.c_int
    L.gcv_points_bounds.argtypes = [i64, vp, i32, vp, C.POINTER(i32), C.POINTER(i32), vp]
    L.gcv_rows_to_volume.restype = C.c_int
    L.gcv_rows_to_volume.argtypes = [i64, vp, C.POINTER(i32), i32, i32, i32, vp, vp, i32, vp]
    L.gcv_rows_erase_volume.restype = C.c_int
    L.gcv_rows_erase_volume.argtypes = [i64, vp, C.POINTER(i32), i32, i32, i32, vp, vp]
    L.gcv_build_occupancy.restype = C.c_int
    L.gcv_build_occupancy.argtypes = [vp, i32, i32, i32, vp, vp]
    L.gcv_ray_voxel_intersection.restype = C.c_int
    L.gcv_ray_voxel_intersection.argtypes = [vp, C.POINTER(i32), C.POINTER(i64), vp, C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                             C.POINTER(C.c_float), C.POINTER(i32), i32, vp, vp, vp, vp]
    L.gcv_set_option.restype = C.c_int
    L.gcv_set_option.argtypes = [C.c_char_p, C.c_int]
    L.gcv_get_stage_ms.restype = C.c_int
    L.gcv_get_stage_ms.argtypes = [C.POINTER(C.c_float), C.c_int]
    if L.gcv_abi_version() != ABI_VERSION:
        raise RuntimeError("libgcv_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc, what):
    if rc < 0:
        raise RuntimeError("%s failed (gcv_status %d): %s" % (what, rc, lib().gcv_last_error().decode("utf-8", "replace")))
    return rc


def set_option(name, value):
    return lib().gcv_set_option(name.encode(), int(value))


def stage_ms():
    buf = (C.c_float * len(STAGE_NAMES))()
    n = lib().gcv_get_stage_ms(buf, len(STAGE_NAMES))
    return {STAGE_NAMES[i]: float(buf[i]) for i in range(n)}
