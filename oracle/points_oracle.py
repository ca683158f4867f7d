"""ctypes wrapper of oracle/_build/libgcv_oracle.so (oracle/gcv_oracle.c) and loader of the reference's
own CPU extruder built into oracle/_ref/ (oracle/Makefile, target `ref`).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libgcv_oracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "footprint_extruder.so")
_lib = None
_ref = None


class SegIns(C.Structure):
    _fields_ = [(n, C.c_int16) for n in ("bldg_ins_min_id", "car_ins_min_id", "car_semantic_id",
                                         "bldg_facade_semantic_id", "roof_ins_offset")]


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    if os.path.exists("/root/reference/extensions/footprint_extruder/footprint_extruder.cpp"):
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"], stderr=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orv_extrude.restype = C.c_int64
        L.orv_extrude.argtypes = [C.c_int, C.c_void_p, C.POINTER(SegIns), C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orv_points_to_volume.restype = None
        L.orv_points_to_volume.argtypes = [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]
        L.orv_maps_to_volume.restype = C.c_int64
        L.orv_maps_to_volume.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
        L.orv_rvip.restype = None
        L.orv_rvip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def reference_extruder():
    """The reference's footprint_extruder module compiled from its own source (None if not built)."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        spec = importlib.util.spec_from_file_location("footprint_extruder", REF_PATH)
        _ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_ref)
    return _ref


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def scale_lut(classes_inv, scales):
    """scale_of_semantic[s] = scales[classes_inv[s]] (footprint_extruder.cpp:187-188); 0 = unknown."""
    lut = np.zeros(32768, np.int16)
    for sem, name in classes_inv.items():
        if 0 <= int(sem) < 32768 and name in scales:
            lut[int(sem)] = scales[name]
    return lut


def seg_ins_struct(m):
    return SegIns(m["BLDG_INS_MIN_ID"], m["CAR_INS_MIN_ID"], m["CAR_SEMANTIC_ID"], m["BLDG_FACADE_SEMANTIC_ID"],
                  m["ROOF_INS_OFFSET"])


def extrude(include_btm_pts, classes_inv, scales, seg_ins_map, seg_map, td_hf, bu_hf, pts_map):
    """Restatement of footprint_extruder.get_points_from_projection: uint16 [N,5] or None."""
    seg = np.ascontiguousarray(seg_map, np.int16)
    td = np.ascontiguousarray(td_hf, np.int16)
    bu = np.ascontiguousarray(bu_hf, np.int16)
    pts = np.ascontiguousarray(pts_map).astype(np.uint8)
    H, W = pts.shape
    lut = scale_lut(classes_inv, scales)
    m = seg_ins_struct(seg_ins_map)
    n = lib().orv_extrude(int(bool(include_btm_pts)), _p(lut), C.byref(m), H, W, _p(seg), _p(td), _p(bu), _p(pts),
                          None, 0)
    if n < 0:
        raise RuntimeError("pixel %d: semantic id without a positive scale" % (-1 - n))
    if n == 0:
        return None
    out = np.empty((n, 5), np.int16)
    lib().orv_extrude(int(bool(include_btm_pts)), _p(lut), C.byref(m), H, W, _p(seg), _p(td), _p(bu), _p(pts),
                      _p(out), n)
    return out.view(np.uint16)


def points_to_volume(points, pt_ids, scales, h, w, d):
    points = np.ascontiguousarray(points, np.int16)
    pt_ids = np.ascontiguousarray(pt_ids, np.int32).reshape(-1)
    scales = np.ascontiguousarray(scales, np.int16)
    vol = np.zeros((h, w, d), np.int32)
    lib().orv_points_to_volume(len(points), h, w, d, _p(points), _p(pt_ids), _p(scales), _p(vol))
    return vol


def maps_to_volume(inst_map, td_hf, bu_hf, pts_map, scales, depth=504):
    inst = np.ascontiguousarray(inst_map, np.int16)
    td, bu = np.ascontiguousarray(td_hf, np.int16), np.ascontiguousarray(bu_hf, np.int16)
    pts = np.ascontiguousarray(pts_map).astype(np.uint8)
    sc = np.ascontiguousarray(scales, np.int8)
    H, W = inst.shape
    vol = np.empty((H, W, depth), np.int16)
    rc = lib().orv_maps_to_volume(H, W, depth, _p(sc), len(sc), _p(inst), _p(td), _p(bu), _p(pts), _p(vol))
    if rc < 0:
        raise RuntimeError("pixel %d: class without a positive scale" % (-2 - rc))
    return vol


def ray_voxel_intersection_perspective(in_voxel, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples):
    vol = np.asarray(in_voxel)
    assert vol.dtype == np.int32 and vol.ndim == 3
    dims = np.array(vol.shape, np.int32)
    strides = np.array([s // 4 for s in vol.strides], np.int64)
    ori = np.ascontiguousarray(cam_ori, np.float32)
    dr = np.ascontiguousarray(cam_dir, np.float32)
    up = np.ascontiguousarray(cam_up, np.float32)
    c = np.ascontiguousarray(cam_c, np.float32)
    img = np.array(img_dims, np.int32)
    H, W = int(img[0]), int(img[1])
    vid = np.empty((H, W, max_samples, 1), np.int32)
    dep = np.empty((2, H, W, max_samples, 1), np.float32)
    rd = np.empty((H, W, 1, 3), np.float32)
    lib().orv_rvip(_p(vol), _p(dims), _p(strides), _p(ori), _p(dr), _p(up), float(cam_f), _p(c), _p(img),
                   int(max_samples), _p(vid), _p(dep), _p(rd))
    return vid, dep, rd
