"""Host mirror of the reference's point-generation / visibility interfaces over libgcv_hip.so
(SURVEY.md section 8 row f2).  Same names, argument meaning and error behaviour as upstream:

  get_points_from_projection   footprint_extruder.get_points_from_projection
                               (extensions/footprint_extruder/footprint_extruder.cpp:143-213)
  points_to_volume             extensions.voxlib.points_to_volume        (voxlib/bindings.cpp:36)
  ray_voxel_intersection_perspective
                               extensions.voxlib.ray_voxel_intersection_perspective (bindings.cpp:33)
  get_visible_points           scripts/dataset_generator.py:1414-1461 (the caller of the two above)

torch supplies device memory and the current stream; every computation happens in the HIP library.
There is no CPU path: without a GPU (or without the built library) these raise RuntimeError.
"""
import ctypes as C

import numpy as np
import scipy.spatial.transform
import torch

from . import _native_v as V


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(dev=None):
    if not torch.cuda.is_available():
        raise RuntimeError("gaussiancity_amd.points needs a ROCm GPU: the product has no CPU path")
    return torch.device("cuda", torch.cuda.current_device()) if dev is None else dev


def _scale_lut(classes, scales, device):
    """scale_of_semantic[s] = scales[classes[s]] (footprint_extruder.cpp:187-188); 0 marks an unknown id."""
    lut = np.zeros(32768, np.int16)
    for sem, name in classes.items():
        if 0 <= int(sem) < 32768 and name in scales:
            lut[int(sem)] = int(scales[name])
    return torch.from_numpy(lut).to(device)


def _seg_ins(m):
    # std::map::at throws std::out_of_range upstream for a missing key (footprint_extruder.cpp:92-99)
    try:
        return V.SegIns(int(m["BLDG_INS_MIN_ID"]), int(m["CAR_INS_MIN_ID"]), int(m["CAR_SEMANTIC_ID"]),
                        int(m["BLDG_FACADE_SEMANTIC_ID"]), int(m["ROOF_INS_OFFSET"]))
    except KeyError as e:
        raise IndexError("map::at: missing key %s in seg_ins_map" % e)


def extrude_points(include_btm_pts, classes, scales, seg_ins_map, seg_map, td_hf, bu_hf, pts_map):
    """Device-resident form: int16 [H,W] maps + bool/uint8 [H,W] point map as CUDA tensors ->
    int16 CUDA tensor [N,5] = (x, y, z, scale, instanceID) in upstream's order (N may be 0)."""
    dev = seg_map.device
    if dev.type != "cuda":
        raise RuntimeError("extrude_points expects CUDA tensors (the product has no CPU path)")
    for t, dt in ((seg_map, torch.int16), (td_hf, torch.int16), (bu_hf, torch.int16)):
        if t.dtype != dt or t.dim() != 2 or t.shape != seg_map.shape:
            raise TypeError("seg_map / td_hf / bu_hf must be int16 [H,W] of one shape")
    if pts_map.shape != seg_map.shape or pts_map.dtype not in (torch.bool, torch.uint8):
        raise TypeError("pts_map must be bool [H,W]")
    H, W = int(seg_map.shape[0]), int(seg_map.shape[1])
    seg, td, bu = seg_map.contiguous(), td_hf.contiguous(), bu_hf.contiguous()
    pts = pts_map.contiguous().view(torch.uint8)
    L = V.lib()
    with torch.cuda.device(dev):
        lut = _scale_lut(classes, scales, dev)
        m = _seg_ins(seg_ins_map)
        nbytes = L.gcv_extrude_scratch_bytes(H, W)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        n = C.c_int64(0)
        a = (int(bool(include_btm_pts)), lut.data_ptr(), C.byref(m), H, W, seg.data_ptr(), td.data_ptr(), bu.data_ptr(),
             pts.data_ptr(), scratch.data_ptr(), nbytes)
        V.check(L.gcv_extrude_count(*a, C.byref(n), _stream()), "gcv_extrude_count")
        out = torch.empty((n.value, 5), dtype=torch.int16, device=dev)
        if n.value:
            V.check(L.gcv_extrude_emit(*a, out.data_ptr(), n.value, _stream()), "gcv_extrude_emit")
    return out


def get_points_from_projection(include_btm_pts, classes, scales, seg_ins_map, seg_map, td_hf, bu_hf, pts_map):
    """Drop-in for footprint_extruder.get_points_from_projection: numpy in, numpy uint16 [N,5] out,
    `None` when no point is generated (footprint_extruder.cpp:208-212).  Argument types are checked as
    PyArg_ParseTuple("O!O!O!O!O!O!O!O!") does upstream (:152-158)."""
    if not isinstance(include_btm_pts, bool):
        raise TypeError("argument 1 must be bool, not %s" % type(include_btm_pts).__name__)
    for k, d in enumerate((classes, scales, seg_ins_map)):
        if not isinstance(d, dict):
            raise TypeError("argument %d must be dict, not %s" % (k + 2, type(d).__name__))
    for k, a in enumerate((seg_map, td_hf, bu_hf, pts_map)):
        if not isinstance(a, np.ndarray):
            raise TypeError("argument %d must be numpy.ndarray, not %s" % (k + 5, type(a).__name__))
    dev = _need_gpu()
    # upstream reinterprets the buffers as short / bool without looking at dtype; the callers hand over
    # contiguous int16 / bool arrays (scripts/dataset_generator.py:1316-1325), which is what is required here
    if seg_map.dtype != np.int16 or td_hf.dtype != np.int16 or bu_hf.dtype != np.int16:
        raise TypeError("seg_map, td_hf and bu_hf must be int16 arrays")
    if pts_map.dtype != np.bool_:
        raise TypeError("pts_map must be a bool array")
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (seg_map, td_hf, bu_hf, pts_map)]
    out = extrude_points(include_btm_pts, classes, scales, seg_ins_map, *t)
    if out.shape[0] == 0:
        return None
    return out.cpu().numpy().view(np.uint16)


def points_to_volume(points, pt_ids, scales, h, w, d, return_occupancy=False):
    """voxlib.points_to_volume: points int16 [N,3], pt_ids int32 [N,1], scales int16 [N,3] (CUDA) ->
    int32 volume [h,w,d].  Overlapping cubes: the highest id wins (deterministic; upstream races)."""
    if not points.is_cuda or not pt_ids.is_cuda or not scales.is_cuda:
        raise RuntimeError("points must be a CUDA tensor")  # CHECK_CUDA, points_to_volume.cu:56-58
    if points.dtype != torch.int16 or scales.dtype != torch.int16 or pt_ids.dtype != torch.int32:
        raise RuntimeError("expected scalar type Short / Int")  # data_ptr<short>() / data_ptr<int>() upstream
    dev = points.device
    n = int(points.shape[0])
    points, pt_ids, scales = points.contiguous(), pt_ids.contiguous(), scales.contiguous()
    L = V.lib()
    with torch.cuda.device(dev):
        volume = torch.empty((int(h), int(w), int(d)), dtype=torch.int32, device=dev)
        occ = None
        if return_occupancy:
            occ = torch.empty(max(1, L.gcv_occupancy_bytes(int(h), int(w), int(d)) // 4), dtype=torch.int32, device=dev)
        V.check(L.gcv_points_to_volume(n, points.data_ptr(), pt_ids.data_ptr(), scales.data_ptr(), int(h), int(w),
                                       int(d), volume.data_ptr(), occ.data_ptr() if occ is not None else None,
                                       _stream()), "gcv_points_to_volume")
    return (volume, occ) if return_occupancy else volume


MAPS_TO_VOLUME_DEPTH = 504  # BLDG_MAX_HEIGHT, extensions/voxlib/maps_to_volume.cu:16


def maps_to_volume(inst_map, td_hf, bu_hf, pts_map, scales, depth=MAPS_TO_VOLUME_DEPTH):
    """voxlib.maps_to_volume (extensions/voxlib/maps_to_volume.cu:103-142): int16 CUDA maps [H,W], bool point map,
    int8 per-class scales -> int16 instance volume [H, W, 504]."""
    for t, n in ((inst_map, "inst_map"), (td_hf, "td_hf"), (bu_hf, "bu_hf"), (pts_map, "pts_map"), (scales, "scales")):
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % n)  # CHECK_CUDA, :108-112
    if inst_map.dtype != torch.int16 or td_hf.dtype != torch.int16 or bu_hf.dtype != torch.int16:
        raise RuntimeError("expected scalar type Short")
    if pts_map.dtype not in (torch.bool, torch.uint8) or scales.dtype != torch.int8:
        raise RuntimeError("expected pts_map Bool and scales Char")
    dev = inst_map.device
    H, W = int(inst_map.shape[0]), int(inst_map.shape[1])
    with torch.cuda.device(dev):
        volume = torch.empty((H, W, int(depth)), dtype=torch.int16, device=dev)
        scratch = torch.empty(1, dtype=torch.int64, device=dev)
        V.check(V.lib().gcv_maps_to_volume(inst_map.contiguous().data_ptr(), td_hf.contiguous().data_ptr(),
                                           bu_hf.contiguous().data_ptr(), pts_map.contiguous().view(torch.uint8).data_ptr(),
                                           scales.contiguous().data_ptr(), int(scales.numel()), H, W, int(depth),
                                           volume.data_ptr(), scratch.data_ptr(), _stream()), "gcv_maps_to_volume")
    return volume


# The drop-in traversal builds a macro-cell bitmask on the fly for contiguous volumes with at least this
# many voxels (None = never).  Off by default: on the dense city workload the jumps are slower than the
# plain walk; they pay off on sparse volumes.
OCCUPANCY_MIN_VOXELS = None


def ray_voxel_intersection_perspective(in_voxel, cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims, max_samples,
                                       occupancy=None):
    """voxlib.ray_voxel_intersection_perspective (ray_voxel_intersection.cu:232-332): returns
    [out_voxel_id int32 [H,W,S,1], out_depth float [2,H,W,S,1], out_raydirs float [H,W,1,3]]."""
    if not in_voxel.is_cuda:
        raise RuntimeError("in_voxel must be a CUDA tensor")  # CHECK_CUDA, :238
    if in_voxel.dtype != torch.int32 or in_voxel.dim() != 3:
        raise RuntimeError("in_voxel must be an int32 tensor with 3 dimensions")  # asserts :245-246
    cam = []
    for name, t in (("cam_ori", cam_ori), ("cam_dir", cam_dir), ("cam_up", cam_up)):
        t = torch.as_tensor(t)
        if t.dtype != torch.float32 or t.numel() != 3:
            raise RuntimeError("%s must be a float32 tensor with 3 elements" % name)  # asserts :247-252
        cam.append((C.c_float * 3)(*[float(v) for v in t.detach().cpu().reshape(-1)]))
    if len(img_dims) != 2 or len(cam_c) != 2:
        raise RuntimeError("img_dims and cam_c must have 2 elements")
    dev = in_voxel.device
    H, W, S = int(img_dims[0]), int(img_dims[1]), int(max_samples)
    L = V.lib()
    with torch.cuda.device(dev):
        vid = torch.empty((H, W, S, 1), dtype=torch.int32, device=dev)
        dep = torch.empty((2, H, W, S, 1), dtype=torch.float32, device=dev)
        rd = torch.empty((H, W, 1, 3), dtype=torch.float32, device=dev)
        dims = (C.c_int32 * 3)(*[int(v) for v in in_voxel.shape])
        strides = (C.c_int64 * 3)(*[int(v) for v in in_voxel.stride()])
        if (occupancy is None and OCCUPANCY_MIN_VOXELS is not None and in_voxel.is_contiguous()
                and in_voxel.numel() >= OCCUPANCY_MIN_VOXELS):
            h, w, d = [int(v) for v in in_voxel.shape]
            occupancy = torch.empty(max(1, L.gcv_occupancy_bytes(h, w, d) // 4), dtype=torch.int32, device=dev)
            V.check(L.gcv_build_occupancy(in_voxel.data_ptr(), h, w, d, occupancy.data_ptr(), _stream()),
                    "gcv_build_occupancy")
        V.check(L.gcv_ray_voxel_intersection(
            in_voxel.data_ptr(), dims, strides, occupancy.data_ptr() if occupancy is not None else None,
            cam[0], cam[1], cam[2], float(cam_f), (C.c_float * 2)(float(cam_c[0]), float(cam_c[1])),
            (C.c_int32 * 2)(H, W), S, vid.data_ptr(), dep.data_ptr(), rd.data_ptr(), _stream()),
            "gcv_ray_voxel_intersection")
    return [vid, dep, rd]


class VolumeWorkspace:
    """A resident, all-zero int32 buffer for visible_point_map.  The 5 GB bounding-box volume of a city frame
    takes 0.78 ms just to clear; with a workspace each frame writes its ~32 M voxels into the resident buffer and
    erases exactly those voxels after the traversal, so the buffer is zero again for the next frame (sized for the
    largest volume seen; grows on demand).  One workspace per stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.buf = None

    def take(self, n_voxels):
        if self.buf is None or self.buf.numel() < n_voxels:
            self.buf = None
            self.buf = torch.zeros(int(n_voxels), dtype=torch.int32, device=self.device)
        return self.buf[:n_voxels]


def visible_point_map(rows, cam_rig, cam_pos, cam_quat, null_class_id=0, use_jumps=False, workspace=None):
    """Device-resident get_visible_points (scripts/dataset_generator.py:1414-1461) for rows as the extruder
    writes them: int16 CUDA tensor [N,5] = (x, y, z, scale, instance) -> (vp_map int64 [H,W], ins_map [H,W]) on
    the GPU.  Cube side = column 3 (get_point_scales without special classes); the only host round trip is
    the bounding box.  Same outputs as get_visible_points on the same points."""
    if not rows.is_cuda or rows.dtype != torch.int16 or rows.dim() != 2 or rows.shape[1] != 5:
        raise RuntimeError("rows must be an int16 CUDA tensor [N,5]")
    dev = rows.device
    rows = rows.contiguous()
    n = int(rows.shape[0])
    L = V.lib()
    with torch.cuda.device(dev):
        scratch = torch.empty(L.gcv_bounds_scratch_bytes() // 4, dtype=torch.int32, device=dev)
        mn, mx = (C.c_int32 * 3)(), (C.c_int32 * 3)()
        V.check(L.gcv_points_bounds(n, rows.data_ptr(), 5, scratch.data_ptr(), mn, mx, _stream()), "gcv_points_bounds")
        w, h, d = mx[0] - mn[0] + 1, mx[1] - mn[1] + 1, mx[2] - mn[2] + 2  # :1376
        off = (C.c_int32 * 3)(mn[0], mn[1], mn[2] - 1)                     # _get_localized_pt_cords, :1359-1363
        if workspace is not None:
            volume = workspace.take(h * w * d).view(h, w, d)   # known to be all zero
        else:
            volume = torch.empty((h, w, d), dtype=torch.int32, device=dev)
        occ = None
        if use_jumps:
            occ = torch.empty(max(1, L.gcv_occupancy_bytes(h, w, d) // 4), dtype=torch.int32, device=dev)
        V.check(L.gcv_rows_to_volume(n, rows.data_ptr(), off, h, w, d, volume.data_ptr(),
                                     occ.data_ptr() if occ is not None else None, int(workspace is not None), _stream()),
                "gcv_rows_to_volume")
        cp = np.array(cam_pos, dtype=np.float64) - np.array([mn[0], mn[1], mn[2]], dtype=np.int16)  # :1440
        look = get_camera_look_at(cp, cam_quat)
        K, sensor = cam_rig["intrinsics"], cam_rig["sensor_size"]
        vid = ray_voxel_intersection_perspective(
            volume, torch.tensor([cp[1], cp[0], cp[2]], dtype=torch.float32),
            torch.tensor([look[1] - cp[1], look[0] - cp[0], look[2] - cp[2]], dtype=torch.float32),
            torch.tensor([0, 0, 1], dtype=torch.float32), K[0], [K[5], K[2]], [sensor[1], sensor[0]], 1,
            occupancy=occ)[0]
        if workspace is not None:  # give the buffer back all-zero
            V.check(L.gcv_rows_erase_volume(n, rows.data_ptr(), off, h, w, d, volume.data_ptr(), _stream()),
                    "gcv_rows_erase_volume")
        vp_map = vid.view(sensor[1], sensor[0]).long() - 1
        ins_map = rows[:, 4][vp_map.clamp(min=0)]
        ins_map[vp_map == -1] = null_class_id
    return vp_map, ins_map


def get_camera_look_at(cam_position, cam_quaternion, step=1000):
    """utils/helpers.py:162-164."""
    mat3 = scipy.spatial.transform.Rotation.from_quat(cam_quaternion).as_matrix()
    return cam_position + mat3[:3, 0] * step


def get_visible_points(points, scales, cam_rig, cam_pos, cam_quat, null_class_id=0, reduce_mem=False):
    """scripts/dataset_generator.py:1414-1461: points int16 [N,5] (numpy), per-point scales [N,3] ->
    (vp_map [H,W] int: index of the point each pixel sees or -1, ins_map [H,W]: its instance id).
    Same steps as upstream -- volume of point ids, perspective traversal, id - 1 -- with nothing but the
    two result maps leaving the GPU.  `visible_point_map` is the device-resident form."""
    dev = _need_gpu()
    cam_pos = np.array(cam_pos, dtype=np.float64)  # upstream mutates its argument; callers pass a copy (:329)
    pts_np = np.asarray(points)
    instances = torch.from_numpy(np.ascontiguousarray(pts_np[:, 4])).to(dev)
    pts = torch.from_numpy(np.ascontiguousarray(pts_np[:, [0, 1, 2]])).to(dev)
    scales = (torch.from_numpy(np.ascontiguousarray(scales)) if isinstance(scales, np.ndarray) else scales).to(dev)
    if reduce_mem:  # :1428-1433
        factor = 1 / 3.0
        cam_pos *= factor
        scales = (scales * factor).clamp(min=1).short()
        pts = torch.floor(pts * factor).short()
    pts, scales = pts.short(), scales.short()
    # _get_volume, :1366-1388
    mn, mx = pts.min(dim=0).values.cpu().numpy(), pts.max(dim=0).values.cpu().numpy()
    offsets = np.array([mn[0], mn[1], mn[2]], dtype=np.int16)
    loc = pts.clone()
    loc[:, 0] -= int(offsets[0])
    loc[:, 1] -= int(offsets[1])
    loc[:, 2] -= int(offsets[2]) - 1
    w, h, d = int(mx[0]) - int(mn[0]) + 1, int(mx[1]) - int(mn[1]) + 1, int(mx[2]) - int(mn[2]) + 2
    assert loc.shape[0] < 2147483648
    pt_ids = torch.arange(1, loc.shape[0] + 1, dtype=torch.int32, device=dev).unsqueeze(1)
    volume = points_to_volume(loc, pt_ids, scales, h, w, d)
    # _get_ray_voxel_intersection, :1391-1411
    cam_pos = cam_pos - offsets
    look = get_camera_look_at(cam_pos, cam_quat)
    ori = torch.tensor([cam_pos[1], cam_pos[0], cam_pos[2]], dtype=torch.float32)
    view = torch.tensor([look[1] - cam_pos[1], look[0] - cam_pos[0], look[2] - cam_pos[2]], dtype=torch.float32)
    K, sensor = cam_rig["intrinsics"], cam_rig["sensor_size"]
    vid, _, _ = ray_voxel_intersection_perspective(volume, ori, view, torch.tensor([0, 0, 1], dtype=torch.float32),
                                                   K[0], [K[5], K[2]], [sensor[1], sensor[0]], 1)
    vp_map = vid.squeeze().long() - 1
    ins_map = instances[vp_map.clamp(min=0)]
    ins_map[vp_map == -1] = null_class_id
    return vp_map.cpu().numpy(), ins_map.cpu().numpy()
