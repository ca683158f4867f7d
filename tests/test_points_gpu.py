"""-m gpu: the HIP point-generation / visibility path (gaussiancity_amd.points -> C ABI include/gcv.h ->
gfx950 kernels) against the oracle and the reference-generated golden vectors.  Everything here is
integer / index work or IEEE fp32 in a fixed order, so the bar is BIT-EXACT throughout."""
import numpy as np
import pytest
import torch

import points_util as U
from gaussiancity_amd import points as P
from gaussiancity_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def po():
    from oracle import points_oracle as PO
    PO.lib()
    return PO


@pytest.mark.parametrize("case", U.golden_cases(), ids=lambda c: c[0])
def test_extruder_matches_reference_golden(cuda_device, case):
    name, inv, scales, seg_ins, seg, td, bu, pts, want = case
    for inc in (True, False):
        got = P.get_points_from_projection(inc, inv, scales, seg_ins, seg, td, bu, pts)
        if want[inc].shape[0] == 0:
            assert got is None
        else:
            assert got.dtype == np.uint16 and got.shape == want[inc].shape
            assert np.array_equal(got, want[inc])


def test_extruder_matches_oracle_on_ragged_random_maps(cuda_device, po):
    rng = np.random.default_rng(321)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    for trial in range(8):
        H, W = int(rng.integers(1, 300)), int(rng.integers(1, 300))  # incl. maps smaller than a scale
        L = synth.s_layout(max(H, W, 16), 4000 + trial, block=int(rng.integers(16, 64)), road=int(rng.integers(2, 9)),
                           max_height=int(rng.integers(10, 120)))
        seg, td, pts = (np.ascontiguousarray(L[k][:H, :W]) for k in ("INS", "TD_HF", "PTS"))
        bu = rng.integers(-1, 4, (H, W)).astype(np.int16)
        for inc in (True, False):
            a = po.extrude(inc, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu, pts)
            b = P.get_points_from_projection(inc, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu, pts)
            assert (a is None) == (b is None), (H, W, inc)
            if a is not None:
                assert np.array_equal(a, b), (H, W, inc)


def test_extruder_argument_and_class_errors(cuda_device):
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    z = np.zeros((8, 8), np.int16)
    ok = (True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, z + 1, z + 3, z, np.ones((8, 8), bool))
    assert P.get_points_from_projection(*ok) is not None
    for k, bad in ((0, 1), (1, []), (2, None), (3, 7), (4, z.tolist()), (7, "x")):  # PyArg_ParseTuple O! checks
        args = list(ok)
        args[k] = bad
        with pytest.raises(TypeError):
            P.get_points_from_projection(*args)
    with pytest.raises(RuntimeError, match="semantic id"):  # upstream would never terminate here
        P.get_points_from_projection(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, z + 9, z + 3, z,
                                     np.ones((8, 8), bool))
    with pytest.raises(IndexError):  # std::map::at upstream
        P.get_points_from_projection(True, inv, synth.LAYOUT_SCALES, {"BLDG_INS_MIN_ID": 100}, z + 1, z + 3, z,
                                     np.ones((8, 8), bool))


def test_points_to_volume_and_occupancy(cuda_device, po):
    rng = np.random.default_rng(11)
    for h, w, d, n in ((13, 17, 11, 500), (64, 40, 33, 6000), (8, 8, 8, 1), (5, 3, 70, 0)):
        pts, ids, sc = U.random_points(rng, n, h, w, d)
        want = po.points_to_volume(pts, ids, sc, h, w, d)
        t = [torch.from_numpy(a).to(cuda_device) for a in (pts, ids, sc)]
        vol, occ = P.points_to_volume(*t, h, w, d, return_occupancy=True)
        assert vol.dtype == torch.int32 and tuple(vol.shape) == (h, w, d)
        assert np.array_equal(vol.cpu().numpy(), want)
        assert torch.equal(P.points_to_volume(*t, h, w, d), vol)
        # occupancy: exactly the 16^3 macro cells holding a non-zero voxel
        hb, wb, db = (h + 15) // 16, (w + 15) // 16, (d + 15) // 16
        pad = np.zeros((hb * 16, wb * 16, db * 16), np.int32)
        pad[:h, :w, :d] = want
        brick_any = pad.reshape(hb, 16, wb, 16, db, 16).any(axis=(1, 3, 5)).reshape(-1)
        bits = np.unpackbits(occ.cpu().numpy().view(np.uint8), bitorder="little")[:brick_any.size].astype(bool)
        assert np.array_equal(bits, brick_any)
        # the same bitmask from the dense volume alone
        from gaussiancity_amd import _native_v as V
        occ2 = torch.zeros_like(occ)
        V.check(V.lib().gcv_build_occupancy(vol.data_ptr(), h, w, d, occ2.data_ptr(), None), "gcv_build_occupancy")
        torch.cuda.synchronize()
        assert torch.equal(occ2, occ)
    with pytest.raises(RuntimeError):
        P.points_to_volume(torch.zeros((1, 3), dtype=torch.int16), t[1], t[2], 4, 4, 4)  # CHECK_CUDA


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("use_occ", [False, True])
def test_traversal_bit_exact(cuda_device, po, variant, use_occ):
    rng = np.random.default_rng(70 + variant)
    h, w, d = 72, 61, 37
    vol = U.shell_volume(rng, h, w, d, 8)
    rows, cols = 45, 83  # ragged: not multiples of the 8x8 tile
    ori, dr, up, f, c, img = U.camera_for_volume(h, w, d, rows, cols, variant)
    want = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, 2)
    v = torch.from_numpy(vol).to(cuda_device)
    occ = None
    if use_occ:
        from gaussiancity_amd import _native_v as V
        occ = torch.zeros(max(1, V.lib().gcv_occupancy_bytes(h, w, d) // 4), dtype=torch.int32, device=cuda_device)
        V.check(V.lib().gcv_build_occupancy(v.data_ptr(), h, w, d, occ.data_ptr(), None), "gcv_build_occupancy")
    got = P.ray_voxel_intersection_perspective(v, torch.from_numpy(ori), torch.from_numpy(dr), torch.from_numpy(up),
                                               f, c, img, 2, occupancy=occ)
    assert [tuple(t.shape) for t in got] == [(rows, cols, 2, 1), (2, rows, cols, 2, 1), (rows, cols, 1, 3)]
    assert np.array_equal(got[0].cpu().numpy(), want[0])
    assert np.array_equal(got[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32))  # incl. the NaN pattern
    assert np.array_equal(got[2].cpu().numpy().view(np.uint32), want[2].view(np.uint32))
    assert (want[0] != 0).mean() > 0.2


def test_traversal_jumps_reproduce_the_reference_walk(cuda_device, po):
    """The empty-space jump must land exactly where upstream's cell-by-cell walk would: sparse volumes
    (long jumps), origins on integer coordinates and axis-parallel / diagonal view directions (crossing-time
    ties between axes), origins inside, outside and far from the volume, several samples per ray."""
    from gaussiancity_amd import _native_v as V
    rng = np.random.default_rng(2024)
    n_checked = 0
    for trial in range(20):
        h, w, d = (int(v) for v in rng.integers(17, 150, 3))
        vol = np.zeros((h, w, d), np.int32)
        nvox = int(rng.integers(1, 60))
        vol[rng.integers(0, h, nvox), rng.integers(0, w, nvox), rng.integers(0, d, nvox)] = rng.integers(1, 1 << 30, nvox)
        if trial % 3 == 0:
            vol[:, :, 0] = 5  # a floor far below sparse clutter
        rows, cols = int(rng.integers(9, 70)), int(rng.integers(9, 70))
        centre = np.array([h, w, d], np.float32) * rng.uniform(0.2, 0.8, 3).astype(np.float32)
        kind = trial % 5
        if kind == 0:    # integer origin, view along -z: dx = dy = 0 for the centre ray, ties everywhere
            ori = np.array([h // 2, w // 2, d + 40], np.float32); dr = np.array([0, 0, -1], np.float32); up = np.array([0, 1, 0], np.float32)
        elif kind == 1:  # exact diagonal
            ori = np.array([-8, -8, -8], np.float32); dr = np.array([1, 1, 1], np.float32); up = np.array([0, 0, 1], np.float32)
        elif kind == 2:  # inside the volume, half-integer origin
            ori = (np.floor(centre) + 0.5).astype(np.float32); dr = rng.normal(size=3).astype(np.float32); up = np.array([0, 0, 1], np.float32)
        elif kind == 3:  # far outside, grazing
            ori = np.array([-3.0 * h, 0.37 * w, 0.9 * d], np.float32); dr = (centre - ori).astype(np.float32); up = np.array([0, 0, 1], np.float32)
        else:            # outside, looking away over an edge of the grid
            ori = np.array([h + 5.25, w * 0.5, d * 0.5], np.float32); dr = np.array([-1.0, 0.2, -0.1], np.float32); up = np.array([0, 0, 1], np.float32)
        if trial >= 15:  # out of range on two or three axes at once, integer and fractional origins
            ori = np.array([-7 - trial, w + 3.0 + 0.5 * (trial & 1), d + 11.25], np.float32)
            dr = (centre - ori).astype(np.float32); up = np.array([0, 0, 1], np.float32)
        f, c, img = float(rng.uniform(0.4, 2.0) * cols), [rows * 0.5, cols * 0.5], [rows, cols]
        S = 3 if trial % 2 else 1
        want = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, S)
        v = torch.from_numpy(vol).to(cuda_device)
        occ = torch.zeros(max(1, V.lib().gcv_occupancy_bytes(h, w, d) // 4), dtype=torch.int32, device=cuda_device)
        V.check(V.lib().gcv_build_occupancy(v.data_ptr(), h, w, d, occ.data_ptr(), None), "gcv_build_occupancy")
        cam = [torch.from_numpy(a) for a in (ori, dr, up)]
        for entry in (1, 0):  # closed-form walk from the camera to the grid on / off
            V.set_option("entry_jump", entry)
            for o in (None, occ):
                got = P.ray_voxel_intersection_perspective(v, *cam, f, c, img, S, occupancy=o)
                assert np.array_equal(got[0].cpu().numpy(), want[0]), (trial, o is not None, entry)
                assert np.array_equal(got[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32)), (trial, o is not None, entry)
        V.set_option("entry_jump", 1)
        n_checked += int((want[0] != 0).sum())
    assert n_checked > 200


def test_traversal_strided_volume_and_errors(cuda_device, po):
    rng = np.random.default_rng(3)
    h, w, d = 30, 26, 20
    vol = U.shell_volume(rng, h, w, d, 5)
    ori, dr, up, f, c, img = U.camera_for_volume(h, w, d, 24, 40, 0)
    want = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, 1)
    vT = torch.from_numpy(np.ascontiguousarray(vol.transpose(2, 0, 1))).to(cuda_device).permute(1, 2, 0)  # same values, other strides
    assert not vT.is_contiguous()
    cam = [torch.from_numpy(a) for a in (ori, dr, up)]
    got = P.ray_voxel_intersection_perspective(vT, *cam, f, c, img, 1)
    assert np.array_equal(got[0].cpu().numpy(), want[0])
    assert np.array_equal(got[1].cpu().numpy().view(np.uint32), want[1].view(np.uint32))
    with pytest.raises(RuntimeError):
        P.ray_voxel_intersection_perspective(torch.from_numpy(vol), *cam, f, c, img, 1)  # CPU tensor
    with pytest.raises(RuntimeError):
        P.ray_voxel_intersection_perspective(vT.float(), *cam, f, c, img, 1)
    with pytest.raises(RuntimeError):
        P.ray_voxel_intersection_perspective(vT, cam[0].double(), cam[1], cam[2], f, c, img, 1)


def _visible_points_oracle(po, points, scales, rig, cam_pos, cam_quat, null_id):
    """scripts/dataset_generator.py:1414-1461 composed from the oracle's pieces (numpy)."""
    pts = points[:, [0, 1, 2]].astype(np.int16)
    mn, mx = pts.min(0), pts.max(0)
    loc = pts.copy()
    loc[:, 0] -= mn[0]; loc[:, 1] -= mn[1]; loc[:, 2] -= mn[2] - 1
    w, h, d = int(mx[0]) - int(mn[0]) + 1, int(mx[1]) - int(mn[1]) + 1, int(mx[2]) - int(mn[2]) + 2
    vol = po.points_to_volume(loc, np.arange(1, len(pts) + 1, dtype=np.int32), scales.astype(np.int16), h, w, d)
    cam = np.array(cam_pos, np.float64) - mn.astype(np.int16)
    look = P.get_camera_look_at(cam, cam_quat)
    ori = np.array([cam[1], cam[0], cam[2]], np.float32)
    view = np.array([look[1] - cam[1], look[0] - cam[0], look[2] - cam[2]], np.float32)
    K, sensor = rig["intrinsics"], rig["sensor_size"]
    vid, _, _ = po.ray_voxel_intersection_perspective(vol, ori, view, np.array([0, 0, 1], np.float32), K[0],
                                                      [K[5], K[2]], [sensor[1], sensor[0]], 1)
    vp = vid.squeeze().astype(np.int64) - 1
    ins = points[:, 4][np.clip(vp, 0, None)].copy()
    ins[vp == -1] = null_id
    return vp, ins


def test_get_visible_points_end_to_end(cuda_device, po):
    """maps -> extruded points -> volume -> traversal, HIP vs oracle, on a 256 px layout."""
    size = 256
    L = synth.s_layout(size, 2201, block=64, road=8, max_height=70)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    pts = P.get_points_from_projection(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"],
                                       L["BU_HF"], L["PTS"])
    want_pts = po.extrude(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"], L["BU_HF"], L["PTS"])
    assert np.array_equal(pts, want_pts)
    points = pts.astype(np.int16)
    scales = np.repeat(points[:, [3]], 3, axis=1)  # utils/helpers.get_point_scales without special classes
    rig, cam_pos, cam_quat = synth.layout_camera(size, W=240, H=136)
    vp, ins = P.get_visible_points(points, scales, rig, cam_pos.copy(), cam_quat, null_class_id=0)
    vp_o, ins_o = _visible_points_oracle(po, points, scales, rig, cam_pos, cam_quat, 0)
    assert vp.shape == (136, 240) and np.array_equal(vp, vp_o) and np.array_equal(ins, ins_o)
    assert (vp >= 0).mean() > 0.5
    # the device-resident form, plain walk and with empty-space jumps
    rows = torch.from_numpy(points).to(cuda_device)
    for jumps in (False, True):
        vp_d, ins_d = P.visible_point_map(rows, rig, cam_pos.copy(), cam_quat, 0, use_jumps=jumps)
        assert np.array_equal(vp_d.cpu().numpy(), vp_o) and np.array_equal(ins_d.cpu().numpy(), ins_o)
    # resident workspace: same maps, and the buffer is all-zero again after every call (also after it grew)
    ws = P.VolumeWorkspace(cuda_device)
    for k in range(3):
        sub = rows[: len(rows) // (3 - k)] if k < 2 else rows
        want_vp, want_ins = P.visible_point_map(sub, rig, cam_pos.copy(), cam_quat, 0)
        got_vp, got_ins = P.visible_point_map(sub, rig, cam_pos.copy(), cam_quat, 0, workspace=ws)
        assert torch.equal(got_vp, want_vp) and torch.equal(got_ins, want_ins)
        assert int(ws.buf.count_nonzero()) == 0
    # reduce_mem (scale 1/3, :1428-1433) runs and sees the same scene
    vp3, _ = P.get_visible_points(points, scales, rig, cam_pos.copy(), cam_quat, 0, reduce_mem=True)
    assert vp3.shape == vp.shape and (vp3 >= 0).mean() > 0.5


def test_replay_of_a_dataset_frame_as_the_generator_wrote_it(cuda_device, tmp_path):
    """Row f4 end to end: a Points/%04d.pkl + CameraPoses.csv pair produced the way scripts/dataset_generator.py
    produces them (:1543-1549 camera normalisation, :1591-1617 visible-only points re-indexed with searchsorted,
    KITTI-360 flipped) is replayed by formats.replay_visible_points and reproduces the stored map exactly."""
    from gaussiancity_amd import formats as F
    size = 256
    L = synth.s_layout(size, 2301, block=64, road=8, max_height=60)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    points = P.get_points_from_projection(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"],
                                          L["BU_HF"], L["PTS"]).astype(np.int16)
    scales = np.repeat(points[:, [3]], 3, axis=1)
    rig, cam_pos, cam_quat = synth.layout_camera(size, W=200, H=120)
    rig["intrinsics"][0] *= 0.3   # a wide field of view: part of the image looks past the map's edge ("no hit" pixels)
    rig["intrinsics"][4] *= 0.3
    for dataset in ("GOOGLE_EARTH", "KITTI_360"):
        c = F.DATASET_CONSTANTS[dataset]
        # the CSV stores the pose BEFORE normalisation: invert cam_pos = csv / SCALE + MAP_SIZE // 2
        csv_pos = cam_pos.astype(np.float32).copy()
        csv_pos[:2] -= c["MAP_SIZE"] // 2
        csv_pos *= c["SCALE"]
        row = dict(id=7, tx=float(csv_pos[0]), ty=float(csv_pos[1]), tz=float(csv_pos[2]),
                   qx=float(cam_quat[0]), qy=float(cam_quat[1]), qz=float(cam_quat[2]), qw=float(cam_quat[3]))
        F.write_camera_poses_csv(str(tmp_path / "CameraPoses.csv"), [row])
        row_read = F.read_camera_poses_csv(str(tmp_path / "CameraPoses.csv"))[7]
        pos_n, quat_n = F.pose_arrays(row_read, c["SCALE"], c["MAP_SIZE"])
        vp, _ = P.get_visible_points(points, scales, rig, pos_n.astype(np.float64), quat_n.astype(np.float64), 0)
        if dataset == "KITTI_360":
            vp = np.fliplr(vp)
        vp_idx = np.sort(np.unique(vp))
        vp_idx = vp_idx[vp_idx >= 0]
        stored_pts, stored_vpm = points[vp_idx], np.searchsorted(vp_idx, vp)
        assert len(stored_pts) < len(points) and (vp < 0).any()
        pkl = str(tmp_path / ("%04d.pkl" % 7))
        F.write_points_pkl(pkl, {k: L[k] for k in ("INS", "SEG", "TD_HF", "BU_HF", "PTS")}, stored_vpm, vp >= 0, stored_pts)
        got, stored, frac = F.replay_visible_points(pkl, row_read, rig, dataset=dataset)
        assert frac == 1.0 and np.array_equal(got, stored_vpm)


def test_full_size_extruder_properties(cuda_device):
    """BASELINE-size map (2048 x 2048, ~17 M points): order, ranges and hollowness hold; two runs agree."""
    L = synth.s_layout(2048, 2001)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    t = [torch.from_numpy(L[k]).to(cuda_device) for k in ("INS", "TD_HF", "BU_HF", "PTS")]
    out = P.extrude_points(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, *t)
    out2 = P.extrude_points(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, *t)
    assert torch.equal(out, out2) and out.shape[0] > 10_000_000
    x, y, z, s, ins = (out[:, k].long() for k in range(5))
    key = ((y * 2048 + x) << 16) + (z + 1024)
    assert bool((key[1:] > key[:-1]).all())  # row-major pixels, z ascending, no duplicates
    seg, td, bu, pm = (v.long() for v in t)
    assert bool(pm[y, x].all()) and bool((z <= td[y, x]).all()) and bool((z >= bu[y, x]).all())
    assert bool(((z - bu[y, x]) % s == 0).all())
    roof = ins != seg[y, x]
    assert bool((ins[roof] == seg[y, x][roof] + 1).all()) and bool((z[roof] > td[y, x][roof] - s[roof]).all())
    # every pixel of the point map contributes its top point exactly once
    top = z > td[y, x] - s
    assert int(top.sum()) == int(((td >= bu) & (pm > 0)).sum())


def test_maps_to_volume_bit_exact(cuda_device, po):
    import extensions.voxlib as voxlib
    rng = np.random.default_rng(78)
    for H, W, depth in ((64, 48, 504), (33, 70, 61)):
        L = synth.s_layout(max(H, W), 5000 + H, block=24, road=4, max_height=58)
        inst = np.ascontiguousarray(np.where(L["INS"][:H, :W] >= 100, L["INS"][:H, :W] - 90, L["INS"][:H, :W])).astype(np.int16)
        td, bu = np.ascontiguousarray(L["TD_HF"][:H, :W]), rng.integers(0, 3, (H, W)).astype(np.int16)
        pts = np.ascontiguousarray(L["PTS"][:H, :W])
        scales = np.array([1, 2, 1, 2, 1, 4, 2, 1, 1, 1], np.int8)
        want = po.maps_to_volume(inst, td, bu, pts, scales, depth)
        t = [torch.from_numpy(a).to(cuda_device) for a in (inst, td, bu, pts, scales)]
        got = voxlib.maps_to_volume(*t) if depth == 504 else P.maps_to_volume(*t, depth=depth)
        assert got.dtype == torch.int16 and tuple(got.shape) == (H, W, depth)
        assert np.array_equal(got.cpu().numpy(), want) and (want != 0).sum() > 1000
    with pytest.raises(RuntimeError, match="positive scale"):
        P.maps_to_volume(t[0], t[1], t[2], t[3], torch.zeros(10, dtype=torch.int8, device=cuda_device))
    with pytest.raises(RuntimeError, match="CUDA"):
        P.maps_to_volume(t[0].cpu(), t[1], t[2], t[3], t[4])
