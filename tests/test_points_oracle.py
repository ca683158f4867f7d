"""CPU tests of the point-generation / visibility oracle (oracle/gcv_oracle.c):
  K15 against the golden vectors produced by the reference's own extruder, and -- where
      oracle/_ref/footprint_extruder.so exists (build container; it also travels to the GPU box) --
      against that reference directly on fresh random maps;
  K14 against a numpy restatement of "highest id wins";
  K12 against an independent float64 traversal (accumulated-t Amanatides-Woo), plus hand-checkable rays.
"""
import numpy as np
import pytest

import points_util as U
from gaussiancity_amd import synth
from oracle import points_oracle as PO


@pytest.fixture(scope="module")
def po():
    PO.lib()
    return PO


@pytest.mark.parametrize("case", U.golden_cases(), ids=lambda c: c[0])
def test_extruder_oracle_matches_reference_golden(po, case):
    name, inv, scales, seg_ins, seg, td, bu, pts, want = case
    for inc in (True, False):
        got = po.extrude(inc, inv, scales, seg_ins, seg, td, bu, pts)
        if want[inc].shape[0] == 0:
            assert got is None  # footprint_extruder.cpp:208-212 returns None for an empty result
        else:
            assert got.dtype == np.uint16 and np.array_equal(got, want[inc])


def test_extruder_oracle_matches_reference_build(po):
    ref = po.reference_extruder()
    if ref is None:
        pytest.skip("oracle/_ref/footprint_extruder.so not built (reference sources absent)")
    rng = np.random.default_rng(123)
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    for trial in range(6):
        H, W = int(rng.integers(9, 90)), int(rng.integers(9, 90))
        L = synth.s_layout(max(H, W), 3000 + trial, block=int(rng.integers(16, 48)), road=int(rng.integers(2, 8)),
                           max_height=int(rng.integers(10, 80)))
        seg, td, pts = L["INS"][:H, :W].copy(), L["TD_HF"][:H, :W].copy(), L["PTS"][:H, :W].copy()
        bu = rng.integers(0, 3, (H, W)).astype(np.int16)
        for inc in (True, False):
            a = po.extrude(inc, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu, pts)
            b = ref.get_points_from_projection(inc, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu, pts)
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a, b)


def test_extruder_unknown_class_is_an_error(po):
    inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
    seg = np.full((8, 8), 9, np.int16)  # semantic id 9 has no class -> upstream loops forever
    z = np.zeros((8, 8), np.int16)
    with pytest.raises(RuntimeError):
        po.extrude(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, z + 3, z, np.ones((8, 8), bool))


def test_points_to_volume_highest_id_wins(po):
    rng = np.random.default_rng(5)
    h, w, d = 13, 17, 11
    pts, ids, sc = U.random_points(rng, 400, h, w, d)
    vol = po.points_to_volume(pts, ids, sc, h, w, d)
    want = np.zeros((h, w, d), np.int32)
    for (x, y, z), pid, (sx, sy, sz) in zip(pts, ids[:, 0], sc):  # points_to_volume.cu:36-48
        if x >= w or y >= h or z >= d or x < 0 or y < 0 or z < 0:
            continue
        want[y:min(y + sy, h), x:min(x + sx, w), z:min(z + sz, d)] = pid
    assert np.array_equal(vol, want)
    assert vol.max() > 0 and (vol == 0).any()


def _traverse_f64(vol, ori, raydir, max_steps=100000):
    """Independent formulation: Amanatides-Woo with accumulated float64 t; first non-zero voxel."""
    pos = np.floor(ori).astype(np.int64)
    step = np.sign(raydir).astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tmax = np.where(raydir > 0, (pos + 1 - ori) / raydir, np.where(raydir < 0, (pos - ori) / raydir, np.inf))
        tdelta = np.where(raydir != 0, 1.0 / np.abs(raydir), np.inf)
    dims = np.array(vol.shape)
    for _ in range(max_steps):
        a = int(np.argmin(tmax))
        pos[a] += step[a]
        tmax[a] += tdelta[a]
        if (step[a] > 0 and pos[a] >= dims[a]) or (step[a] < 0 and pos[a] < 0):
            return 0
        if np.all(pos >= 0) and np.all(pos < dims):
            v = vol[pos[0], pos[1], pos[2]]
            if v != 0:
                return int(v)
    return 0


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_traversal_oracle_matches_float64_formulation(po, variant):
    rng = np.random.default_rng(40 + variant)
    h, w, d = 40, 48, 24
    vol = U.shell_volume(rng, h, w, d)
    rows, cols = 30, 44
    ori, dr, up, f, c, img = U.camera_for_volume(h, w, d, rows, cols, variant)
    vid, dep, rd = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, 1)
    assert vid.shape == (rows, cols, 1, 1) and dep.shape == (2, rows, cols, 1, 1) and rd.shape == (rows, cols, 1, 3)
    assert np.allclose(np.linalg.norm(rd.reshape(-1, 3), axis=1), 1.0, atol=1e-6)
    hit = vid[..., 0, 0] != 0
    assert hit.mean() > 0.3
    t, t2 = dep[0, ..., 0, 0], dep[1, ..., 0, 0]
    assert np.all(np.isnan(t[~hit])) and np.all(t2[hit] >= t[hit]) and np.all(t[hit] >= 0)
    # the entry point ori + t*dir lies on the boundary of the hit voxel
    mism = 0
    for r in range(rows):
        for cc in range(cols):
            want = _traverse_f64(vol, ori.astype(np.float64), rd[r, cc, 0].astype(np.float64))
            mism += int(want != vid[r, cc, 0, 0])
    assert mism <= 0.01 * rows * cols, "%d of %d pixels differ from the float64 traversal" % (mism, rows * cols)
    ys, xs = np.nonzero(hit)
    for r, cc in list(zip(ys, xs))[::7]:
        if vid[r, cc, 0, 0] == 7:  # the ground plane's id is not unique
            continue
        p = ori.astype(np.float64) + float(t[r, cc]) * rd[r, cc, 0].astype(np.float64)
        cell = np.argwhere(vol == vid[r, cc, 0, 0])[0]
        assert np.all(p >= cell - 1e-3) and np.all(p <= cell + 1 + 1e-3)


def test_traversal_multiple_samples_and_strides(po):
    rng = np.random.default_rng(9)
    h, w, d = 24, 20, 16
    vol = U.shell_volume(rng, h, w, d, 4)
    ori, dr, up, f, c, img = U.camera_for_volume(h, w, d, 16, 24, 0)
    v3, d3, _ = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, 3)
    v1, d1, _ = po.ray_voxel_intersection_perspective(vol, ori, dr, up, f, c, img, 1)
    assert np.array_equal(v3[:, :, 0], v1[:, :, 0])  # first sample == single-sample result
    assert np.array_equal(d3[:, :, :, 0].view(np.uint32), d1[:, :, :, 0].view(np.uint32))
    two = (v3[:, :, 1, 0] != 0)
    assert two.any() and np.all(d3[0][two][:, 1, 0] >= d3[1][two][:, 0, 0] - 1e-4)  # 2nd entry after 1st exit
    # a permuted (non-contiguous) view gives the same answer as its contiguous copy with permuted camera
    volT = np.ascontiguousarray(vol.transpose(1, 0, 2)).transpose(1, 0, 2)
    assert not volT.flags["C_CONTIGUOUS"]
    vT, dT, _ = po.ray_voxel_intersection_perspective(volT, ori, dr, up, f, c, img, 1)
    assert np.array_equal(vT, v1) and np.array_equal(dT.view(np.uint32), d1.view(np.uint32))


def test_maps_to_volume_oracle_against_numpy(po):
    """extensions/voxlib/maps_to_volume.cu:21-101 restated in numpy, pixel by pixel."""
    rng = np.random.default_rng(77)
    H, W, depth = 37, 41, 40
    inst = rng.choice(np.array([1, 3, 4, 6, 12, 14], np.int16), size=(H, W))
    inst[8:20, 10:30] = 12
    td = rng.integers(0, 6, (H, W)).astype(np.int16)
    td[8:20, 10:30] = 33
    td[25:30, 5:12] = 47  # taller than the volume: those z are skipped, not written out of bounds
    bu = rng.integers(0, 2, (H, W)).astype(np.int16)
    pts = rng.random((H, W)) < 0.8
    scales = np.array([1, 2, 1, 2, 1, 4, 2, 1, 1, 1], np.int8)
    vol = po.maps_to_volume(inst, td, bu, pts, scales, depth)
    want = np.zeros((H, W, depth), np.int16)
    for j in range(H):
        for i in range(W):
            if not pts[j, i]:
                continue
            ins = int(inst[j, i]); sem = ins if ins < 10 else 2; s = int(scales[sem]); up = int(td[j, i])
            edge = i < s or i >= W - s - 1 or j < s or j >= H - s - 1
            border = edge
            if not edge:
                nb = [(j - s, i - s), (j - s, i), (j - s, i + s), (j, i - s), (j, i + s), (j + s, i - s), (j + s, i), (j + s, i + s)]
                border = any(td[a, b] != up for a, b in nb) or any(inst[a, b] != ins for a, b in nb)
            for k in range(int(bu[j, i]), up + 1, s):
                top = k > up - s
                if (top or border) and 0 <= k < depth:
                    want[j, i, k] = ins + 1 if (top and sem == 2) else ins
    assert np.array_equal(vol, want) and (vol == 13).any() and (vol != 0).sum() > 500
    with pytest.raises(RuntimeError):
        po.maps_to_volume(np.full((4, 4), 5, np.int16), td[:4, :4], bu[:4, :4], np.ones((4, 4), bool), scales[:5], depth)
