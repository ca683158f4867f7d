#!/usr/bin/env python
"""Generates tests/golden/points_golden.npz with the REFERENCE's own footprint extruder, compiled from
/root/reference/extensions/footprint_extruder/footprint_extruder.cpp into oracle/_ref/ (oracle/Makefile,
target `ref`).  Run in the build container only (the reference does not exist on the GPU box):

    make -C oracle ref && python tests/golden/make_points_golden.py

The fixture holds inputs (BEV maps, flags, class tables) and the reference's outputs -- data only.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gaussiancity_amd import synth  # noqa: E402
from oracle import points_oracle as PO  # noqa: E402

KITTI_CLASSES = {"NULL": 0, "ROAD": 1, "BLDG_FACADE": 2, "CAR": 3, "VEGETATION": 4, "SKY": 5, "ZONE": 6, "BLDG_ROOF": 7}
KITTI_SCALES = {"ROAD": 2, "BLDG_FACADE": 1, "CAR": 1, "VEGETATION": 1, "SKY": 4, "ZONE": 2, "BLDG_ROOF": 1}
KITTI_SEG_INS = {"BLDG_INS_MIN_ID": 100, "ROOF_INS_OFFSET": 1, "BLDG_FACADE_SEMANTIC_ID": 2, "BLDG_ROOF_SEMANTIC_ID": 7,
                 "CAR_INS_MIN_ID": 10000, "CAR_SEMANTIC_ID": 3}


def cases():
    out = []
    # 1. a crop of the synthetic GOOGLE_EARTH-style layout (roads, blocks, buildings)
    L = synth.s_layout(192, 2101, block=64, road=8, max_height=60)
    out.append(("layout192", synth.LAYOUT_CLASSES, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"],
                L["BU_HF"], L["PTS"]))
    # 2. random noise maps: every pixel differs from its neighbours -> all columns are border columns
    rng = np.random.default_rng(7)
    H, W = 40, 56
    seg = rng.choice(np.array([1, 2, 3, 4, 5, 6], np.int16), size=(H, W))
    td = rng.integers(-2, 12, (H, W)).astype(np.int16)
    bu = rng.integers(-3, 4, (H, W)).astype(np.int16)  # includes bu > td columns (emit nothing)
    pts = rng.random((H, W)) < 0.6
    out.append(("noise", synth.LAYOUT_CLASSES, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu, pts))
    # 3. KITTI-style tables: car instances (>= 10000) map to the CAR class, buildings 100..9999
    H, W = 48, 48
    seg = np.full((H, W), 1, np.int16)
    seg[4:20, 6:30] = 100; seg[24:40, 10:26] = 2046; seg[30:36, 30:44] = 10004; seg[8:12, 34:40] = 4
    td = np.full((H, W), 2, np.int16)
    td[4:20, 6:30] = 17; td[24:40, 10:26] = 9; td[30:36, 30:44] = 5; td[8:12, 34:40] = 7
    bu = np.zeros((H, W), np.int16); bu[30:36, 30:44] = 2
    pts = synth.layout_point_map(np.where(seg >= 10000, 3, np.where(seg >= 100, 2, seg)).astype(np.int16),
                                 KITTI_CLASSES, KITTI_SCALES)
    out.append(("kitti", KITTI_CLASSES, KITTI_SCALES, KITTI_SEG_INS, seg, td, bu, pts))
    # 4. flat uniform map: only top (+ bottom) points, and the image-border rule
    H, W = 24, 20
    seg = np.full((H, W), 6, np.int16); td = np.full((H, W), 8, np.int16); bu = np.full((H, W), 1, np.int16)
    out.append(("flat", synth.LAYOUT_CLASSES, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu,
                np.ones((H, W), bool)))
    # 5. nothing to emit
    out.append(("empty", synth.LAYOUT_CLASSES, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, seg, td, bu,
                np.zeros((H, W), bool)))
    return out


def main():
    ref = PO.reference_extruder()
    if ref is None:
        raise SystemExit("oracle/_ref/footprint_extruder.so missing: run `make -C oracle ref` first")
    blob, meta = {}, {}
    for name, classes, scales, seg_ins, seg, td, bu, pts in cases():
        inv = {v: k for k, v in classes.items()}
        meta[name] = dict(classes=classes, scales=scales, seg_ins=seg_ins)
        for k, a in (("seg", seg), ("td", td), ("bu", bu), ("pts", pts)):
            blob["%s.%s" % (name, k)] = np.ascontiguousarray(a)
        for inc in (True, False):
            r = ref.get_points_from_projection(inc, inv, scales, seg_ins, np.ascontiguousarray(seg.astype(np.int16)),
                                               np.ascontiguousarray(td.astype(np.int16)),
                                               np.ascontiguousarray(bu.astype(np.int16)),
                                               np.ascontiguousarray(pts.astype(bool)))
            blob["%s.out.%d" % (name, int(inc))] = np.zeros((0, 5), np.uint16) if r is None else r
            print(name, inc, None if r is None else r.shape)
    blob["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "points_golden.npz"), **blob)


if __name__ == "__main__":
    main()
