"""Shared helpers of the point-generation / visibility tests (CPU and GPU)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "points_golden.npz")


def golden_cases():
    """[(name, classes_inv, scales, seg_ins, seg, td, bu, pts, {True: out, False: out})] from the fixture
    written by tests/golden/make_points_golden.py with the reference's own extruder."""
    z = np.load(GOLDEN)
    meta = json.loads(bytes(z["meta"]).decode())
    out = []
    for name, m in meta.items():
        inv = {int(v): k for k, v in m["classes"].items()}
        out.append((name, inv, m["scales"], m["seg_ins"], z[name + ".seg"].astype(np.int16),
                    z[name + ".td"].astype(np.int16), z[name + ".bu"].astype(np.int16), z[name + ".pts"].astype(bool),
                    {True: z[name + ".out.1"], False: z[name + ".out.0"]}))
    return out


def random_points(rng, n, h, w, d, max_scale=4):
    """Random int16 points / scales incl. out-of-range rows and overlapping cubes; ids 1..n."""
    pts = np.stack([rng.integers(-2, w + 2, n), rng.integers(-2, h + 2, n), rng.integers(-2, d + 2, n)], 1).astype(np.int16)
    sc = rng.integers(1, max_scale + 1, (n, 3)).astype(np.int16)
    ids = np.arange(1, n + 1, dtype=np.int32)[:, None]
    return pts, ids, sc


def shell_volume(rng, h, w, d, n_boxes=6):
    """Hollow boxes of distinct ids in an int32 [h,w,d] volume (what extruded buildings look like)."""
    vol = np.zeros((h, w, d), np.int32)
    next_id = 8
    for b in range(n_boxes):
        y0, x0 = int(rng.integers(0, h - 6)), int(rng.integers(0, w - 6))
        y1, x1 = min(h, y0 + int(rng.integers(4, h // 2))), min(w, x0 + int(rng.integers(4, w // 2)))
        z1 = int(rng.integers(2, d))
        box = vol[y0:y1, x0:x1, 0:z1]
        ids = (next_id + np.arange(box.size, dtype=np.int32)).reshape(box.shape)  # unique per voxel
        next_id += box.size
        shell = np.zeros(box.shape, bool)
        shell[0], shell[-1], shell[:, 0], shell[:, -1], shell[:, :, -1] = True, True, True, True, True
        box[shell] = ids[shell]
    vol[:, :, 0] = np.where(vol[:, :, 0] == 0, 7, vol[:, :, 0])  # ground plane
    return vol


def camera_for_volume(h, w, d, rows, cols, variant=0):
    """(cam_ori, cam_dir, cam_up, cam_f, cam_c, img_dims) looking at the volume centre from outside
    (variant 0), from inside (1), or straight down an axis (2: zero ray-direction components)."""
    centre = np.array([h / 2, w / 2, d / 4], np.float32)
    if variant == 0:
        ori = np.array([-0.35 * h, -0.2 * w, 1.7 * d], np.float32)
    elif variant == 1:
        ori = np.array([0.31 * h + 0.25, 0.42 * w + 0.5, 0.8 * d], np.float32)
    else:
        ori = np.array([h / 2 + 0.5, w / 2 + 0.5, 3.0 * d], np.float32)
        centre = np.array([h / 2 + 0.5, w / 2 + 0.5, 0.0], np.float32)
    up = np.array([0, 0, 1], np.float32) if variant != 2 else np.array([0, 1, 0], np.float32)
    return ori, (centre - ori).astype(np.float32), up, float(0.9 * cols), [rows / 2.0, cols / 2.0], [rows, cols]
