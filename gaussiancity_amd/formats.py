"""On-disk formats either side of the hot paths, so that real GaussianCity
frames and checkpoints can be replayed through this build when the datasets are available (they are
external downloads; nothing here needs them).  Plain Python I/O -- no GPU involved.

  Points/%04d.pkl      dict {prj, vpm, msk, pts}           scripts/dataset_generator.py:1672-1686
  CameraPoses.csv      id, tx, ty, tz, qx, qy, qz, qw      scripts/dataset_generator.py:1137-1153, utils/io.py:104-107
  ckpt-*.pth           {cfg, epoch_index, gaussian_g[, gaussian_d]}   core/train.py:376-387
"""
import csv
import pickle

import numpy as np

POINT_KEYS = ("prj", "vpm", "msk", "pts")
POSE_FIELDS = ("id", "tx", "ty", "tz", "qx", "qy", "qz", "qw")


def write_points_pkl(path, prj, vpm, msk, pts):
    """prj: dict of local BEV maps, vpm: [H,W] visible-point map (-1 = none), msk: [H,W] bool, pts: [N,5] int16."""
    with open(path, "wb") as fp:
        pickle.dump({"prj": prj, "vpm": vpm, "msk": msk, "pts": pts}, fp)


def read_points_pkl(path):
    with open(path, "rb") as fp:
        d = pickle.load(fp)
    missing = [k for k in POINT_KEYS if k not in d]
    if missing:
        raise KeyError("%s: missing keys %s" % (path, missing))
    return d


def write_camera_poses_csv(path, cam_poses):
    """cam_poses: iterable of dicts with POSE_FIELDS (save_camera_poses upstream)."""
    with open(path, "w", newline="") as fp:
        w = csv.DictWriter(fp, fieldnames=list(POSE_FIELDS))
        w.writeheader()
        w.writerows(cam_poses)


def read_camera_poses_csv(path):
    """{id: row dict of strings} exactly as utils/io.py:104-107 returns it."""
    with open(path) as fp:
        return {int(r["id"]): r for r in csv.DictReader(fp)}


def pose_arrays(row):
    """One CSV row -> (cam_pos float32 [3], cam_quat float32 [4] in (x, y, z, w)), utils/datasets.py:119-127."""
    return (np.array([row["tx"], row["ty"], row["tz"]], dtype=np.float32),
            np.array([row["qx"], row["qy"], row["qz"], row["qw"]], dtype=np.float32))


def save_checkpoint(path, cfg, epoch_index, gaussian_g, gaussian_d=None):
    import torch
    ckpt = {"cfg": cfg, "epoch_index": epoch_index, "gaussian_g": gaussian_g}
    if gaussian_d is not None:
        ckpt["gaussian_d"] = gaussian_d
    torch.save(ckpt, path)


def load_checkpoint(path, map_location="cpu"):
    """Returns the dict; `gaussian_g` is the generator's state_dict -- its `pos_encoder.embeddings` /
    `pos_encoder.offsets` entries load into gaussiancity_amd.grid_encoder.GridEncoder unchanged."""
    import torch
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    for k in ("cfg", "epoch_index", "gaussian_g"):
        if k not in ckpt:
            raise KeyError("%s: not a GaussianCity checkpoint (missing %r)" % (path, k))
    return ckpt


def replay_visible_points(points_pkl, pose_row, cam_rig, null_class_id=0):
    """Recomputes the stored `vpm` of one dataset frame with this build's visibility path and returns
    (vp_map, stored_vpm, fraction_equal) -- the end-to-end check to run once real data is at hand."""
    from . import points as P
    d = read_points_pkl(points_pkl) if isinstance(points_pkl, str) else points_pkl
    pts = np.asarray(d["pts"], np.int16)
    scales = np.repeat(pts[:, [3]], 3, axis=1)
    cam_pos, cam_quat = pose_arrays(pose_row)
    vp, _ = P.get_visible_points(pts, scales, cam_rig, cam_pos.astype(np.float64), cam_quat.astype(np.float64), null_class_id)
    stored = np.asarray(d["vpm"])
    return vp, stored, float((vp == stored).mean()) if vp.shape == stored.shape else 0.0
