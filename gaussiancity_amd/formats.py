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
    """prj: dict of local BEV maps; pts: [N,5] int16, ONLY the points visible in this view, in ascending order
    of their original index; vpm: [H,W] index into `pts` of the point each pixel sees -- upstream re-indexes with
    np.searchsorted over the visible ids (scripts/dataset_generator.py:1606-1617), so a pixel that sees nothing
    holds 0, not -1; msk: [H,W] bool."""
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


# cfg.DATASETS.<name>.SCALE / MAP_SIZE of the reference (config.py:45-46,73-74)
DATASET_CONSTANTS = {"GOOGLE_EARTH": {"SCALE": 1, "MAP_SIZE": 2048}, "KITTI_360": {"SCALE": 1, "MAP_SIZE": 0}}


def pose_arrays(row, scale=1, map_size=0):
    """One CSV row -> (cam_pos float32 [3], cam_quat float32 [4] in (x, y, z, w)) normalised to the map the way
    utils/datasets.py:119-127 and scripts/dataset_generator.py:1543-1549 do: position / SCALE, then x and y
    shifted by MAP_SIZE // 2.  The defaults leave the CSV values as they are."""
    pos = np.array([row["tx"], row["ty"], row["tz"]], dtype=np.float32) / scale
    pos[:2] += map_size // 2
    return pos, np.array([row["qx"], row["qy"], row["qz"], row["qw"]], dtype=np.float32)


def save_checkpoint(path, cfg, epoch_index, gaussian_g, gaussian_d=None):
    import torch
    ckpt = {"cfg": cfg, "epoch_index": epoch_index, "gaussian_g": gaussian_g}
    if gaussian_d is not None:
        ckpt["gaussian_d"] = gaussian_d
    torch.save(ckpt, path)


class _EasyDict(dict):
    """Attribute-access dict standing in for easydict.EasyDict while a checkpoint is unpickled on a machine without
    that package (upstream pickles its `cfg` as one: config.py:10, core/train.py:376-380)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _CheckpointPickle:
    """`pickle_module` for torch.load: the stock pickle machinery, except that the global `easydict.EasyDict`
    resolves to the installed class or, without the package, to the stand-in above.  Nothing is registered in
    sys.modules, so a later `import easydict` elsewhere in the process still fails (or finds the real package)
    instead of silently receiving a partial shim."""
    import pickle as _p
    __name__ = "pickle"
    load = staticmethod(_p.load)
    loads = staticmethod(_p.loads)
    dump = staticmethod(_p.dump)
    dumps = staticmethod(_p.dumps)
    Pickler = _p.Pickler
    HIGHEST_PROTOCOL = _p.HIGHEST_PROTOCOL
    DEFAULT_PROTOCOL = _p.DEFAULT_PROTOCOL
    UnpicklingError = _p.UnpicklingError
    PicklingError = _p.PicklingError

    class Unpickler(_p.Unpickler):
        def find_class(self, module, name):
            if module == "easydict" and name == "EasyDict":
                try:
                    import easydict
                    return easydict.EasyDict
                except ImportError:
                    return _EasyDict
            return super().find_class(module, name)


def load_checkpoint(path, map_location="cpu"):
    """Returns the dict; `gaussian_g` is the generator's state_dict -- its `pos_encoder.embeddings` /
    `pos_encoder.offsets` entries load into gaussiancity_amd.grid_encoder.GridEncoder unchanged.
    Upstream checkpoints carry an EasyDict `cfg`, i.e. arbitrary pickled globals: load files you trust
    (weights_only=False, as upstream's own torch.load calls)."""
    import torch
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_CheckpointPickle)
    for k in ("cfg", "epoch_index", "gaussian_g"):
        if k not in ckpt:
            raise KeyError("%s: not a GaussianCity checkpoint (missing %r)" % (path, k))
    return ckpt


def replay_visible_points(points_pkl, pose_row, cam_rig, dataset="GOOGLE_EARTH", null_class_id=0):
    """Recomputes the stored `vpm` of one dataset frame with this build's visibility path and returns
    (vp_map, stored_vpm, fraction_equal) -- the end-to-end check to run once real data is at hand.

    Follows how upstream produced the file (scripts/dataset_generator.py:1543-1549,1591-1617): the camera position
    is the CSV row divided by SCALE and shifted by MAP_SIZE // 2; cube sides are column 3 (get_point_scales without
    special classes); KITTI-360 maps are flipped left-right; the stored `pts` are only the points visible in the
    view (a first hit stays a first hit when hidden points are removed, so tracing them alone reproduces the map)
    and the stored `vpm` was re-indexed with np.searchsorted, which sends "no hit" (-1) to 0."""
    from . import points as P
    d = read_points_pkl(points_pkl) if isinstance(points_pkl, str) else points_pkl
    pts = np.asarray(d["pts"], np.int16)
    scales = np.repeat(pts[:, [3]], 3, axis=1)
    c = DATASET_CONSTANTS[dataset]
    cam_pos, cam_quat = pose_arrays(pose_row, c["SCALE"], c["MAP_SIZE"])
    vp, _ = P.get_visible_points(pts, scales, cam_rig, cam_pos.astype(np.float64), cam_quat.astype(np.float64), null_class_id)
    if dataset == "KITTI_360":
        vp = np.fliplr(vp)
    vp = np.where(vp < 0, 0, vp)  # what np.searchsorted(visible_ids, -1) yields upstream
    stored = np.asarray(d["vpm"])
    return vp, stored, float((vp == stored).mean()) if vp.shape == stored.shape else 0.0
