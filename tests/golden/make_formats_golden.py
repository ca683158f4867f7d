"""Generates tests/golden/formats/* by running the REFERENCE's own Python for the on-disk formats (SURVEY.md
section 8 row f4).  Run in the build container only (needs /root/reference); the fixtures are data, the reference
sources never leave the container.

  CameraPoses.csv        written by scripts/dataset_generator.save_camera_poses            (:1137-1153)
  camera_poses_read.json what utils.io.IO.get returns for that file                        (utils/io.py:104-107)
  points_0007.pkl        the dict layout of scripts/dataset_generator.py:1672-1686, written here and
                         read back through utils.io.IO.get (round trip asserted at generation time)
  points_expect.npz      arrays of that frame + the camera normalisation utils/datasets.py:119-127 applies
  ckpt-last.pth          {cfg (EasyDict), epoch_index, gaussian_g, gaussian_d} as core/train.py:376-387 saves it

Modules the image lacks and the imported reference files do not use on these paths are stubbed as empty modules
(cv2, lxml, open3d, plyfile, the CUDA extensions); `easydict` -- needed by config.py -- is provided by the same
minimal shim the product installs when loading a checkpoint (gaussiancity_amd.formats._ensure_easydict).
"""
import json
import os
import pickle
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
OUT = os.path.join(HERE, "formats")


def main():
    sys.path.insert(0, ROOT)
    from gaussiancity_amd import formats as F
    F._ensure_easydict()
    for name in ("cv2", "open3d", "plyfile", "lxml", "lxml.etree", "footprint_extruder", "extensions.voxlib"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lxml"].etree = sys.modules["lxml.etree"]
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "scripts"))
    import utils.io as ref_io                      # the reference's reader
    import dataset_generator as ref_gen            # the reference's writers
    os.makedirs(OUT, exist_ok=True)

    # ---- CameraPoses.csv: reference writer -> reference reader
    rng = np.random.default_rng(404)
    poses = []
    for i in range(6):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        poses.append({"id": i, "tx": float(rng.uniform(-800, 800)), "ty": float(rng.uniform(-800, 800)),
                      "tz": float(rng.uniform(300, 900)), "qx": float(q[0]), "qy": float(q[1]), "qz": float(q[2]),
                      "qw": float(q[3])})
    csv_path = os.path.join(OUT, "CameraPoses.csv")
    ref_gen.save_camera_poses(csv_path, poses)
    rows = ref_io.IO.get(csv_path)
    json.dump({str(k): v for k, v in rows.items()}, open(os.path.join(OUT, "camera_poses_read.json"), "w"), indent=1)
    json.dump(poses, open(os.path.join(OUT, "camera_poses_in.json"), "w"), indent=1)

    # ---- Points/0007.pkl: the dict the generator dumps (:1672-1686), read back by the reference reader
    n = 40
    pts = np.concatenate([rng.integers(0, 2048, (n, 2)), rng.integers(0, 300, (n, 1)), rng.integers(1, 5, (n, 1)),
                          rng.integers(1, 300, (n, 1))], axis=1).astype(np.int16)
    raw_vp = rng.integers(-1, n, (9, 16)).astype(np.int32)                  # what get_visible_points returns
    vp_idx = np.sort(np.unique(raw_vp))
    vp_idx = vp_idx[vp_idx >= 0]
    vis_pts = pts[vp_idx]                                                    # :1599-1603
    vpm = np.searchsorted(vp_idx, raw_vp)                                    # :1613
    prj = {k: rng.integers(0, 200, (12, 12)).astype(np.int16) for k in ("INS", "SEG", "TD_HF", "BU_HF")}
    prj["PTS"] = rng.integers(0, 2, (12, 12)).astype(bool)
    msk = rng.integers(0, 2, (9, 16)).astype(bool)
    pkl_path = os.path.join(OUT, "points_0007.pkl")
    with open(pkl_path, "wb") as fp:
        pickle.dump({"prj": prj, "vpm": vpm, "msk": msk, "pts": vis_pts}, fp)
    back = ref_io.IO.get(pkl_path)
    assert sorted(back) == ["msk", "prj", "pts", "vpm"] and np.array_equal(back["vpm"], vpm)
    # camera normalisation of utils/datasets.py:119-127 on row 3, for both datasets' constants (config.py:45-46,73-74)
    from config import cfg
    norm = {}
    for ds in ("GOOGLE_EARTH", "KITTI_360"):
        Rt = rows[3]
        dcfg = cfg.DATASETS[ds]
        cam_pos = np.array([Rt["tx"], Rt["ty"], Rt["tz"]], dtype=np.float32) / dcfg.SCALE
        cam_pos[:2] += dcfg.MAP_SIZE // 2
        norm[ds + "_pos"] = cam_pos
        norm[ds + "_quat"] = np.array([Rt["qx"], Rt["qy"], Rt["qz"], Rt["qw"]], dtype=np.float32)
        norm[ds + "_const"] = np.array([dcfg.SCALE, dcfg.MAP_SIZE])
    np.savez(os.path.join(OUT, "points_expect.npz"), pts=vis_pts, vpm=vpm, msk=msk, raw_vp=raw_vp,
             **{"prj_" + k: v for k, v in prj.items()}, **norm)

    # ---- checkpoint as core/train.py:376-387 saves it (cfg is an EasyDict)
    from easydict import EasyDict
    small_cfg = EasyDict({"NETWORK": {"GAUSSIAN": {"N_FREQ_BANDS": 10, "SCALE_FACTOR": 0.65}},
                          "TRAIN": {"GAUSSIAN": {"DISCRIMINATOR": {"ENABLED": True}}}})
    torch.manual_seed(5)
    g_sd = {"pos_encoder.embeddings": torch.randn(40, 2), "pos_encoder.offsets": torch.tensor([0, 8, 24, 40], dtype=torch.int32),
            "mlp.0.weight": torch.randn(4, 3)}
    ckpt = {"cfg": small_cfg, "epoch_index": 123, "gaussian_g": g_sd}
    if small_cfg.TRAIN.GAUSSIAN.DISCRIMINATOR.ENABLED:
        ckpt["gaussian_d"] = {"conv.weight": torch.randn(2, 2)}
    torch.save(ckpt, os.path.join(OUT, "ckpt-last.pth"))
    np.savez(os.path.join(OUT, "ckpt_expect.npz"), **{k: v.numpy() for k, v in g_sd.items()},
             d_conv=ckpt["gaussian_d"]["conv.weight"].numpy())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
