"""Row f4: round trips of the on-disk formats (no GPU)."""
import numpy as np
import torch

from gaussiancity_amd import formats as F
from gaussiancity_amd.grid_encoder import GridEncoder


def test_points_pkl_and_camera_csv_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    pts = rng.integers(0, 300, (50, 5)).astype(np.int16)
    vpm = rng.integers(-1, 50, (12, 20)).astype(np.int64)
    prj = {"INS": rng.integers(0, 9, (8, 8)).astype(np.int16), "TD_HF": np.zeros((8, 8), np.int16)}
    p = str(tmp_path / "0007.pkl")
    F.write_points_pkl(p, prj, vpm, vpm >= 0, pts)
    d = F.read_points_pkl(p)
    assert sorted(d) == sorted(F.POINT_KEYS) and np.array_equal(d["pts"], pts) and np.array_equal(d["vpm"], vpm)
    assert d["msk"].dtype == bool and np.array_equal(d["prj"]["INS"], prj["INS"])
    poses = [dict(id=i, tx=1.5 * i, ty=-2.0, tz=640.0, qx=0.0, qy=0.1 * i, qz=0.0, qw=1.0) for i in range(4)]
    c = str(tmp_path / "CameraPoses.csv")
    F.write_camera_poses_csv(c, poses)
    assert open(c).readline().strip() == "id,tx,ty,tz,qx,qy,qz,qw"
    rows = F.read_camera_poses_csv(c)
    assert sorted(rows) == [0, 1, 2, 3] and rows[2]["tx"] == "3.0"       # values stay strings, as upstream
    pos, quat = F.pose_arrays(rows[3])
    assert pos.dtype == np.float32 and np.allclose(pos, [4.5, -2.0, 640.0]) and np.allclose(quat, [0, 0.3, 0, 1])


def test_checkpoint_round_trip_loads_into_grid_encoder(tmp_path):
    enc = GridEncoder(in_channels=3, n_levels=4, lvl_channels=2, desired_resolution=64, log2_hashmap_size=8)
    sd = {"pos_encoder." + k: v for k, v in enc.state_dict().items()}
    p = str(tmp_path / "ckpt-last.pth")
    F.save_checkpoint(p, {"NETWORK": {"GAUSSIAN": {"HASH_GRID_N_LEVELS": 4}}}, 7, sd, gaussian_d={"w": torch.ones(2)})
    ck = F.load_checkpoint(p)
    assert ck["epoch_index"] == 7 and "gaussian_d" in ck
    enc2 = GridEncoder(in_channels=3, n_levels=4, lvl_channels=2, desired_resolution=64, log2_hashmap_size=8)
    enc2.load_state_dict({k[len("pos_encoder."):]: v for k, v in ck["gaussian_g"].items()})
    assert torch.equal(enc2.embeddings, enc.embeddings) and torch.equal(enc2.offsets, enc.offsets)
