"""`gcity-synth-v1` -- deterministic synthetic scenes and cameras for tests and bench.py
(definitions: SURVEY.md section 8d; there is no network, so no real GaussianCity data).

Everything is generated on the host with numpy's PCG64 (`default_rng(seed)`) in fp32, and the
camera always goes through GaussianRasterizerWrapper's own math so the drop-in boundary is
exercised.  Scene families:

  S-rand(P, seed)  uniform box of anisotropic, rotated, semi-transparent Gaussians (C1, C2)
  S-city(P, seed)  BEV-shell-like point cloud: 70 % ground, 30 % facade/roof, identity
                   rotation, opacity 1, scales from {1,2,4}*0.45 (C3, C5) -- what GaussianCity
                   actually feeds the rasterizer (utils/helpers.py:212-246,
                   scripts/inference.py:37)
"""
import math

import numpy as np

GE_FOCAL_960 = 1528.1469407006614  # config.py:36 (CAM_K for SENSOR_SIZE (960,540))
PROJ_SIZE = 2048                   # config.py:40


def intrinsics(W, H):
    """K = [[f,0,W/2],[0,f,H/2],[0,0,1]] with the GoogleEarth focal length scaled to W."""
    f = GE_FOCAL_960 * W / 960.0
    return np.array([[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]], dtype=np.float64)


def quat_from_look_at(cam_pos, look_at):
    """(qx,qy,qz,qw) whose rotation columns are [Forward|Right|Up]
    (scripts/dataset_generator.py:1071-1085)."""
    import scipy.spatial.transform
    fwd = np.asarray(look_at, dtype=np.float64) - np.asarray(cam_pos, dtype=np.float64)
    fwd /= np.linalg.norm(fwd)
    right = np.cross(np.array([0.0, 0.0, 1.0]), fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    R = np.stack([fwd, right, up], axis=1)
    return scipy.spatial.transform.Rotation.from_matrix(R).as_quat()


def orbit_poses(n=24, radius=512.0, altitude=640.0, centre=(PROJ_SIZE // 2, PROJ_SIZE // 2)):
    """The inference orbit (scripts/inference.py:168-199) with radius/altitude fixed instead
    of drawn at random.  Returns a list of (position[3], quaternion[4]) float64 arrays."""
    cx, cy = centre
    poses = []
    for i in range(n):
        theta = 2 * math.pi / n * i
        pos = np.array([cx + radius * math.cos(theta), cy + radius * math.sin(theta), altitude])
        poses.append((pos, quat_from_look_at(pos, (cx, cy, 1.0))))
    return poses


def _sh(rng, P, degree):
    M = (degree + 1) ** 2
    shs = np.empty((P, M, 3), dtype=np.float32)
    shs[:, 0, :] = rng.normal(0.0, 0.5, size=(P, 3))
    if M > 1:
        shs[:, 1:, :] = rng.normal(0.0, 0.1, size=(P, M - 1, 3))
    return shs


def s_rand(P, seed, sh_degree=0):
    """S-rand: xyz ~ U([512,1536]^2 x [0,256]); scale = exp(U(ln .3, ln 3)) per axis;
    rot = normalize(N(0,1)^4); opacity ~ U(.05,1); colours U(-1,1) (models/generator.py:421-422);
    SH DC ~ N(0,.5), higher bands ~ N(0,.1)."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((P, 3), dtype=np.float32)
    xyz[:, 0] = rng.uniform(512, 1536, P)
    xyz[:, 1] = rng.uniform(512, 1536, P)
    xyz[:, 2] = rng.uniform(0, 256, P)
    scales = np.exp(rng.uniform(math.log(0.3), math.log(3.0), size=(P, 3))).astype(np.float32)
    rot = rng.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opacity = rng.uniform(0.05, 1.0, size=(P, 1)).astype(np.float32)
    colors = rng.uniform(-1.0, 1.0, size=(P, 3)).astype(np.float32)
    shs = _sh(rng, P, sh_degree)
    return dict(means3D=xyz, scales=scales, rotations=rot.astype(np.float32), opacities=opacity,
                colors_precomp=colors, shs=shs, sh_degree=sh_degree)


def s_city(P, seed, sh_degree=3):
    """S-city: 70 % ground (z=0, scale (s,s,1)), 30 % facade/roof (integer x,y, z in 0..200,
    scale (s,s,s)); s in {1,2,4}*0.45 with prob (.6,.3,.1); identity rotation; opacity 1."""
    rng = np.random.default_rng(seed)
    n_ground = int(P * 0.7)
    xyz = np.empty((P, 3), dtype=np.float32)
    xyz[:, 0] = rng.uniform(0, PROJ_SIZE, P)
    xyz[:, 1] = rng.uniform(0, PROJ_SIZE, P)
    xyz[:n_ground, 2] = 0.0
    xyz[n_ground:, :2] = np.floor(xyz[n_ground:, :2])
    xyz[n_ground:, 2] = rng.integers(0, 201, P - n_ground)
    s = (rng.choice(np.array([1.0, 2.0, 4.0]), size=P, p=[0.6, 0.3, 0.1]) * 0.45).astype(np.float32)
    scales = np.stack([s, s, s], axis=1)
    scales[:n_ground, 2] = 1.0
    rot = np.zeros((P, 4), dtype=np.float32)
    rot[:, 0] = 1.0
    opacity = np.ones((P, 1), dtype=np.float32)
    colors = rng.uniform(-1.0, 1.0, size=(P, 3)).astype(np.float32)
    shs = _sh(rng, P, sh_degree)
    return dict(means3D=xyz, scales=scales, rotations=rot, opacities=opacity,
                colors_precomp=colors, shs=shs, sh_degree=sh_degree)


def s_dense(P, seed, sh_degree=3, sigma=110.0):
    """S-dense: a general-3DGS-like stress scene (NOT a GaussianCity workload): anisotropic, rotated,
    mostly translucent Gaussians (opacity U(.02,.5)) piled up around the orbit centre with a normal density
    (x,y ~ N(1024, sigma), z ~ U(0,80)), scales exp(U(ln .6, ln 8)) per axis.  From the inference orbit this
    gives thousands of list entries per tile, the central tiles beyond the 4096-entry LDS sort capacity."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((P, 3), dtype=np.float32)
    xyz[:, 0] = PROJ_SIZE // 2 + rng.normal(0, sigma, P)
    xyz[:, 1] = PROJ_SIZE // 2 + rng.normal(0, sigma, P)
    xyz[:, 2] = rng.uniform(0, 80, P)
    scales = np.exp(rng.uniform(math.log(0.6), math.log(8.0), size=(P, 3))).astype(np.float32)
    rot = rng.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opacity = rng.uniform(0.02, 0.5, size=(P, 1)).astype(np.float32)
    colors = rng.uniform(-1.0, 1.0, size=(P, 3)).astype(np.float32)
    return dict(means3D=xyz, scales=scales, rotations=rot.astype(np.float32), opacities=opacity,
                colors_precomp=colors, shs=_sh(rng, P, sh_degree), sh_degree=sh_degree)


def grad_image(W, H, seed):
    """dL/d(out_color) ~ N(0,1) [3,H,W] with a fixed seed (backward tests / bench)."""
    return np.random.default_rng(seed + 77).normal(size=(3, H, W)).astype(np.float32)


# BASELINE.json configs C1..C5 (SURVEY.md section 8d "Config mapping")
CONFIGS = {
    "C1": dict(scene="s_rand", P=10_000, W=256, H=256, sh_degree=0, seed=1001, backward=False),
    "C2": dict(scene="s_rand", P=500_000, W=640, H=448, sh_degree=3, seed=1002, backward=True),
    "C3": dict(scene="s_city", P=5_000_000, W=1920, H=1080, sh_degree=3, seed=1003, backward=False),
    "C4": dict(scene="s_city", P=16_384, W=960, H=540, sh_degree=0, seed=1004, backward=True,
               crop=(640, 448), precomp_color=True),
    "C5": dict(scene="s_city", P=20_000_000, W=3840, H=2160, sh_degree=3, seed=1005, backward=False),
    # not a BASELINE config: dense general-3DGS-like stress scene (long tile lists), informational
    "D1": dict(scene="s_dense", P=2_700_000, W=1920, H=1080, sh_degree=3, seed=1006, backward=False,
               sigma=330.0),
}


def make_scene(name, P=None):
    cfg = dict(CONFIGS[name])
    if P is not None:
        cfg["P"] = int(P)
    if cfg["scene"] == "s_dense":
        scene = s_dense(cfg["P"], cfg["seed"], cfg["sh_degree"], cfg.get("sigma", 110.0))
    else:
        gen = s_rand if cfg["scene"] == "s_rand" else s_city
        scene = gen(cfg["P"], cfg["seed"], cfg["sh_degree"])
    return cfg, scene


# ------------------------------------------------------------------------------------------------
# gcity-layout-v1: synthetic BEV city layout for the point-generation / visibility path (SURVEY.md
# section 8 row f2).  Shapes, dtypes, class ids, per-class scales and the instance-id conventions are
# the GOOGLE_EARTH ones of the reference (scripts/dataset_generator.py:42-118, 984-1004: buildings
# carry even instance ids from 100, roof = facade id + 1; `PTS` = per-class stride grid masked by the
# class, :198-221); the content is seeded noise, not OSM data.
LAYOUT_CLASSES = {"NULL": 0, "ROAD": 1, "BLDG_FACADE": 2, "GREEN_LANDS": 3, "CONSTRUCTION": 4, "WATER": 5,
                  "ZONE": 6, "BLDG_ROOF": 7}
LAYOUT_SCALES = {"ROAD": 2, "BLDG_FACADE": 1, "BLDG_ROOF": 1, "GREEN_LANDS": 2, "CONSTRUCTION": 1, "WATER": 4,
                 "ZONE": 2}
LAYOUT_SEG_INS = {"BLDG_INS_MIN_ID": 100, "ROOF_INS_OFFSET": 1, "BLDG_FACADE_SEMANTIC_ID": 2,
                  "BLDG_ROOF_SEMANTIC_ID": 7, "CAR_INS_MIN_ID": 32767, "CAR_SEMANTIC_ID": 32767}


def layout_point_map(seg_map, classes=LAYOUT_CLASSES, scales=LAYOUT_SCALES):
    """scripts/dataset_generator.py:198-221 (_get_point_maps): stride-`scale` lattice per class."""
    pts = np.zeros(seg_map.shape, bool)
    inv = {v: k for k, v in classes.items()}
    for c in np.unique(seg_map):
        name = inv[int(c)]
        if name == "NULL":
            continue
        s = scales[name]
        lattice = np.zeros(seg_map.shape, bool)
        lattice[::s, ::s] = True
        pts |= lattice & (seg_map == c)
    return pts


def s_layout(size=2048, seed=2001, block=128, road=16, elevation=4, max_height=300):
    """Returns dict(INS, SEG, TD_HF, BU_HF int16 [size,size]; PTS bool): a road grid, blocks of green
    land / zone / water / construction, and rectangular buildings with heights up to `max_height`."""
    rng = np.random.default_rng(seed)
    S = int(size)
    seg = np.full((S, S), LAYOUT_CLASSES["GREEN_LANDS"], np.int16)
    ins = seg.copy()
    td = np.full((S, S), elevation, np.int16)
    nb = (S + block - 1) // block
    base_choices = np.array([LAYOUT_CLASSES[k] for k in ("GREEN_LANDS", "ZONE", "WATER", "CONSTRUCTION")], np.int16)
    next_ins = LAYOUT_SEG_INS["BLDG_INS_MIN_ID"]
    for by in range(nb):
        for bx in range(nb):
            y0, x0 = by * block + road, bx * block + road
            y1, x1 = min(S, (by + 1) * block), min(S, (bx + 1) * block)
            if y0 >= y1 or x0 >= x1:
                continue
            base = base_choices[rng.choice(4, p=[0.45, 0.3, 0.1, 0.15])]
            seg[y0:y1, x0:x1] = base
            ins[y0:y1, x0:x1] = base
            if base == LAYOUT_CLASSES["GREEN_LANDS"]:  # rough vegetation: height noise -> many border columns
                td[y0:y1, x0:x1] = elevation + rng.integers(0, 4, (y1 - y0, x1 - x0))
            elif base == LAYOUT_CLASSES["WATER"]:
                td[y0:y1, x0:x1] = elevation - 1
                continue
            for _ in range(int(rng.integers(1, 5))):
                bw, bh = int(rng.integers(12, 64)), int(rng.integers(12, 64))
                yy = int(rng.integers(y0, max(y0 + 1, y1 - bh)))
                xx = int(rng.integers(x0, max(x0 + 1, x1 - bw)))
                ye, xe = min(y1, yy + bh), min(x1, xx + bw)
                if next_ins + 2 >= 16384:
                    break
                height = int(elevation + rng.integers(8, max_height))
                seg[yy:ye, xx:xe] = LAYOUT_CLASSES["BLDG_FACADE"]
                ins[yy:ye, xx:xe] = next_ins
                td[yy:ye, xx:xe] = height
                next_ins += 2
    # roads last so that they cut cleanly between blocks
    for k in range(0, S, block):
        seg[k:k + road, :] = LAYOUT_CLASSES["ROAD"]; ins[k:k + road, :] = LAYOUT_CLASSES["ROAD"]; td[k:k + road, :] = elevation
        seg[:, k:k + road] = LAYOUT_CLASSES["ROAD"]; ins[:, k:k + road] = LAYOUT_CLASSES["ROAD"]; td[:, k:k + road] = elevation
    bu = np.zeros((S, S), np.int16)
    return dict(INS=ins, SEG=seg, TD_HF=td, BU_HF=bu, PTS=layout_point_map(seg))


def layout_camera(size=2048, W=960, H=540, pose=3, n_poses=24):
    """cam_rig / pose for get_visible_points on an s_layout map: the reference's intrinsics
    (config.py:36-37) and inference orbit scaled to the map (scripts/inference.py:168-199)."""
    K = intrinsics(W, H)
    rig = {"intrinsics": [float(v) for v in K.reshape(-1)], "sensor_size": [W, H]}
    c = size / 2.0
    pos, quat = orbit_poses(n_poses, radius=size / 4.0, altitude=size * 0.3125, centre=(c, c))[pose]
    return rig, np.asarray(pos, np.float64), np.asarray(quat, np.float64)
