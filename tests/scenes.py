"""Small seeded scenes + cameras shared by the CPU and GPU tests (no reference code, no GPU)."""
import math

import numpy as np
import torch

from gaussiancity_amd import synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper


def camera(W, H, pose_index=3, radius=60.0, altitude=50.0, n_poses=24, device="cpu"):
    """GaussianRasterizationSettings produced by the wrapper's own camera math."""
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=torch.device(device))
    pos, quat = synth.orbit_poses(n_poses, radius, altitude)[pose_index]
    return wr._get_gaussian_rasterization_settings(pos, quat)


def blob_scene(P, seed, sh_degree=3, spread=40.0, smin=0.5, smax=6.0, omin=0.05, omax=1.0):
    """Anisotropic rotated Gaussians around the orbit centre, sized to cover several pixels
    when seen from camera(radius=60, altitude=50)."""
    rng = np.random.default_rng(seed)
    xyz = np.empty((P, 3), np.float32)
    xyz[:, 0] = 1024 + rng.uniform(-spread, spread, P)
    xyz[:, 1] = 1024 + rng.uniform(-spread, spread, P)
    xyz[:, 2] = rng.uniform(0, spread * 0.5, P)
    scales = np.exp(rng.uniform(math.log(smin), math.log(smax), (P, 3))).astype(np.float32)
    rot = rng.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    rot *= rng.uniform(0.8, 1.2, (P, 1))  # the rasterizer never normalises quaternions
    M = (sh_degree + 1) ** 2
    shs = np.empty((P, M, 3), np.float32)
    shs[:, 0] = rng.normal(0.3, 0.8, (P, 3))
    shs[:, 1:] = rng.normal(0, 0.25, (P, M - 1, 3))
    return dict(means3D=xyz, scales=scales, rotations=rot.astype(np.float32),
                opacities=rng.uniform(omin, omax, (P, 1)).astype(np.float32),
                colors_precomp=rng.uniform(-1, 1, (P, 3)).astype(np.float32), shs=shs,
                sh_degree=sh_degree)


def settings_kwargs(rs):
    """GaussianRasterizationSettings -> keyword arguments of oracle.Frame (numpy)."""
    return dict(img_h=rs.img_h, img_w=rs.img_w, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy,
                bg=rs.bg.cpu().numpy(), scale_modifier=rs.scale_modifier,
                view_matrix=rs.view_matrix.cpu().numpy(), proj_matrix=rs.proj_matrix.cpu().numpy(),
                sh_degree=rs.sh_degree, campos=rs.campos.cpu().numpy())
