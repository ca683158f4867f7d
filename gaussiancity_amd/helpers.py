"""Glue between a generator's outputs and the rasterizer: counterparts of the two helpers
GaussianCity calls around the hot path (reference utils/helpers.py:226-247 and :250-270).
"""
import torch


def get_gaussian_points(xyz, scales, attrs):
    """Build the [B,N,14] Gaussian tensor: xyz(3) opacity(1) scale(3) rotation(4) rgb(3).

    Same contract as utils/helpers.py:226-247, including its in-place updates: `xyz` is
    shifted by attrs["xyz"] and `scales` multiplied by attrs["scale"] in place; opacity
    defaults to 1; rotations are the identity quaternion (1,0,0,0).
    """
    B, N = xyz.size(0), xyz.size(1)
    if "xyz" in attrs:
        xyz += attrs["xyz"]
    if "scale" in attrs:
        scales *= attrs["scale"]
    opacity = attrs["opacity"] if "opacity" in attrs else torch.ones((B, N, 1), device=xyz.device)
    rotations = torch.zeros((B, N, 4), device=xyz.device)
    rotations[..., 0] = 1.0
    return torch.cat((xyz, opacity, scales, rotations, attrs["rgb"]), dim=-1)


def get_gaussian_rasterization(gs_points, rasterizator, cam_pos, cam_quat, crop_bboxes=None):
    """Render each frame of the batch, optionally crop, and stack (utils/helpers.py:250-270).

    Frames are independent, so this loop is also the unit that shards one-per-GPU
    (gaussiancity_amd.frames).
    """
    frames = []
    for i in range(gs_points.size(0)):
        img = rasterizator(gs_points[i], cam_pos[i], cam_quat[i])
        if crop_bboxes is not None:
            box = crop_bboxes[i]
            img = img[:, box["y"]: box["y"] + box["h"], box["x"]: box["x"] + box["w"]]
        frames.append(img)
    return torch.stack(frames, dim=0)
