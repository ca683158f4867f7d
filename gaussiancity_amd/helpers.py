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
    # This build's wrapper takes the crop itself (GaussianRasterizerWrapper.forward(crop=...): only the window is
    # stored and differentiated); any other callable gets the reference's render-then-slice.
    takes_crop = getattr(rasterizator, "forward", None) is not None and "crop" in getattr(
        rasterizator.forward, "__code__", type("", (), {"co_varnames": ()})).co_varnames
    frames = []
    for i in range(gs_points.size(0)):
        box = crop_bboxes[i] if crop_bboxes is not None else None
        if box is not None and takes_crop:
            img = rasterizator(gs_points[i], cam_pos[i], cam_quat[i], crop=(box["x"], box["y"], box["w"], box["h"]))
        else:
            img = rasterizator(gs_points[i], cam_pos[i], cam_quat[i])
            if box is not None:
                img = img[:, box["y"]: box["y"] + box["h"], box["x"]: box["x"] + box["w"]]
        frames.append(img)
    # B == 1 (GaussianCity's batch size): the stacked batch is a view of the frame, not a copy
    return frames[0].unsqueeze(0) if len(frames) == 1 else torch.stack(frames, dim=0)
