#!/usr/bin/env python
"""Generates tests/golden/*.npz|json by IMPORTING the reference's Python half in this container
(/root/reference is read-only and does not exist on the GPU box, so only the resulting vectors
-- inputs and expected outputs, no reference source -- are committed).

What the reference can pin here: it has no tests and its native code
cannot be built (nvcc/CUB/GLM missing), but dgr/__init__.py and utils/helpers.py import fine
once `diff_gaussian_rasterization_ext` (and `plyfile`) are stubbed.  Captured:
  camera.npz     GaussianRasterizerWrapper camera math for 8 intrinsics/pose tuples + 24-pose orbit
  arg_order.json positional argument layout handed to the native forward/backward, and which
                 native gradient lands on which autograd input
  split_flip.npz 14-channel split + LR/UD flip behaviour with a fake native returning a ramp
  helpers.npz    utils.helpers.get_gaussian_points / get_gaussian_rasterization
  validation.json which argument combinations raise

Run:  python tests/golden/make_golden.py        (needs /root/reference)
"""
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


class Recorder(types.ModuleType):
    """Stand-in for the reference's native module: records calls, returns tagged tensors."""

    def __init__(self):
        super().__init__("diff_gaussian_rasterization_ext")
        self.fw_args = None
        self.bw_args = None
        self.ramp = None

    def rasterize_gaussians(self, *args):
        self.fw_args = args
        H, W = args[12], args[13]
        P = args[1].shape[0]
        color = self.ramp if self.ramp is not None else torch.zeros(3, H, W)
        return (7, color, torch.zeros(P, dtype=torch.int32), torch.zeros(5, dtype=torch.uint8),
                torch.zeros(6, dtype=torch.uint8), torch.zeros(7, dtype=torch.uint8))

    def rasterize_gaussians_backward(self, *args):
        self.bw_args = args
        P = args[1].shape[0]
        M = args[13].shape[1] if args[13].numel() else 0
        f = lambda shape, v: torch.full(shape, float(v))  # noqa: E731
        # order of dgr/rasterize_points.cu:153-154, each filled with its 1-based position
        return (f((P, 3), 1), f((P, 3), 2), f((P, 1), 3), f((P, 3), 4), f((P, 6), 5), f((P, M, 3), 6),
                f((P, 3), 7), f((P, 4), 8))

    def mark_visible(self, *a):
        raise NotImplementedError


def describe(a):
    if isinstance(a, torch.Tensor):
        return {"kind": "tensor", "shape": list(a.shape), "dtype": str(a.dtype).replace("torch.", ""),
                "tag": (float(a.reshape(-1)[0]) if a.numel() else None)}
    if isinstance(a, bool):
        return {"kind": "bool", "value": a}
    if isinstance(a, int):
        return {"kind": "int", "value": a}
    if isinstance(a, float):
        return {"kind": "float", "value": a}
    return {"kind": type(a).__name__}


def main():
    rec = Recorder()
    sys.modules["diff_gaussian_rasterization_ext"] = rec
    sys.modules.setdefault("plyfile", types.ModuleType("plyfile"))
    sys.path.insert(0, REF)
    import extensions.diff_gaussian_rasterization as dgr  # the reference's own Python
    import utils.helpers as ref_helpers
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from gaussiancity_amd import synth  # only for pose generation inputs (stored in the fixture)

    cpu = torch.device("cpu")

    # ------------------------------------------------------------------ camera.npz
    cam = {}
    tuples = []
    K_ge = np.array([1528.1469407006614, 0, 480, 0, 1528.1469407006614, 270, 0, 0, 1]).reshape(3, 3)
    K_kitti = np.array([552.554261, 0, 682.049453, 0, 552.554261, 238.769549, 0, 0, 1]).reshape(3, 3)
    rng = np.random.default_rng(2024)
    for i, (K, ss) in enumerate([(K_ge, (960, 540)), (K_kitti, (1408, 376)), (synth.intrinsics(640, 448), (640, 448)),
                                 (synth.intrinsics(1920, 1080), (1920, 1080)), (K_ge.astype(np.float32), (960, 540)),
                                 (synth.intrinsics(256, 256), (256, 256)), (synth.intrinsics(3840, 2160), (3840, 2160)),
                                 (K_kitti, (1408, 376))]):
        pos = rng.uniform(-500, 1500, 3)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        tuples.append((K, ss, pos, q))
    for i, (K, ss, pos, q) in enumerate(tuples):
        wr = dgr.GaussianRasterizerWrapper(K, ss, device=cpu)
        rs = wr._get_gaussian_rasterization_settings(pos, q)
        cam["K_%d" % i] = np.asarray(K)
        cam["sensor_%d" % i] = np.asarray(ss)
        cam["pos_%d" % i] = pos
        cam["quat_%d" % i] = q
        cam["fov_%d" % i] = np.array([wr.fov_x, wr.fov_y], dtype=np.float64)
        cam["P_%d" % i] = wr.P.numpy()
        cam["w2c_%d" % i] = wr._get_w2c_matrix(pos, q).numpy()
        cam["tanfov_%d" % i] = np.array([rs.tanfovx, rs.tanfovy], dtype=np.float64)
        cam["view_%d" % i] = rs.view_matrix.numpy()
        cam["proj_%d" % i] = rs.proj_matrix.numpy()
        cam["campos_%d" % i] = rs.campos.numpy()
        cam["hw_%d" % i] = np.array([rs.img_h, rs.img_w])
        assert rs.sh_degree == 0 and rs.scale_modifier == 1.0 and not rs.prefiltered and not rs.debug
        assert float(rs.bg.abs().sum()) == 0.0
    cam["n_tuples"] = np.array(len(tuples))
    # the 24-pose inference orbit (radius/altitude fixed), GoogleEarth intrinsics
    wr = dgr.GaussianRasterizerWrapper(K_ge, (960, 540), device=cpu)
    poses = synth.orbit_poses()
    cam["orbit_pos"] = np.stack([p for p, _ in poses])
    cam["orbit_quat"] = np.stack([q for _, q in poses])
    views, projs, camposs = [], [], []
    for p, q in poses:
        rs = wr._get_gaussian_rasterization_settings(p, q)
        views.append(rs.view_matrix.numpy())
        projs.append(rs.proj_matrix.numpy())
        camposs.append(rs.campos.numpy())
    cam["orbit_view"], cam["orbit_proj"], cam["orbit_campos"] = np.stack(views), np.stack(projs), np.stack(camposs)
    np.savez_compressed(os.path.join(HERE, "camera.npz"), **cam)

    # ------------------------------------------------------------------ arg_order.json
    P, M, H, W = 5, 4, 6, 8
    tag = lambda shape, v: torch.full(shape, float(v), requires_grad=True)  # noqa: E731
    inputs = dict(means3D=tag((P, 3), 101), means2D=tag((P, 3), 102), sh=tag((P, M, 3), 103),
                  colors_precomp=torch.Tensor([]), opacities=tag((P, 1), 105), scales=tag((P, 3), 106),
                  rotations=tag((P, 4), 107), cov3Ds_precomp=torch.Tensor([]))
    rs = dgr.GaussianRasterizationSettings(
        img_h=H, img_w=W, tanfovx=0.25, tanfovy=0.125, bg=torch.full((3,), 201.0), scale_modifier=1.5,
        view_matrix=torch.full((4, 4), 202.0), proj_matrix=torch.full((4, 4), 203.0), sh_degree=1,
        campos=torch.full((3,), 204.0), prefiltered=False, debug=False)
    color, radii = dgr.RasterizeGaussiansFunction.apply(
        inputs["means3D"], inputs["means2D"], inputs["sh"], inputs["colors_precomp"], inputs["opacities"],
        inputs["scales"], inputs["rotations"], inputs["cov3Ds_precomp"], rs)
    color.sum().backward()
    grads = {k: (float(v.grad.reshape(-1)[0]) if (v.requires_grad and v.grad is not None) else None)
             for k, v in inputs.items()}
    order = {
        "settings_fields": list(dgr.GaussianRasterizationSettings._fields),
        "input_tags": {k: (float(v.reshape(-1)[0]) if v.numel() else None) for k, v in inputs.items()},
        "forward_args": [describe(a) for a in rec.fw_args],
        "backward_args": [describe(a) for a in rec.bw_args],
        "grad_position_to_input": grads,   # value = 1-based position in the native return tuple
        "forward_returns": ["color", "radii"],
    }
    json.dump(order, open(os.path.join(HERE, "arg_order.json"), "w"), indent=1)

    # ------------------------------------------------------------------ validation.json
    r = dgr.GaussianRasterizer(rs)
    t = torch.zeros
    combos = []
    for name, kw in [
        ("neither_colour", dict(scales=t(P, 3), rotations=t(P, 4))),
        ("both_colour", dict(shs=t(P, M, 3), colors_precomp=t(P, 3), scales=t(P, 3), rotations=t(P, 4))),
        ("no_cov", dict(shs=t(P, M, 3))),
        ("scale_only", dict(shs=t(P, M, 3), scales=t(P, 3))),
        ("rot_only", dict(shs=t(P, M, 3), rotations=t(P, 4))),
        ("scale_rot_and_cov", dict(shs=t(P, M, 3), scales=t(P, 3), rotations=t(P, 4), cov3D_precomp=t(P, 6))),
        ("scale_and_cov", dict(shs=t(P, M, 3), scales=t(P, 3), cov3D_precomp=t(P, 6))),
        ("ok_sh_scale_rot", dict(shs=t(P, M, 3), scales=t(P, 3), rotations=t(P, 4))),
        ("ok_colour_cov", dict(colors_precomp=t(P, 3), cov3D_precomp=t(P, 6))),
    ]:
        try:
            r(means3D=t(P, 3), means2D=t(P, 3), opacities=t(P, 1), **kw)
            combos.append({"name": name, "keys": sorted(kw), "raises": None})
        except Exception as ex:  # the reference raises a bare Exception
            combos.append({"name": name, "keys": sorted(kw), "raises": type(ex).__name__, "message": str(ex)})
    json.dump(combos, open(os.path.join(HERE, "validation.json"), "w"), indent=1)

    # ------------------------------------------------------------------ split_flip.npz
    H, W, Np = 5, 7, 9
    ramp = torch.arange(3 * H * W, dtype=torch.float32).reshape(3, H, W)
    pts = torch.arange(Np * 14, dtype=torch.float32).reshape(Np, 14)
    sf = {"points": pts.numpy(), "ramp": ramp.numpy()}
    for flip_lr in (True, False):
        for flip_ud in (True, False):
            rec.ramp = ramp
            wr = dgr.GaussianRasterizerWrapper(K_ge, (W, H), flip_lr=flip_lr, flip_ud=flip_ud, device=cpu)
            img = wr(pts, np.array([1.0, 2.0, 3.0]), np.array([0.0, 0.0, 0.0, 1.0]))
            key = "lr%d_ud%d" % (flip_lr, flip_ud)
            sf["img_" + key] = img.numpy()
            a = rec.fw_args
            sf["fw_means3D"], sf["fw_colors"], sf["fw_opacity"] = a[1].numpy(), a[2].numpy(), a[3].numpy()
            sf["fw_scales"], sf["fw_rotations"] = a[4].numpy(), a[5].numpy()
            sf["fw_sh_numel"], sf["fw_cov_numel"] = np.array(a[14].numel()), np.array(a[7].numel())
            sf["fw_degree"] = np.array(a[15])
    rec.ramp = None
    np.savez_compressed(os.path.join(HERE, "split_flip.npz"), **sf)

    # ------------------------------------------------------------------ helpers.npz
    B, Np = 2, 6
    g = torch.Generator().manual_seed(0)
    xyz = torch.rand(B, Np, 3, generator=g)
    scales = torch.rand(B, Np, 3, generator=g) + 0.5
    attrs = {"rgb": torch.rand(B, Np, 3, generator=g), "xyz": torch.rand(B, Np, 3, generator=g) * 0.1,
             "scale": torch.rand(B, Np, 3, generator=g) + 0.5}
    hp = {"xyz": xyz.numpy().copy(), "scales": scales.numpy().copy(),
          **{"attr_" + k: v.numpy().copy() for k, v in attrs.items()}}
    out = ref_helpers.get_gaussian_points(xyz, scales, attrs)
    hp["points_full"] = out.numpy()
    hp["xyz_after"], hp["scales_after"] = xyz.numpy().copy(), scales.numpy().copy()  # mutated in place
    xyz2 = torch.rand(B, Np, 3, generator=g)
    scales2 = torch.rand(B, Np, 3, generator=g)
    hp["xyz2"], hp["scales2"] = xyz2.numpy().copy(), scales2.numpy().copy()
    rgb2 = torch.rand(B, Np, 3, generator=g)
    hp["rgb2"] = rgb2.numpy()
    hp["points_default"] = ref_helpers.get_gaussian_points(xyz2, scales2, {"rgb": rgb2}).numpy()

    calls = []

    def fake_rasterizer(pts, pos, quat):
        calls.append((pts.clone(), pos.clone(), quat.clone()))
        base = float(pts.sum())
        return base + torch.arange(3 * 10 * 12, dtype=torch.float32).reshape(3, 10, 12)

    pts = out.detach()
    cam_pos = torch.rand(B, 3, generator=g)
    cam_quat = torch.rand(B, 4, generator=g)
    boxes = [{"x": 2, "y": 1, "w": 5, "h": 4}, {"x": 0, "y": 3, "w": 5, "h": 4}]
    hp["cam_pos"], hp["cam_quat"] = cam_pos.numpy(), cam_quat.numpy()
    hp["boxes"] = np.array([[b["x"], b["y"], b["w"], b["h"]] for b in boxes])
    hp["raster_nocrop"] = ref_helpers.get_gaussian_rasterization(pts, fake_rasterizer, cam_pos, cam_quat).numpy()
    hp["raster_crop"] = ref_helpers.get_gaussian_rasterization(pts, fake_rasterizer, cam_pos, cam_quat, boxes).numpy()
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **hp)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
