#!/usr/bin/env python
"""Threshold-flip census: how many pixels of a frame change a DISCRETE decision --
`n_contrib` (alpha < 1/255 / T < 1e-4 / power > 0) -- or move by more than 1e-4 when every exp() result is off by
one ulp?  This is the difference between the bit-reproducible gcr_expf shared by oracle and HIP kernels and "some
1-ulp exp" (CUDA's expf, v_exp_f32).  Runs on the CPU oracle only.

    python tools/exp_census.py [C2|C3] [n_points]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def census(cfg_name="C2", points=None, pose=3):
    import scenes
    from gaussiancity_amd import synth
    from oracle import oracle as O
    cfg, sc = synth.make_scene(cfg_name, points)
    W, H = cfg["W"], cfg["H"]
    rs = scenes.camera(W, H, pose_index=pose, radius=512.0, altitude=640.0)._replace(sh_degree=cfg["sh_degree"])
    kw = dict(scenes.settings_kwargs(rs), means3D=sc["means3D"], opacities=sc["opacities"], shs=sc["shs"],
              scales=sc["scales"], rotations=sc["rotations"])
    frames = {}
    try:
        for bias in (0, -1, 1):
            O.set_exp_bias(bias)
            fr = O.Frame(**kw)
            frames[bias] = (fr.n_contrib.copy(), fr.out_color.copy(), fr.R)
    finally:
        O.set_exp_bias(0)
    nc0, img0, R = frames[0]
    out = {"config": cfg_name, "gaussians": int(cfg["P"]), "pixels": int(W * H), "num_rendered": int(R)}
    for bias in (-1, 1):
        nc, img, _ = frames[bias]
        d = np.abs(img - img0).max(axis=0).reshape(-1)
        out["exp%+d_ulp" % bias] = {"pixels_with_other_n_contrib": int((nc != nc0).sum()),
                                    "pixels_moved_more_than_1e-4": int((d > 1e-4).sum()),
                                    "max_abs_image_change": float(d.max())}
    return out


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    pts = int(sys.argv[2]) if len(sys.argv) > 2 else None
    print(json.dumps(census(name, pts)))
