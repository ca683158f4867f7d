"""Backward pieces (option `bwd_piece`): C2 forward+backward stage times and gradient parity per piece size, and what
the piece size costs the forward blend at C3.  One JSON line per setting.

    python tools/piece_probe.py [--pieces 256,192,128,96,64] [--no-c3] [--no-parity]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper

ap = argparse.ArgumentParser()
ap.add_argument("--pieces", default="256,192,128,96,64")
ap.add_argument("--no-c3", action="store_true")
ap.add_argument("--no-parity", action="store_true")
args = ap.parse_args()
pieces = [int(x) for x in args.pieces.split(",")]
dev = torch.device("cuda:0")
E = torch.Tensor([])
NAMES = ("dL_dmean2D", "dL_dcolor", "dL_dopacity", "dL_dmean3D", "dL_dcov3D", "dL_dsh", "dL_dscale", "dL_drot")


def scene(name):
    cfg, sc = synth.make_scene(name)
    W, H = cfg["W"], cfg["H"]
    wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
    cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
    t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
    return cfg, sc, cams, t


cfg, sc, cams, t = scene("C2")
W, H = cfg["W"], cfg["H"]
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)


def fwd(i, tt=t, cc=cams, for_bwd=False):
    rs = cc[i % 24]
    a = (rs.bg, tt["means3D"], E, tt["opacities"], tt["scales"], tt["rotations"], rs.scale_modifier, E, rs.view_matrix,
         rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, tt["shs"], 3, rs.campos, False, False)
    return rs, ext.rasterize_gaussians(*a, _for_backward=for_bwd)


def fb(i):
    rs, (R, color, radii, geom, binning, img) = fwd(i, for_bwd=True)
    g = ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], rs.scale_modifier, E,
                                         rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], 3,
                                         rs.campos, geom, R, binning, img, False)
    return color, g


ref = None
if not args.no_parity:
    from oracle import oracle as O
    rs = cams[3]
    fr = O.Frame(img_h=H, img_w=W, tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, bg=rs.bg.cpu().numpy(), scale_modifier=1.0,
                 view_matrix=rs.view_matrix.cpu().numpy(), proj_matrix=rs.proj_matrix.cpu().numpy(), sh_degree=3,
                 campos=rs.campos.cpu().numpy(), means3D=sc["means3D"], opacities=sc["opacities"], scales=sc["scales"],
                 rotations=sc["rotations"], shs=sc["shs"])
    ref = (fr.out_color.copy(), fr.backward(dpix.cpu().numpy()))

for P in pieces:
    N.set_option("bwd_piece", P)
    for i in range(6):
        fb(i)
    line = {"config": "C2", "bwd_piece": P}
    if ref is not None:
        color, g = fb(3)
        torch.cuda.synchronize()
        line["image_bit_exact"] = bool(np.array_equal(color.cpu().numpy().view(np.uint32), ref[0].view(np.uint32)))
        worst = {}
        for nme, tg in zip(NAMES, g):
            r = ref[1][nme]
            worst[nme] = float("%.3g" % (float(np.abs(r - tg.cpu().numpy().reshape(r.shape)).max()) / max(1.0, float(np.abs(r).max()))))
        line["grad_err_over_max"] = worst
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(48):
        fb(i)
    torch.cuda.synchronize()
    line["fwd_bwd_ms"] = round(1e3 * (time.perf_counter() - t0) / 48, 4)
    N.set_option("timing", 1)
    N.stage_ms()
    for i in range(48):
        fb(i)
    torch.cuda.synchronize()
    st = N.stage_ms()
    N.set_option("timing", 0)
    line.update({k + "_ms": round(st[k], 4) for k in ("blend_fwd", "blend_bwd", "preprocess_bwd")})
    print(json.dumps(line), flush=True)

if not args.no_c3:
    del t
    torch.cuda.empty_cache()
    cfg3, sc3, cams3, t3 = scene("C3")
    for P in pieces:
        N.set_option("bwd_piece", P)
        for i in range(30):
            fwd(i, t3, cams3, True)
        N.set_option("timing", 1)
        N.stage_ms()
        for i in range(96):
            fwd(i, t3, cams3, True)
        torch.cuda.synchronize()
        st = N.stage_ms()
        N.set_option("timing", 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(96):
            fwd(i, t3, cams3, True)
        torch.cuda.synchronize()
        print(json.dumps({"config": "C3 forward announced as a training frame, one stream", "bwd_piece": P, "blend_fwd_ms": round(st["blend_fwd"], 4),
                          "frame_ms": round(1e3 * (time.perf_counter() - t0) / 96, 4)}), flush=True)
N.set_option("bwd_piece", 160)
