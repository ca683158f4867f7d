"""C2 forward + backward, N frames on one stream (a plain workload for rocprofv3 --pmc passes on the backward kernels).
    python tools/fb_loop.py [--piece 128] [--frames 24] [--config C2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
ap = argparse.ArgumentParser()
ap.add_argument("--piece", type=int, default=0)
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--config", default="C2")
args = ap.parse_args()
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfg, sc = synth.make_scene(args.config); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
if args.piece:
    N.set_option("bwd_piece", args.piece)
for i in range(args.frames):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"], cfg["sh_degree"], rs.campos, False, False)
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a, _for_backward=True)
    ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], cfg["sh_degree"], rs.campos, geom, R, binning, img, False)
torch.cuda.synchronize()
