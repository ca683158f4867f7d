"""K7 with / without its global flush (timing experiment).  Needs an experiment build of the library:
    make -C gaussiancity_amd/csrc clean all EXTRA=-DGCR_EXPERIMENTS
(the shipping library does not know the option `k7_skip_flush`)."""
import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0")
cfg, sc = synth.make_scene("C2")
W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
E = torch.Tensor([])
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
def fb(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], 3, rs.campos, False, False)
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a)
    ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], 3, rs.campos, geom, R, binning, img, False)
for flag in (0, 1, 0, 1):
    N.set_option("k7_skip_flush", flag)
    for i in range(5): fb(i)
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(24): fb(i)
    torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    print(json.dumps({"k7_skip_flush": flag, "blend_bwd_ms": round(st["blend_bwd"], 4), "blend_fwd_ms": round(st["blend_fwd"], 4), "preprocess_bwd_ms": round(st["preprocess_bwd"],4)}))
N.set_option("k7_skip_flush", 0)
# wall-clock check of the event-based stage times: the backward alone, repeated on one frame's state
# (every output is rewritten by each call, so repeating it is valid)
rs = cams[3]
a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], 3, rs.campos, False, False)
R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a)
def bwd():
    return ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], 3, rs.campos, geom, R, binning, img, False)
for i in range(10): bwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): bwd()
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 100
def fwd():
    return ext.rasterize_gaussians(*a)
for i in range(10): fwd()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(100): fwd()
torch.cuda.synchronize(); wallf = (time.perf_counter() - t0) / 100
print(json.dumps({"backward_wall_ms": round(wall * 1e3, 4), "forward_wall_ms": round(wallf * 1e3, 4)}))
