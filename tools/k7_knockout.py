"""K7 knock-out timings (experiment build: results are WRONG with any flag set): which part of the backward blend the
launch time hangs on.  flags: 1 = no global flush, 4 = no LDS adds, 8 = one LDS read per step instead of three,
16 = no step arithmetic (loads and loops stay), 32 = only the units' prologues (pixel state, early exits)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GCR_LIB_PATH"] = os.path.join(ROOT, "tools", "_build", "libgcr_hip_exp.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfg, sc = synth.make_scene("C2"); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
N.set_option("bwd_piece", int(sys.argv[1]) if len(sys.argv) > 1 else 128)
def fb(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"], 3, rs.campos, False, False)
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a, _for_backward=True)
    ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], 3, rs.campos, geom, R, binning, img, False)
for flag in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, 4, 5, 16, 21, 32, 0)):
    N.set_option("k7_skip_flush", flag)
    for i in range(5): fb(i)
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(48): fb(i)
    torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    print(json.dumps({"flags": flag, "blend_bwd_ms": round(st["blend_bwd"], 4), "preprocess_bwd_ms": round(st["preprocess_bwd"], 4)}), flush=True)
N.set_option("k7_skip_flush", 0)
