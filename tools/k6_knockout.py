"""Forward blend (K6) knock-out timings at C3 (experiment build; images are WRONG with a flag set):
1 = no step arithmetic (staging, masks, lists, LDS reads and loops stay), 2 = no chunk loop at all (launch + output stores)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GCR_LIB_PATH"] = os.path.join(ROOT, "tools", "_build", "libgcr_hip_exp.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
def fwd(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"], 3, rs.campos, False, False)
    return ext.rasterize_gaussians(*a)
for i in range(30): fwd(i)
for flag in (0, 1, 2, 0):
    N.set_option("k6_debug", flag)
    for i in range(8): fwd(i)
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(96): fwd(i)
    torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    print(json.dumps({"config": cfgname, "flags": flag, "blend_fwd_ms": round(st["blend_fwd"], 4), "sort_ms": round(st["sort"], 4)}), flush=True)
N.set_option("k6_debug", 0)
