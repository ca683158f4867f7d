"""K1 (the fused cull) alone at C3 through the stage timer, once per cache policy (option stream_policy 0 / 1), for the library
GCR_LIB_PATH names -- the timing side of the knock-out builds recorded in profiles/r06_k1_knockouts.jsonl (tools/ab_variants.sh).
    gpurun -- 'REPS=1 bash tools/ab_variants.sh tag "python tools/k1_knockout.py" ship <variant>'"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfg, sc = synth.make_scene("C3"); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
def f(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"], 3, rs.campos, False, False)
    return ext.rasterize_gaussians(*a, _for_backward=False)
for pol in (0, 1):
    N.set_option("stream_policy", pol)
    for i in range(30): f(i)
    torch.cuda.synchronize()
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(96): f(i); torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    print(json.dumps({"lib": os.path.basename(N.LIB_PATH), "stream_policy": pol, "preprocess_ms_alone": round(st["preprocess"], 4)}), flush=True)
