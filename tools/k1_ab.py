"""K1 (preprocess) A/B on ONE box in ONE process: the stateless cull against the static scene's cull cache
(gaussiancity_amd/cull_cache.py), forward frames of a BASELINE config one after the other on one stream -- the stage timer's
average for K1 alone, the serial wall time per frame, and whether image / radii / num_rendered of the two are the same bits.
    gpurun -- 'python tools/k1_ab.py C3 > gpurun_out/r05_k1_ab.jsonl'
GCR_LIB_PATH selects a variant build (tools/ab_variants.sh; e.g. EXTRA=-DGCR_K1_STAGES_CACHED=3)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth, cull_cache
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 96
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
use_sh = not cfg.get("precomp_color", False)
def f(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E if use_sh else t["colors_precomp"], t["opacities"], t["scales"], t["rotations"], 1.0, E,
         rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"] if use_sh else E, cfg["sh_degree"], rs.campos,
         False, False)
    return ext.rasterize_gaussians(*a, _for_backward=False)
ref = {}
# (The experiment switches this tool drove for profiles/r05_k1_ab_same_process.jsonl and r05_k1_ab_stateless_clamp_touch.jsonl
# -- the grouped loop for the stateless stream, the prefetch clamp, the early SH load, each per launch through the
# environment of the experiment build -- are in the tree at commit 3257f5d; what they decided is in the kernels' comments.)
for mode in (0, 1, 0, 1, 0, 1):
    cull_cache.enable(bool(mode))
    same = True
    for pose in (0, 5, 17):
        R, color, radii = f(pose)[:3]
        torch.cuda.synchronize()
        if pose not in ref:
            ref[pose] = (int(R), color.clone(), radii.clone())
        else:
            r0 = ref[pose]
            same = same and int(R) == r0[0] and torch.equal(color, r0[1]) and torch.equal(radii, r0[2])
    for i in range(8): f(i)
    torch.cuda.synchronize()
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(frames): f(i)
    torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    walls = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(frames): f(i)
        torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) / frames * 1e3)
    print(json.dumps({"config": cfgname, "cull_cache": mode, "lib": os.path.basename(N.LIB_PATH),
                      "preprocess_ms": round(st["preprocess"], 4), "serial_wall_ms": round(float(np.median(walls)), 4),
                      "same_bits_as_first": bool(same)}), flush=True)
cull_cache.enable(False)
