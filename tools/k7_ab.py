"""C2 forward + backward A/B of the two backward blend kernels (option "bwd_wave_units") over piece sizes, in ONE process
on ONE box: per configuration the stage-timer averages (blend forward of the training-type frame, blend backward incl. the
5 us record-zeroing kernel, preprocess backward), the serial wall time per frame, and the largest gradient difference
against the first configuration (1e-4 * max is the bar the tests hold against the oracle).
    gpurun -- 'python tools/k7_ab.py > gpurun_out/r05_k7_ab.jsonl'
GCR_LIB_PATH selects a variant build (tools/ab_variants.sh)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
def fb(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, H, W, t["shs"], 3, rs.campos, False, False)
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a, _for_backward=True)
    return ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"], 3, rs.campos, geom, R, binning, img, False)
NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
ref = None
configs = [(1, 128), (0, 128), (0, 96), (0, 160), (0, 192), (0, 223), (1, 223), (1, 128), (0, 128)]
if len(sys.argv) > 2:
    configs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
for wave_units, piece in configs:
    N.set_option("bwd_wave_units", wave_units); N.set_option("bwd_piece", piece)
    g = [x.clone() for x in fb(3)]
    torch.cuda.synchronize()
    if ref is None: ref = g
    worst = max(float((a - b).abs().max() / max(1.0, float(b.abs().max()))) for a, b in zip(g, ref))
    for i in range(8): fb(i)
    torch.cuda.synchronize()
    N.set_option("timing", 1); N.stage_ms(); torch.cuda.synchronize()
    for i in range(96): fb(i)
    torch.cuda.synchronize()
    st = N.stage_ms(); N.set_option("timing", 0)
    walls = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(96): fb(i)
        torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) / 96 * 1e3)
    print(json.dumps({"config": cfgname, "bwd_wave_units": wave_units, "bwd_piece": piece, "lib": os.path.basename(N.LIB_PATH),
                      "blend_fwd_ms": round(st["blend_fwd"], 4), "blend_bwd_ms": round(st["blend_bwd"], 4),
                      "preprocess_bwd_ms": round(st["preprocess_bwd"], 4), "fwd_bwd_wall_ms": round(float(np.median(walls)), 4),
                      "wall_min_ms": round(min(walls), 4), "worst_rel_diff_vs_first": worst}), flush=True)
N.set_option("bwd_wave_units", 0); N.set_option("bwd_piece", 160)
