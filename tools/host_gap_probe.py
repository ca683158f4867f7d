"""Host time between gcr_forward returning (R known) and the next gcr_forward being entered, in the bench loop."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda", 0)
cfg, sc = synth.make_scene("C3")
W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=3) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
e = torch.Tensor([])
L = N.lib()
orig = L.gcr_forward
marks = []
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); t1 = time.perf_counter(); marks.append((t0, t1)); return r
L.gcr_forward = Wrap()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def fwd(i):
    rs = cams[i % 24]
    return ext.rasterize_gaussians(rs.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, e,
                                   rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], 3, rs.campos, False, False)
for i in range(300):
    with torch.cuda.stream(streams[i % 2]):
        fwd(i)
torch.cuda.synchronize()
m = marks[60:]
gaps = [1e6 * (m[i + 1][0] - m[i][1]) for i in range(len(m) - 1)]
inside = [1e6 * (b - a) for a, b in m]
print("host gap between calls: median %.1f us, mean %.1f; inside gcr_forward: median %.1f us; period %.1f us" % (np.median(gaps), np.mean(gaps), np.median(inside), 1e6 * (m[-1][1] - m[0][1]) / (len(m) - 1)))
import cProfile, pstats
L.gcr_forward = orig
pr = cProfile.Profile(); pr.enable()
for i in range(300):
    with torch.cuda.stream(streams[i % 2]):
        fwd(i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(16)
