"""Where the host's time goes between two frames of the native module's int-returning rasterize_gaussians() (the entry point
that is like for like with the reference's binding: one host wait per frame).  The next frame's K1 cannot be enqueued before
this frame's num_rendered has come back AND the Python in between has run, so that turnaround is part of the frame period
(profiles/r05_timeline_c3_native_int_api.txt).  Prints: the turnaround (return of gcr_forward -> next call of gcr_forward),
the time inside gcr_forward (enqueue + wait), frames/s, and a cProfile of the loop.
    gpurun -- 'python tools/host_profile_int_api.py [C3] > gpurun_out/r05_host_profile_int_api.txt'"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def f(i):
    rs = cams[i % 24]
    a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], 1.0, E, rs.view_matrix, rs.proj_matrix,
         rs.tanfovx, rs.tanfovy, H, W, t["shs"], cfg["sh_degree"], rs.campos, False, False)
    with torch.cuda.stream(streams[i % 3]):
        return ext.rasterize_gaussians(*a, _for_backward=False)
def loop(n):
    for i in range(n): f(i)
    torch.cuda.synchronize()
loop(48)
L = N.lib()
real = L.gcr_forward
marks = []
def timed(*a):
    t0 = time.perf_counter(); r = real(*a); marks.append((t0, time.perf_counter())); return r
L.gcr_forward = timed
n = 600
t0 = time.perf_counter(); loop(n); wall = time.perf_counter() - t0
L.gcr_forward = real
inside = np.array([b - a for a, b in marks]) * 1e6
turn = np.array([marks[i + 1][0] - marks[i][1] for i in range(len(marks) - 1)]) * 1e6
print("frames/s %.0f   period %.1f us" % (n / wall, wall / n * 1e6))
print("inside gcr_forward (enqueue + wait for num_rendered): median %.1f us  p10 %.1f  p90 %.1f" % (np.median(inside), np.percentile(inside, 10), np.percentile(inside, 90)))
print("turnaround (return -> next call), pure Python:         median %.1f us  p10 %.1f  p90 %.1f" % (np.median(turn), np.percentile(turn, 10), np.percentile(turn, 90)))
pr = cProfile.Profile(); pr.enable(); loop(n); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
