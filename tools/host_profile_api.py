"""Host time of ONE frame through the Python API (GaussianRasterizer.forward -> gcr_forward_async), the headline's entry
point: how long the calling thread needs to enqueue a frame when the GPU is not the limit (the first frames of a block
after a synchronize -- bench.py --steps 20 pays this at the start of every block), where it goes (cProfile), and
how much of it is the C call.   python tools/host_profile_api.py [C3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiancity_amd import _native as N, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper, GaussianRasterizer
dev = torch.device("cuda", 0)
cfg, sc = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "C3")
W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
rasters = [GaussianRasterizer(rs) for rs in cams]
means2D = torch.zeros_like(t["means3D"])
L = N.lib()
orig = L.gcr_forward_async
marks = []
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); marks.append(time.perf_counter() - t0); return r
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def api(i):
    with torch.no_grad():
        return rasters[i % 24](means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], scales=t["scales"],
                               rotations=t["rotations"], shs=t["shs"])
for i in range(120):  # hints, clocks
    with torch.cuda.stream(streams[i % 3]):
        api(i)
torch.cuda.synchronize()
L.gcr_forward_async = Wrap()
per, inside = [], []
for rep in range(40):  # bursts of three frames into an idle GPU: pure host time
    torch.cuda.synchronize()
    del marks[:]
    ts = [time.perf_counter()]
    for i in range(3):
        with torch.cuda.stream(streams[i % 3]):
            api(rep * 3 + i)
        ts.append(time.perf_counter())
    per.append([1e6 * (b - a) for a, b in zip(ts[:-1], ts[1:])])
    inside.append([1e6 * m for m in marks])
per, inside = np.array(per), np.array(inside)
print("host us per frame, burst of 3 after a synchronize (median over 40): frames 0/1/2 = %s; inside gcr_forward_async = %s"
      % (np.round(np.median(per, 0), 1), np.round(np.median(inside, 0), 1)))
L.gcr_forward_async = orig
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for rep in range(100):
    torch.cuda.synchronize()
    for i in range(3):
        with torch.cuda.stream(streams[i % 3]):
            api(rep * 3 + i)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

# ---- where the Python share goes: the same frame entered at four depths, bursts of three into an idle GPU --------------
from gaussiancity_amd import ext
from gaussiancity_amd.rasterizer import RasterizeGaussiansFunction, _absent
def burst(fn, reps=60, with_stream=True):
    out = []
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3):
            if with_stream:
                with torch.cuda.stream(streams[i % 3]):
                    fn(rep * 3 + i)
            else:
                fn(rep * 3 + i)
        out.append(1e6 * (time.perf_counter() - t0) / 3)
    return float(np.median(out))
e = torch.Tensor([])
def native(i):
    rs = cams[i % 24]
    return ext.rasterize_gaussians_ticket(rs.bg, t["means3D"], e, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, e,
                                          rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], 3,
                                          rs.campos, False, False, _for_backward=False)
def function(i):
    with torch.no_grad():
        return RasterizeGaussiansFunction.apply(t["means3D"], means2D, t["shs"], _absent(), t["opacities"], t["scales"], t["rotations"],
                                                _absent(), cams[i % 24])
def nothing(i):
    pass
def set_stream_only(i):
    torch.cuda.set_stream(streams[i % 3])
print("us per frame: stream context alone %.1f | set_stream alone %.1f | native ticket call %.1f (no stream switch %.1f) | "
      "Function.apply %.1f | Module call %.1f"
      % (burst(nothing), burst(set_stream_only, with_stream=False), burst(native), burst(native, with_stream=False),
         burst(function), burst(api)))
