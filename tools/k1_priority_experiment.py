"""Does a HIGH-PRIORITY stream get K1's workgroups onto the CUs ahead of the previous frame's blend?  (The int-returning
line is a host chain through K1's end -- tools/host_profile_int_api.py -- and K1 is stretched from 78 to ~150 us by the
blend beside it.)  Frames through the staged C ABI -- gcr_forward_preprocess (K1 + tile count + column scan, one host wait)
on stream A, gcr_forward_render (scatter, sort, blend) on stream B after an event -- with A = B's priority, then A high:
    gpurun -- 'python tools/k1_priority_experiment.py > gpurun_out/r05_k1_priority_experiment.jsonl'"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
dev = torch.device("cuda:0"); E = torch.Tensor([])
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]; P = cfg["P"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
L = N.lib()
lo_pri, hi_pri = 0, -1
NS = 3
render_streams = [torch.cuda.Stream(device=dev, priority=lo_pri) for _ in range(NS)]
k1_streams = {0: [torch.cuda.Stream(device=dev, priority=lo_pri) for _ in range(NS)],
              1: [torch.cuda.Stream(device=dev, priority=hi_pri) for _ in range(NS)]}
gbytes, ibytes = L.gcr_geometry_bytes(P), L.gcr_image_bytes(W, H)
keep = []
ref = {}
def frame(i, hi, same_stream=False):
    rs = cams[i % 24]
    sR = render_streams[i % NS]
    sK = sR if same_stream else k1_streams[hi][i % NS]
    byte = dict(dtype=torch.uint8, device=dev)
    with torch.cuda.stream(sR):
        cam, kc = ext._camera(dev, rs.bg, rs.view_matrix, rs.proj_matrix, rs.campos, rs.tanfovx, rs.tanfovy, H, W, 1.0,
                              cfg["sh_degree"], False, False, False)
        g, kg = ext._gaussians(dev, P, t["means3D"], t["opacities"], t["shs"], None, t["scales"], t["rotations"], None)
        geom = torch.empty((gbytes,), **byte); img = torch.empty((ibytes,), **byte)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        out = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    info = N.FrameInfo()
    N.check(L.gcr_forward_preprocess(C.byref(cam), C.byref(g), geom.data_ptr(), gbytes, img.data_ptr(), ibytes,
                                     radii.data_ptr(), C.byref(info), C.c_void_p(sK.cuda_stream)), "preprocess")
    R = int(info.num_rendered)
    with torch.cuda.stream(sR):
        binning = torch.empty((L.gcr_binning_bytes_lean(R, W, H),), **byte)
    if sK is not sR:
        sR.wait_stream(sK)
    N.check(L.gcr_forward_render(C.byref(cam), C.byref(g), geom.data_ptr(), gbytes, binning.data_ptr(), binning.numel(),
                                 img.data_ptr(), ibytes, C.byref(info), out.data_ptr(), C.c_void_p(sR.cuda_stream)), "render")
    keep.append((geom, img, radii, out, binning, kc, kg))
    if len(keep) > 12: keep.pop(0)
    return R, out
def run(n, **kw):
    for i in range(n): frame(i, **kw)
    torch.cuda.synchronize()
for name, kw in (("one stream per frame", dict(hi=0, same_stream=True)), ("K1 on a second stream, same priority", dict(hi=0)),
                 ("K1 on a second stream, HIGH priority", dict(hi=1))) * 3:
    R, out = frame(5, **kw); torch.cuda.synchronize()
    if "img" not in ref: ref["img"] = out.clone(); ref["R"] = R
    same = R == ref["R"] and torch.equal(out, ref["img"])
    run(48, **kw)
    t0 = time.perf_counter(); run(600, **kw); wall = time.perf_counter() - t0
    print(json.dumps({"config": cfgname, "variant": name, "frames_per_s": round(600 / wall, 1), "period_us": round(wall / 600 * 1e6, 1),
                      "same_bits": bool(same), "priority_range": [lo_pri, hi_pri]}), flush=True)
