"""C4 rasterizer leg (helpers -> wrapper -> autograd + crop + L1 + backward) and the inference loop's frame, with and
without the host wait for num_rendered, ALTERNATING in one process (box-to-box and run-order effects cancel):
ext._SYNC_ONLY toggles between blocks.  One JSON line.
    python tools/c4_leg_ab.py [--iters 300] [--blocks 6] [--camera closed-form|reference]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiancity_amd import _native as N, ext, helpers, synth
from gaussiancity_amd.frames import InferenceLoop
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--blocks", type=int, default=6)
ap.add_argument("--camera", default="closed-form")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg, sc = synth.make_scene("C4")
W, H = cfg["W"], cfg["H"]; cw, ch = cfg["crop"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev, host_camera=True if args.camera == "closed-form" else "reference")
rot = sc["rotations"][:, [1, 2, 3, 0]]
pts = np.concatenate([sc["means3D"], sc["opacities"], sc["scales"], rot, sc["colors_precomp"]], axis=1).astype(np.float32)
leaf = torch.from_numpy(pts).to(dev).requires_grad_(True)
target = torch.zeros((3, ch, cw), device=dev)
box = [{"x": (W - cw) // 2, "y": (H - ch) // 2, "w": cw, "h": ch}]
poses = synth.orbit_poses()
def leg(i):
    pos, quat = poses[i % 24]
    leaf.grad = None
    img = helpers.get_gaussian_rasterization(leaf[None], wr, [pos], [quat], crop_bboxes=box)[0]
    (img - target).abs().mean().backward()
n_inf = 518400
cfg2, sc2 = synth.make_scene("C4", n_inf)
rot2 = np.zeros((n_inf, 4), np.float32); rot2[:, 0] = 1.0
pinf = torch.from_numpy(np.concatenate([sc2["means3D"], np.ones((n_inf, 1), np.float32), sc2["scales"], rot2, sc2["colors_precomp"]], axis=1).astype(np.float32)).to(dev)
loop = InferenceLoop(lambda p, cp, cq: wr(p, cp, cq), device=dev, n_streams=3)
def inf(n):
    with torch.no_grad():
        loop.run(pinf, [poses[i % 24] for i in range(n)], consume=lambda i, f: None)
res = {"leg_ms": {"async": [], "sync": []}, "inference_fps": {"async": [], "sync": []}}
for mode in (False, True):
    ext._SYNC_ONLY = mode
    for i in range(60): leg(i)
    inf(48)
rescued0 = N.lib().gcr_rescue_count()
for b in range(args.blocks):
    for mode in (False, True):
        ext._SYNC_ONLY = mode
        for i in range(20): leg(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(args.iters): leg(i)
        torch.cuda.synchronize()
        res["leg_ms"]["sync" if mode else "async"].append(round(1e3 * (time.perf_counter() - t0) / args.iters, 4))
        inf(24)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        inf(args.iters)
        torch.cuda.synchronize()
        res["inference_fps"]["sync" if mode else "async"].append(round(args.iters / (time.perf_counter() - t0), 1))
res["median"] = {k: {m: float(np.median(v)) for m, v in d.items()} for k, d in res.items() if k != "median"}
res["rescued_frames"] = N.lib().gcr_rescue_count() - rescued0
res["camera"] = args.camera
print(json.dumps(res))
