"""Where the HOST time of the C4 rasterizer leg goes (the leg is host-bound: ~0.05 ms of GPU work per frame).
cProfile over N iterations of helpers -> wrapper -> autograd + crop + L1 + backward; top entries by cumulative time.
    python tools/host_profile_c4.py [--host-camera] [--iters 400]"""
import argparse, cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiancity_amd import helpers, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper
ap = argparse.ArgumentParser()
ap.add_argument("--host-camera", action="store_true", help="closed-form host camera")
ap.add_argument("--device-camera", action="store_true", help="the reference recipe on the device")
ap.add_argument("--iters", type=int, default=400)
ap.add_argument("--top", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg, sc = synth.make_scene("C4")
W, H = cfg["W"], cfg["H"]; cw, ch = cfg["crop"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev, host_camera=True if args.host_camera else (False if args.device_camera else None))
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
for i in range(50): leg(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(args.iters): leg(i)
torch.cuda.synchronize(); print("ms per frame: %.4f" % (1e3 * (time.perf_counter() - t0) / args.iters))
# forward / backward split on the host
tf = tb = 0.0
for i in range(args.iters):
    pos, quat = poses[i % 24]; leaf.grad = None
    a = time.perf_counter()
    img = helpers.get_gaussian_rasterization(leaf[None], wr, [pos], [quat], crop_bboxes=box)[0]
    loss = (img - target).abs().mean()
    b = time.perf_counter()
    loss.backward()
    c = time.perf_counter()
    tf += b - a; tb += c - b
print("host ms: forward+loss %.4f  backward %.4f" % (1e3 * tf / args.iters, 1e3 * tb / args.iters))
pr = cProfile.Profile(); pr.enable()
for i in range(args.iters): leg(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(args.top)
print("\n".join(l[:150] for l in s.getvalue().splitlines()))
