"""Quick stage timing of the point-generation / visibility path at full size (gpurun helper)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiancity_amd import points as P, synth, _native_v as V
dev = torch.device("cuda", 0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
L = synth.s_layout(size, 2001)
inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
t = [torch.from_numpy(L[k]).to(dev) for k in ("INS", "TD_HF", "BU_HF", "PTS")]
V.lib(); V.set_option("timing", 1)
P.OCCUPANCY_MIN_VOXELS = 1 << 62
for it in range(3):
    pts = P.extrude_points(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, *t)
    loc = pts[:, :3].contiguous(); loc[:, 2] += 1
    sc = pts[:, 3:4].repeat(1, 3).contiguous()
    ids = torch.arange(1, len(pts) + 1, dtype=torch.int32, device=dev)[:, None]
    h = w = size; d = int(loc[:, 2].max()) + 2
    vol, occ = P.points_to_volume(loc, ids, sc, h, w, d, return_occupancy=True)
    rig, cam_pos, cam_quat = synth.layout_camera(size)
    look = P.get_camera_look_at(cam_pos, cam_quat)
    ori = torch.tensor([cam_pos[1], cam_pos[0], cam_pos[2] + 1], dtype=torch.float32)
    view = torch.tensor([look[1] - cam_pos[1], look[0] - cam_pos[0], look[2] - cam_pos[2]], dtype=torch.float32)
    up = torch.tensor([0, 0, 1], dtype=torch.float32)
    K = rig["intrinsics"]
    ref = None
    for name, kw in (("plain", {}), ("jump", dict(occupancy=occ))):
        torch.cuda.synchronize(); t4 = time.perf_counter()
        out = P.ray_voxel_intersection_perspective(vol, ori, view, up, K[0], [K[5], K[2]], [540, 960], 1, **kw)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        ref = out if ref is None else ref
        print("  traversal %s: %.3f ms, hit %.2f same %s" % (name, 1e3 * (t5 - t4), float((out[0] != 0).float().mean()),
              bool(torch.equal(out[0], ref[0]) and torch.equal(out[1].view(torch.int32), ref[1].view(torch.int32)))))
    print("iter", it, "N", len(pts), "vol", (h, w, d), V.stage_ms())
