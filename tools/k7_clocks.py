"""Per-wave phase clocks of the backward blend (K7) at C2: where a wave's life goes, how full the SIMDs are over the
launch.  Needs the experiment build:  make -C gaussiancity_amd/csrc experiments  (tools/_build/libgcr_hip_exp.so).

    python tools/k7_clocks.py [--piece 128] [--config C2]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GCR_LIB_PATH"] = os.path.join(ROOT, "tools", "_build", "libgcr_hip_exp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch

from gaussiancity_amd import _native as N, ext, synth
from gaussiancity_amd.rasterizer import GaussianRasterizerWrapper

ap = argparse.ArgumentParser()
ap.add_argument("--piece", type=int, default=256)
ap.add_argument("--config", default="C2")
ap.add_argument("--pose", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
E = torch.Tensor([])
cfg, sc = synth.make_scene(args.config)
W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
dpix = torch.from_numpy(synth.grad_image(W, H, cfg["seed"])).to(dev)
L = N.lib()
L.gcr_debug_set_clock_buffer.argtypes = [C.c_void_p]
L.gcr_debug_set_clock_buffer.restype = None
N.set_option("bwd_piece", args.piece)

rs = cams[args.pose]
a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix,
     rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], cfg["sh_degree"], rs.campos, False, False)
for _ in range(3):
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a)


def bwd():
    return ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], rs.scale_modifier, E,
                                            rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"],
                                            cfg["sh_degree"], rs.campos, geom, R, binning, img, False)


for _ in range(5):
    bwd()
NITEMS, NW = 4, 10
grid = 4096  # >= the persistent grid
buf = torch.zeros((grid, 4, NITEMS, NW), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
N.set_option("timing", 1)
N.stage_ms()
L.gcr_debug_set_clock_buffer(buf.data_ptr())
bwd()
torch.cuda.synchronize()
L.gcr_debug_set_clock_buffer(None)
st = N.stage_ms()
N.set_option("timing", 0)
b = buf.cpu().numpy().astype(np.uint64)          # [wg, wave, item, 10]
np.save(os.path.join(ROOT, "gpurun_out", "k7_clocks_p%d.npy" % args.piece), b)
used = b[..., 2] != 0
item_idx = np.broadcast_to(np.arange(NITEMS)[None, None, :], used.shape)[used]
waves = b[used]
hw, meta, clk = waves[:, 0], waves[:, 1], waves[:, 2:].astype(np.int64)
n_ent = (meta & np.uint64(0xffffffff)).astype(np.int64)
steps = (meta >> np.uint64(32)).astype(np.int64)
# s_memtime is a per-CU counter here (the bases differ between CUs): put every CU's first clock at 0
lo32 = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = ((hw >> np.uint64(32)).astype(np.int64) & 15) * (1 << 16) + (lo32 & 0xff00)
for x in np.unique(cu):
    clk[cu == x] -= clk[cu == x][:, 0].min()
span = clk[:, 7].max()
names = ("prefetch_issue", "state+stage_own", "stage_skew", "compaction", "walk", "walk_skew", "flush+fill")
ph = {nm: clk[:, k + 1] - clk[:, k] for k, nm in enumerate(names)}
life = clk[:, 7] - clk[:, 0]
edges = np.linspace(0, span, 21)
walking = [round(float(np.clip(np.minimum(clk[:, 5], hi) - np.maximum(clk[:, 4], lo), 0, None).sum()) / (hi - lo) / 1024, 2)
           for lo, hi in zip(edges[:-1], edges[1:])]
out = {
    "config": args.config, "bwd_piece": args.piece, "R": int(R), "blend_bwd_stage_ms": round(st["blend_bwd"], 4),
    "workgroups_with_work": int(used[:, 0, 0].sum()), "items_clocked": int(len(waves) // 4),
    "cu_span_ticks_max": int(span), "ticks_per_us_if_span_is_stage": round(span / (st["blend_bwd"] * 1e3), 1),
    "walking_waves_per_simd_by_twentieth (first %d items of every workgroup)" % NITEMS: walking,
    "phase_share": {k: round(float(v.sum()) / float(life.sum()), 3) for k, v in ph.items()},
    "phase_mean_ticks_first_item": {k: int(v[item_idx == 0].mean()) for k, v in ph.items()},
    "phase_mean_ticks_later_items": {k: int(v[item_idx > 0].mean()) if (item_idx > 0).any() else 0 for k, v in ph.items()},
    "entries_per_item_mean": round(float(n_ent.mean()), 1), "row_steps_per_wave_mean": round(float(steps.mean()), 1),
    "walk_ticks_per_step": round(float(ph["walk"].sum()) / max(1, int(steps.sum())), 1),
}
print(json.dumps(out))
