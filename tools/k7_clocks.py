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
    R, color, radii, geom, binning, img = ext.rasterize_gaussians(*a, _for_backward=True)


def bwd():
    return ext.rasterize_gaussians_backward(rs.bg, t["means3D"], radii, E, t["scales"], t["rotations"], rs.scale_modifier, E,
                                            rs.view_matrix, rs.proj_matrix, rs.tanfovx, rs.tanfovy, dpix, t["shs"],
                                            cfg["sh_degree"], rs.campos, geom, R, binning, img, False)


for _ in range(5):
    bwd()
NW = 10
grid = 256 + 4 * (R // 64 + ((W + 15) // 16) * ((H + 15) // 16)) + 64  # >= the launch's grid (one wave per workgroup)
buf = torch.zeros((grid, NW), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
N.set_option("timing", 1)
N.stage_ms()
L.gcr_debug_set_clock_buffer(buf.data_ptr())
bwd()
torch.cuda.synchronize()
L.gcr_debug_set_clock_buffer(None)
st = N.stage_ms()
N.set_option("timing", 0)
b = buf.cpu().numpy().astype(np.uint64)          # [wave, 10]: hw id | xcc << 32, entries, clk[0..7]
np.save(os.path.join(ROOT, "gpurun_out", "k7_clocks_p%d.npy" % args.piece), b)
waves = b[b[:, 2] != 0]
hw, n_ent, clk = waves[:, 0], waves[:, 1].astype(np.int64), waves[:, 2:].astype(np.int64)
# s_memtime is a per-CU counter here (the bases differ between CUs): put every CU's first clock at 0
lo32 = (hw & np.uint64(0xffffffff)).astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
cu = xcc * (1 << 16) + (lo32 & 0xff00)
simd = cu * 4 + ((lo32 >> 4) & 3)
for x in np.unique(cu):
    clk[cu == x] -= clk[cu == x][:, 0].min()
span = clk[:, 4].max()
life = clk[:, 4] - clk[:, 0]
nsimd = len(np.unique(simd))
edges = np.linspace(0, span, 21)
resident = [round(float(np.clip(np.minimum(clk[:, 4], hi) - np.maximum(clk[:, 0], lo), 0, None).sum()) / (hi - lo) / nsimd, 2)
            for lo, hi in zip(edges[:-1], edges[1:])]
out = {
    "config": args.config, "bwd_piece": args.piece, "R": int(R), "blend_bwd_stage_ms": round(st["blend_bwd"], 4),
    "waves_with_work": int(len(waves)), "simds_seen": int(nsimd), "cu_span_ticks_max": int(span),
    "ticks_per_us_if_span_is_stage": round(span / (st["blend_bwd"] * 1e3), 1),
    "wave_life_ticks_mean_p50_p90_max": [int(life.mean())] + [int(np.percentile(life, q)) for q in (50, 90, 100)],
    "resident_working_waves_per_simd_by_twentieth": resident,
    "wave_start_ticks_p10_p50_p90_p99": [int(np.percentile(clk[:, 0], q)) for q in (10, 50, 90, 99)],
    "phase_mean_ticks": {"prologue (work item, pixel state)": int((clk[:, 1] - clk[:, 0]).mean()),
                         "passes (stage + walk + flush), all but the tail": int((clk[:, 3] - clk[:, 1]).mean()),
                         "last flush": int((clk[:, 4] - clk[:, 3]).mean())},
    "entries_per_wave_mean": round(float(n_ent.mean()), 1),
}
print(json.dumps(out))
