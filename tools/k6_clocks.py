"""When and where the workgroups of the forward blend (K6) run at C3: is the launch's length set by the work or by its
tail?  Needs the experiment build:  make -C gaussiancity_amd/csrc experiments  (tools/_build/libgcr_hip_exp.so).

    python tools/k6_clocks.py [--config C3] [--pose 0]

Reading the numbers: a workgroup's life and the per-CU residency are clock reads only; the phase split of wave 0 puts
`s_waitcnt vmcnt(0)` in front of the hop clocks, which also waits for the tool's OWN clock stores -- the "ranges -> ids"
hop (4.3-4.5 us) is an upper bound, and the tile-head experiment (profiles/r04_tile_head_experiment.jsonl: that hop
removed, nothing gained) says how loose.
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
ap.add_argument("--config", default="C3")
ap.add_argument("--pose", type=int, default=0)
args = ap.parse_args()
dev = torch.device("cuda:0")
E = torch.Tensor([])
cfg, sc = synth.make_scene(args.config)
W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
L = N.lib()
L.gcr_debug_set_clock_buffer.argtypes = [C.c_void_p]
L.gcr_debug_set_clock_buffer.restype = None
rs = cams[args.pose]
a = (rs.bg, t["means3D"], E, t["opacities"], t["scales"], t["rotations"], rs.scale_modifier, E, rs.view_matrix, rs.proj_matrix,
     rs.tanfovx, rs.tanfovy, rs.img_h, rs.img_w, t["shs"], cfg["sh_degree"], rs.campos, False, False)
for _ in range(5):
    R = ext.rasterize_gaussians(*a, _for_backward=False)[0]
T = ((W + 15) // 16) * ((H + 15) // 16)
buf = torch.zeros((T, 16), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
N.set_option("timing", 1)
N.stage_ms()
L.gcr_debug_set_clock_buffer(buf.data_ptr())
ext.rasterize_gaussians(*a, _for_backward=False)
torch.cuda.synchronize()
L.gcr_debug_set_clock_buffer(None)
st = N.stage_ms()
N.set_option("timing", 0)
b = buf.cpu().numpy().astype(np.uint64)
np.save(os.path.join(ROOT, "gpurun_out", "k6_clocks_%s.npy" % args.config), b)
ran = b[:, 2] != 0
hw, t0, t1, n = b[ran, 0], b[ran, 1].astype(np.int64), b[ran, 2].astype(np.int64), b[ran, 3].astype(np.int64)
tile = np.nonzero(ran)[0]
lo32 = (hw & np.uint64(0xffffffff)).astype(np.int64)
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
cu = xcc * (1 << 16) + (lo32 & 0xff00)
# the counter's base differs between CUs: every CU's first entry clock is its 0
cus = np.unique(cu)
span, busy, nblk, first_idle = [], [], [], []
for x in cus:
    m = cu == x
    s0 = t0[m].min()
    a0, a1 = t0[m] - s0, t1[m] - s0
    span.append(a1.max())
    nblk.append(int(m.sum()))
    # block-residency integral and the moment the CU last had seven workgroups
    ev = sorted([(v, 1) for v in a0] + [(v, -1) for v in a1])
    cur, last7, integ, prev = 0, 0, 0, 0
    for v, d in ev:
        integ += cur * (v - prev)
        prev = v
        cur += d
        if cur >= 7:
            last7 = v
    busy.append(integ)
    first_idle.append(last7)
span, busy, first_idle = np.array(span), np.array(busy), np.array(first_idle)
life = t1 - t0
ms = st["blend_fwd"]
tick_us = span.max() / (ms * 1e3)
order = np.argsort(t0 - np.array([t0[cu == x].min() for x in cu]))  # (per-CU time of entry)
out = {
    "config": args.config, "R": int(R), "blend_fwd_stage_ms": round(ms, 4), "workgroups": int(len(tile)), "cus_seen": int(len(cus)),
    "ticks_per_us_if_longest_cu_span_is_the_stage": round(float(tick_us), 1),
    "cu_span_over_longest_p0_p10_p50_p90": [round(float(np.percentile(span, q) / span.max()), 3) for q in (0, 10, 50, 90)],
    "cu_mean_span_over_longest": round(float(span.mean() / span.max()), 3),
    "mean_resident_workgroups_per_cu_over_its_span": round(float((busy / span).mean()), 2),
    "mean_resident_workgroups_per_cu_over_the_longest_span": round(float((busy / span.max()).mean()), 2),
    "cu_has_7_workgroups_until_fraction_of_longest_p10_p50_p90": [round(float(np.percentile(first_idle, q) / span.max()), 3) for q in (10, 50, 90)],
    "workgroups_per_cu_min_mean_max": [int(min(nblk)), round(float(np.mean(nblk)), 1), int(max(nblk))],
    "workgroup_life_us_mean_p50_p90_max": [round(float(v / tick_us), 2) for v in (life.mean(), np.percentile(life, 50), np.percentile(life, 90), life.max())],
    "list_length_mean_p90_max": [round(float(n.mean()), 1), int(np.percentile(n, 90)), int(n.max())],
    "corr_life_vs_list_length": round(float(np.corrcoef(life, n)[0, 1]), 3),
}
# phases of wave 0's first chunk (tiles whose wave 0 walked)
ph = b[ran]
ok = (ph[:, 4] != 0) & (ph[:, 5] != 0) & (ph[:, 6] != 0)
p = ph[ok].astype(np.int64)
steps = p[:, 7]
seg = {"entry -> ranges here": p[:, 10] - p[:, 1], "ranges -> ids here (thread 0)": p[:, 8] - p[:, 10],
       "ids -> records here (thread 0)": p[:, 9] - p[:, 8], "records -> staged (masks, LDS stores, barrier)": p[:, 4] - p[:, 9],
       "entry -> records staged (ranges, ids, records, masks, barrier)": p[:, 4] - p[:, 1], "lists built": p[:, 5] - p[:, 4],
       "walk": p[:, 6] - p[:, 5], "walk done -> exit (vote, stores)": p[:, 2] - p[:, 6]}
out["wave0_first_chunk_phase_us_mean_p50_p90"] = {k: [round(float(v.mean() / tick_us), 2), round(float(np.percentile(v, 50) / tick_us), 2),
                                                     round(float(np.percentile(v, 90) / tick_us), 2)] for k, v in seg.items()}
out["wave0_steps_mean"] = round(float(steps.mean()), 1)
out["wave0_walk_ns_per_step"] = round(float((seg["walk"].sum() / max(steps.sum(), 1)) / tick_us * 1e3), 1)
print(json.dumps(out))
