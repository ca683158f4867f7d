"""The headline loop (Python API, tickets, three frames in flight on three streams) with the three streams at DIFFERENT
priorities: does a fixed order between the streams break their lock-step (profiles/r05_timeline_c3.txt: three blends
together, then three K1, then the middle kernels)?
    gpurun -- 'python tools/stream_priority_experiment.py > gpurun_out/r05_stream_priority_experiment.jsonl'"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gaussiancity_amd import synth
from gaussiancity_amd.rasterizer import GaussianRasterizer, GaussianRasterizerWrapper
dev = torch.device("cuda:0")
cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg, sc = synth.make_scene(cfgname); W, H = cfg["W"], cfg["H"]
wr = GaussianRasterizerWrapper(synth.intrinsics(W, H), (W, H), device=dev)
cams = [wr._get_gaussian_rasterization_settings(p, q)._replace(sh_degree=cfg["sh_degree"]) for p, q in synth.orbit_poses()]
t = {k: torch.from_numpy(v).to(dev) for k, v in sc.items() if isinstance(v, np.ndarray)}
rasters = [GaussianRasterizer(rs) for rs in cams]
means2D = torch.zeros_like(t["means3D"])
try:
    rng = torch.cuda.Stream.priority_range()
except Exception as e:  # noqa: BLE001
    rng = str(e)
def run(streams, n):
    with torch.no_grad():
        for i in range(n):
            with torch.cuda.stream(streams[i % len(streams)]):
                rasters[i % 24](means3D=t["means3D"], means2D=means2D, opacities=t["opacities"], scales=t["scales"],
                                rotations=t["rotations"], shs=t["shs"])
    torch.cuda.synchronize()
for pri in ((0, 0, 0), (-1, 0, 0), (-1, -1, 0), (-1, 0, 1), (0, 0, 0), (-1, 0, 0), (-1, 0, 1), (0, 0, 0, 0), (-1, -1, 0, 0)):
    try:
        streams = [torch.cuda.Stream(device=dev, priority=p) for p in pri]
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"priorities": pri, "error": str(e)[:120]})); continue
    run(streams, 96)
    walls = []
    for rep in range(5):
        t0 = time.perf_counter(); run(streams, 600); walls.append((time.perf_counter() - t0) / 600 * 1e6)
    print(json.dumps({"config": cfgname, "priorities": pri, "actual": [s.priority for s in streams], "priority_range": rng,
                      "period_us": round(float(np.median(walls)), 1), "frames_per_s": round(1e6 / float(np.median(walls)), 0),
                      "min_us": round(min(walls), 1)}), flush=True)
