"""PCIe-inclusive rate of the numpy drop-in extruder (host buffers in, host array out)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiancity_amd import points as P, synth
L = synth.s_layout(2048, 2001)
inv = {v: k for k, v in synth.LAYOUT_CLASSES.items()}
a = (True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, L["INS"], L["TD_HF"], L["BU_HF"], L["PTS"])
for i in range(4):
    t0 = time.perf_counter(); out = P.get_points_from_projection(*a); t1 = time.perf_counter()
    print("numpy drop-in: %.2f ms for %d points (%.1f MB out, 29.4 MB in)" % (1e3 * (t1 - t0), len(out), out.nbytes / 1e6))
dev = torch.device("cuda", 0)
t = [torch.from_numpy(L[k]).to(dev) for k in ("INS", "TD_HF", "BU_HF", "PTS")]
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = P.extrude_points(True, inv, synth.LAYOUT_SCALES, synth.LAYOUT_SEG_INS, *t); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("device-resident: %.3f ms" % (1e3 * (t1 - t0)))
