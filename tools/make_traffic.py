#!/usr/bin/env python
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE rocpd databases of
tools/collect_profiles.sh: per bench.py stage, what ONE FRAME's launches of the stage's kernels move -- every kernel's
per-launch average times its launches, over the launches of the stage's ANCHOR kernels (the ones of which a frame
launches exactly one: alternative instantiations of one kernel -- the streaming cull with and without the non-temporal
policy, the forward blend with and without backward state -- average by their launch counts; kernels a frame launches
besides -- column scan, band sort, record zeroing -- add).  (Until round 6 the averages were simply summed: correct
while every stage ran one instantiation, a double count since two of the cull run in one session.)

usage: make_traffic.py <fetch.db> <write.db> <out.json> [<sq.db> [<workload label>]]
"""
import hashlib
import json
import os
import re
import sqlite3
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gaussiancity_amd", "csrc")


def kernel_source_hashes():
    """sha256[:16] of every kernel source: bench.py compares them with the tree it runs from, so that per-launch
    instruction counts taken from a committed file are flagged when the kernel has changed since."""
    out = {}
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")) and f.startswith("gcr_"):
            out[f] = hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest()[:16]
    return out

# stage -> substrings of its anchor kernels' names (one launch of ONE of them per frame)
ANCHORS = {
    "preprocess": ("k_preprocess_fused", "k_preprocess_cull"),
    "scan": ("k_tile_table<false", "k_tile_table<0", "k_tile_tableILb0"),
    "emit": ("k_tile_table<true", "k_tile_table<1", "k_tile_tableILb1"),
    "sort": ("k_tile_sort",),
    "blend_fwd": ("k_blend_fwd",),
    "blend_bwd": ("k_blend_bwd",),
    "preprocess_bwd": ("k_preprocess_bwd",),
}
# bench.py stage -> substrings of the kernel names launched inside that stage timer (gcr_api.hip)
STAGES = {
    "preprocess": ("k_preprocess_fused", "k_preprocess_cull", "k_preprocess_project"),
    "scan": ("k_tile_table<false", "k_tile_table<0", "k_tile_tableILb0", "k_table_colscan", "k_band<"),   # (<false, 1024>: no ">"; k_band: the band sort of big frames, round 6)
    "emit": ("k_tile_table<true", "k_tile_table<1", "k_tile_tableILb1"),
    "sort": ("k_tile_sort",),
    "blend_fwd": ("k_blend_fwd",),
    "blend_bwd": ("k_blend_bwd", "k_zero_grad_records"),
    "preprocess_bwd": ("k_preprocess_bwd",),
}
SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU",
               "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, avg(value), count(*) from counters_collection where counter_name=? group by %s"
                     % (kcol, kcol), (counter,)).fetchall()
    return {k: (v, n) for k, v, n in rows}


def main(fetch_db, write_db, out, sq_db=None, workload="C3 (5M S-city, 1920x1080, SH3, forward)"):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    sq = {c: per_kernel(sq_db, c) for c in SQ_COUNTERS} if sq_db else {}
    def per_frame(tab, stage):
        """sum over the stage's kernels of (average per launch x launches) / launches of the stage's anchor kernels"""
        pats, anchors = STAGES[stage], ANCHORS[stage]
        frames = sum(n for k, (_, n) in tab.items() if any(p in k for p in anchors))
        total = sum(v * n for k, (v, n) in tab.items() if any(p in k for p in pats))
        return total / frames if frames else 0.0

    kernels = {}
    for stage, pats in STAGES.items():
        names = sorted(set(re.search(r"k_\w+(<[\w, ]+>)?", k).group(0) for k in f if any(p in k for p in pats)))
        if not names:
            continue
        kernels[stage] = {"FETCH_SIZE_KB": round(per_frame(f, stage), 1), "WRITE_SIZE_KB": round(per_frame(w, stage), 1),
                          "kernels": names,
                          "launches": {re.search(r"k_\w+(<[\w, ]+>)?", k).group(0): n for k, (_, n) in sorted(f.items())
                                       if any(p in k for p in pats)}}
        for c, tab in sq.items():
            if tab:
                kernels[stage][c] = round(per_frame(tab, stage))
    doc = {
        "workload": workload,
        "unit": "KB per frame and stage (rocprofv3 FETCH_SIZE / WRITE_SIZE per-dispatch averages x launches, over the launches of the stage's anchor kernel: alternative instantiations average, additional kernels add)",
        "note": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024: MI355X_MICROARCH.md says FETCH_SIZE reports half the "
                "bytes of 16-B/lane reads on gfx950; raw values kept here",
        "kernels": kernels,
        "kernel_source_sha16": kernel_source_hashes(),
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:6])
