#!/usr/bin/env python
"""Print a window of the per-stream kernel timeline from a rocprofv3 --kernel-trace rocpd database."""
import re
import sqlite3
import sys


def main(db, first=98, count=30, everything=0):
    c = sqlite3.connect(db)
    rows = c.execute("select name,queue_id,stream_id,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
    ours = [r for r in rows if int(everything) or ("anonymous" in r[0] and "k_" in r[0])]
    first, count = int(first), int(count)
    t0 = ours[first][3]
    busy_end = None
    for r in ours[first:first + count]:
        m = re.search(r"k_\w+(<\w+>)?", r[0])
        print("%-28s q%-2d start %8.1f end %8.1f dur %7.1f" % (m.group(0) if m else r[0][-28:], r[1], (r[3] - t0) / 1e3, (r[4] - t0) / 1e3, (r[4] - r[3]) / 1e3))
    # steady-state period: distance between consecutive blend starts
    bl = [r[3] for r in ours if "k_blend_fwd" in r[0]]
    d = [(b - a) / 1e3 for a, b in zip(bl[20:-1], bl[21:])]
    if d:
        d.sort()
        print("blend-start period us: median %.1f  mean %.1f  n %d" % (d[len(d) // 2], sum(d) / len(d), len(d)))


if __name__ == "__main__":
    main(*sys.argv[1:])
