#!/usr/bin/env python
"""One K-frame burst of `bench.py --steps K` in a rocprofv3 --kernel-trace database: the kernels between two idle gaps
of the GPU (the barrier + synchronize around a timed block), per stream, relative to the burst's first kernel.
    python tools/burst_timeline.py <db> [which_burst_from_the_end=3]"""
import re
import sqlite3
import sys


def main(db, which=3):
    c = sqlite3.connect(db)
    rows = c.execute("select name,queue_id,start,end from kernels order by start").fetchall()
    rows = [r for r in rows if "k_" in r[0] and "anonymous" in r[0]]
    bursts, cur, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[2] - last_end > 40000 and cur:   # > 40 us of idle GPU: a block boundary
            bursts.append(cur)
            cur = []
        cur.append(r)
        last_end = max(last_end or 0, r[3])
    if cur:
        bursts.append(cur)
    bursts = [b for b in bursts if 100 <= len(b) <= 140]   # 20 frames x 6 kernels
    b = bursts[-int(which)]
    t0 = b[0][2]
    print("burst of %d kernels, %.1f us from first start to last end" % (len(b), (max(r[3] for r in b) - t0) / 1e3))
    for r in b:
        m = re.search(r"k_\w+(<[\w, ]+>)?", r[0])
        print("%-34s q%-2d start %8.1f end %8.1f dur %7.1f" % (m.group(0), r[1], (r[2] - t0) / 1e3, (r[3] - t0) / 1e3, (r[3] - r[2]) / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:])
