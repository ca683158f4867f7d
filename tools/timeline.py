#!/usr/bin/env python
"""Per-stream kernel timeline from a rocprofv3 --kernel-trace rocpd database.

    python tools/timeline.py <db> [first] [count] [everything]

Prints a window of the kernel launches (name, queue, start, end, duration) and, for the steady state, the numbers
VERDICT r03 asked for: the distance between consecutive forward-blend starts (the frame period), the gap between the
END of one frame's K1 (k_preprocess_fused) and the START of the next frame's K1 -- the time the GPU waits for a host
that has to learn num_rendered before it enqueues the next frame --, how often two K1 launches overlap, and the
fraction of the period every HIP queue is busy.  With a `--hip-trace` database the host API calls of the same window
are listed too (table `regions`, when present).
"""
import re
import sqlite3
import sys


def _short(name):
    m = re.search(r"k_\w+(<[\w, ]+>)?", name)
    return m.group(0) if m else name[-28:]


def analyse(rows):
    """rows: (name, queue, start_ns, end_ns) sorted by start.  Returns a dict of steady-state figures (us)."""
    out = {}
    bl = [r[2] for r in rows if "k_blend_fwd" in r[0]]
    skip = min(20, len(bl) // 4)
    d = sorted((b - a) / 1e3 for a, b in zip(bl[skip:-1], bl[skip + 1:]))
    if d:
        out["blend_start_period_us"] = {"median": d[len(d) // 2], "mean": sum(d) / len(d), "n": len(d)}
    k1 = [(r[2], r[3]) for r in rows if "k_preprocess_fused" in r[0] or "k_preprocess_exact" in r[0]]
    k1 = k1[skip:]
    gaps, overlaps = [], 0
    for (s0, e0), (s1, e1) in zip(k1[:-1], k1[1:]):
        g = (s1 - e0) / 1e3
        gaps.append(g)
        overlaps += 1 if g < 0 else 0
    if gaps:
        gs = sorted(gaps)
        out["k1_end_to_next_k1_start_us"] = {"median": gs[len(gs) // 2], "mean": sum(gs) / len(gs),
                                             "p10": gs[len(gs) // 10], "p90": gs[(9 * len(gs)) // 10],
                                             "overlapping_pairs": overlaps, "n": len(gs)}
        ds = sorted((e - s) / 1e3 for s, e in k1)
        out["k1_duration_us"] = {"median": ds[len(ds) // 2], "mean": sum(ds) / len(ds)}
    # busy fraction per queue over the steady-state window, and of the union (any kernel running)
    if len(rows) > 40:
        win = rows[len(rows) // 4: (3 * len(rows)) // 4]
        t0, t1 = win[0][2], max(r[3] for r in win)
        per_q = {}
        for r in win:
            per_q[r[1]] = per_q.get(r[1], 0) + (r[3] - r[2])
        out["queue_busy_fraction"] = {"q%d" % q: round(v / (t1 - t0), 3) for q, v in sorted(per_q.items())}
        ev = sorted([(r[2], 1) for r in win] + [(r[3], -1) for r in win])
        depth, last, busy, weighted = 0, t0, 0, 0
        for t, dlt in ev:
            if depth > 0:
                busy += t - last
                weighted += (t - last) * depth
            last = t
            depth += dlt
        out["any_kernel_running_fraction"] = round(busy / (t1 - t0), 3)
        out["mean_kernels_in_flight"] = round(weighted / (t1 - t0), 3)
    return out


def main(db, first=98, count=30, everything=0):
    c = sqlite3.connect(db)
    rows = c.execute("select name,queue_id,stream_id,start,end,grid_x,workgroup_x from kernels order by start").fetchall()
    ours = [r for r in rows if int(everything) or ("anonymous" in r[0] and "k_" in r[0])]
    first, count = int(first), int(count)
    first = min(first, max(0, len(ours) - count))
    t0 = ours[first][3]
    for r in ours[first:first + count]:
        print("%-32s q%-2d start %8.1f end %8.1f dur %7.1f" % (_short(r[0]), r[1], (r[3] - t0) / 1e3, (r[4] - t0) / 1e3,
                                                                (r[4] - r[3]) / 1e3))
    res = analyse([(r[0], r[1], r[3], r[4]) for r in ours])
    for k, v in res.items():
        if isinstance(v, dict):
            print(k + ": " + "  ".join("%s %s" % (kk, ("%.1f" % vv) if isinstance(vv, float) else vv) for kk, vv in v.items()))
        else:
            print("%s: %s" % (k, v))
    # host API calls of the same window (rocprofv3 --hip-trace), when the database has them
    try:
        tabs = [t[0] for t in c.execute("select name from sqlite_master where type in ('table','view')").fetchall()]
        reg = [t for t in tabs if t.startswith("regions")]
        if reg and "regions" in tabs:
            w0, w1 = ours[first][3] - 300000, ours[min(first + count, len(ours)) - 1][4]
            api = c.execute("select name,start,end from regions where start>=? and start<=? order by start", (w0, w1)).fetchall()
            if api:
                print("host API calls in the window (us relative to the first kernel's start):")
                for n, s, e in api[:400]:
                    print("  %-36s start %8.1f dur %6.1f" % (str(n)[:36], (s - t0) / 1e3, (e - s) / 1e3))
    except sqlite3.Error as e:  # the schema differs between rocprofv3 versions: the kernel timeline above is what matters
        print("(no host API table: %s)" % e)


if __name__ == "__main__":
    main(*sys.argv[1:])
