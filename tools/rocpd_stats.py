#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.x) rocpd SQLite database into a per-kernel stats table
(the same content as `--stats` CSV output): calls, total/avg/min/max duration, share."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["%-72s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct")]
    for n, k, s, a, mn, mx in rows:
        lines.append("%-72s %8d %14d %12.0f %12d %12d %6.2f%%" % (n[:72], k, s, a, mn, mx, 100.0 * s / tot))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
