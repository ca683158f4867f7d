#!/usr/bin/env python
"""Per-launch averages of every collected counter for the kernels whose name contains <pattern>
(rocprofv3 --pmc rocpd database)."""
import sqlite3
import sys


def main(db, pattern):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select counter_name, avg(value), count(*) from counters_collection where %s like ? "
                     "group by counter_name" % kcol, ("%" + pattern + "%",)).fetchall()
    for name, v, n in rows:
        print("%-28s %16.0f   (n=%d)" % (name, v, n))


if __name__ == "__main__":
    main(*sys.argv[1:])
