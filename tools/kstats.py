#!/usr/bin/env python
"""Per-kernel call count / average / min / max duration (us) from a rocprofv3 --kernel-trace rocpd database."""
import re
import sqlite3
import sys


def main(db, pattern="k_"):
    c = sqlite3.connect(db)
    rows = c.execute("select name,count(*),avg(end-start),min(end-start),max(end-start) from kernels group by name "
                     "order by sum(end-start) desc").fetchall()
    for name, n, avg, lo, hi in rows:
        if pattern in name:
            m = re.search(r"k_\w+(<[\w, ]+>)?", name)
            print("%-34s n %5d  avg %8.1f  min %8.1f  max %8.1f" % (m.group(0) if m else name[:34], n, avg / 1e3, lo / 1e3,
                                                                    hi / 1e3))


if __name__ == "__main__":
    main(*sys.argv[1:])
