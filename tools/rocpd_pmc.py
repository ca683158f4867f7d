#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 (ROCm 7.x) rocpd SQLite database."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else None
    if view is None:
        print("no counters_collection view; tables:", [t for t in tabs if "pmc" in t or "counter" in t])
        return
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % view)]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select %s, counter_name, count(*), avg(value) from %s group by %s, counter_name"
                     % (kcol, view, kcol)).fetchall()
    by = {}
    for k, n, cnt, avg in rows:
        by.setdefault(k, {})[n] = (cnt, avg)
    lines = []
    for k in sorted(by, key=lambda k: -max(v[1] for v in by[k].values())):
        if "at::native" in k or "rocclr" in k or "rocsolver" in k or "Cijk" in k:
            continue
        lines.append(k[:90])
        for n, (cnt, avg) in sorted(by[k].items()):
            lines.append("    %-28s calls %4d  avg %16.1f" % (n, cnt, avg))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
