#!/usr/bin/env python3
"""Per-kernel statistics from a rocprofv3 rocpd SQLite database (the default output format of
`rocprofv3 --kernel-trace --stats`), printed as CSV: name, calls, total_ns, avg_ns, min_ns, max_ns, pct."""
import re
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    rows = db.execute(
        "select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
        "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col)
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
    for name, calls, tot, mn, mx in rows:
        name = re.sub(r"\s+", " ", name)
        print('"%s",%d,%d,%.1f,%d,%d,%.2f' % (name, calls, tot, tot / calls, mn, mx, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
