#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 results database (rocpd format): calls, average / min / max
duration in microseconds, optionally the dispatches in launch order.
usage: tools/rocpd_kernels.py <results.db> [--list N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    names = dict(cur.execute("select id, %s from %s" % (name_col, ks)))
    rows = list(cur.execute('select kernel_id, start, "end", grid_size_x, workgroup_size_x from %s order by start' % kd))
    agg = {}
    for kid, s, e, g, w in rows:
        a = agg.setdefault(names.get(kid, str(kid)), [])
        a.append((e - s) / 1e3)
    print("%-90s %7s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-90s %7d %10.2f %10.2f %10.2f" % (k[:90], len(v), sum(v) / len(v), min(v), max(v)))
    if "--list" in sys.argv:
        n = int(sys.argv[sys.argv.index("--list") + 1])
        for kid, s, e, g, w in rows[:n]:
            print("%12.2f us  grid %8d x %4d  %s" % ((e - s) / 1e3, g, w, names.get(kid, str(kid))[:100]))


if __name__ == "__main__":
    main()
