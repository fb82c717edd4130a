#!/usr/bin/env python3
"""Per dispatch (in order, repeats of one kernel with equal counts folded) the counters of a rocprofv3 --pmc run.
usage: pmc_dispatches.py <dir> <kernel name part> [values per dispatch, for per-512-value-tile figures]"""
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
per = float(sys.argv[3]) / 512 if len(sys.argv) > 3 else None
rows = collections.defaultdict(dict)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
last = None
for k in sorted(rows):
    c = rows[k]
    key = tuple(round(v, -4) for _, v in sorted(c.items()) if _.startswith("SQ_INSTS"))
    if key == last:
        continue
    last = key
    if per:
        print(k, " ".join("%s=%.0f" % (n.replace("SQ_", ""), v / per) for n, v in sorted(c.items())), "(per 512 values)")
    else:
        print(k, " ".join("%s=%.4g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
