#!/usr/bin/env python3
"""Kernels of the last plan run in a rocprofv3 kernel trace, in order, with start (us from the first) and duration.
usage: timeline.py <kernel_trace.csv> [name of the kernel a run starts with, default rebase_kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "rebase_kernel"
seq = [(r["Kernel_Name"].split("(")[0].replace("pq::", "").replace("void ", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, s in enumerate(seq) if s[0].startswith(first)]
i0 = idx[-1] if idx else 0
t0 = seq[i0][1]
for name, a, b in seq[i0:]:
    print("%-34s start %9.1f dur %9.1f" % (name[:34], (a - t0) / 1000, (b - a) / 1000))
