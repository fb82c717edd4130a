#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc CSV: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat and pat not in k: continue
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k[:90])
    for name, vals in sorted(c.items()):
        print("   %-28s mean/dispatch = %.4g   (n=%d)" % (name, sum(vals)/len(vals), len(vals)))
