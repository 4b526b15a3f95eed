#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: mean per dispatch per kernel."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print("   %-28s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
