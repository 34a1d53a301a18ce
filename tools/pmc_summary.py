#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch."""
import csv, glob, collections, sys, os
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if pat and pat not in k:
            continue
        acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print("%-28s %-62s n=%3d mean=%.4g" % (c, k, len(v), sum(v) / len(v)))
