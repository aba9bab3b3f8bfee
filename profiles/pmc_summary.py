#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel (matching a substring) sum of each counter over dispatches."""
import csv, glob, sys, collections
root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} dispatches={n[k]:3d} sum={acc[k]:.6g} per_dispatch={acc[k]/n[k]:.6g}")
