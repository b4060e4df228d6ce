#!/usr/bin/env python3
"""Mean PMC value per (kernel, counter) from a rocprofv3 --pmc csv directory.  usage: pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub in k:
            acc[(k[:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(f"{k:60s} {c:28s} n={len(v):4d} mean={sum(v)/len(v):14.1f}")
