#!/usr/bin/env python3
"""Average each PMC counter over the dispatches of one kernel in rocprofv3 counter_collection CSVs."""
import csv, sys, collections, glob
pat = sys.argv[1]
acc = collections.defaultdict(list)
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
