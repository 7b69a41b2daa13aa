#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 --pmc CSV (counter_collection.csv)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    for c, v in cs.items():
        print(f"{k[:60]:60s} {c:14s} n={len(v):5d} avg={sum(v)/len(v):14.1f}")
