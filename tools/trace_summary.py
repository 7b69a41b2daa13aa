#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel count/avg/min/max (us) and the
idle gap between consecutive dispatches of the steady-state region."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stats = defaultdict(list)
for r in rows:
    stats[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s}")
for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[:70]:70s} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {sum(v)/1e3:9.2f}")
# gaps in the densest stretch: last 40% of dispatches
n = len(rows)
seg = rows[int(n * 0.3):int(n * 0.7)]
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(seg, seg[1:])]
busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in seg) / 1e3
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"steady segment: {len(seg)} dispatches, span {span:.1f} us, busy {busy:.1f} us ({100*busy/span:.1f}%), "
      f"mean gap {sum(gaps)/len(gaps):.2f} us")
