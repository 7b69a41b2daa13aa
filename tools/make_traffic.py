#!/usr/bin/env python3
"""profiles/traffic.json from the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs):
bytes per streaming-pass launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts half of the 16-B/lane streaming reads;
WRITE_SIZE exact; both in KiB; L2 <-> fabric bytes, Infinity-Cache hits included).
    python tools/make_traffic.py FETCH_DIR WRITE_DIR ALGORITHMIC_BYTES OUT.json "source note" """
import csv, glob, json, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"source": sys.argv[5] if len(sys.argv) > 5 else "",
       "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md: FETCH_SIZE counts half of "
                     "16-B/lane streaming reads on gfx950; WRITE_SIZE exact; L2<->fabric bytes, Infinity-Cache hits included)",
       "per_kernel_bytes_per_launch": {}, "per_kernel_launches": {}}
tot = n = 0.0
for k in sorted(fetch):
    if k not in write:
        continue
    b = (2.0 * fetch[k][0] + write[k][0]) * 1024.0
    out["per_kernel_bytes_per_launch"][k] = b
    out["per_kernel_launches"][k] = fetch[k][1]
    if k.startswith("pass_kernel"):
        tot += b * fetch[k][1]; n += fetch[k][1]
out["atb_pass_kernel_bytes_per_launch"] = tot / n if n else None
out["algorithmic_bytes_per_launch"] = float(sys.argv[3])
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
