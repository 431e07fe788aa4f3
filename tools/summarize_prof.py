#!/usr/bin/env python3
"""Summarise a tools/gpu_profile.sh output directory: per-kernel average duration
(kernel trace) and per-launch PMC counters of the dominant kernel, with the
gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE counts 1/2 of a wide
coalesced read; WRITE_SIZE is exact; both in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(d, pattern), recursive=True):
        with open(f) as fh:
            yield from csv.DictReader(fh)


stats = list(rows("trace/**/*kernel_stats.csv"))
stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== kernel trace (rocprofv3 --kernel-trace --stats)")
for r in stats[:4]:
    print(f"  {r['Name'][:110]}  calls={r['Calls']} avg_ns={float(r['AverageNs']):.0f} pct={r['Percentage']}")
dom = stats[0]["Name"] if stats else None
print("== dominant kernel:", dom[:100] if dom else None)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_grbm"):
    acc = defaultdict(list)
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if r.get("Kernel_Name") == dom:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"  {k}: mean/launch={sum(v)/len(v):.6g} (n={len(v)})")
        if k == "FETCH_SIZE":
            print(f"    -> HBM read bytes/launch (bytes; FETCH_SIZE KiB x 1024 x 2, the gfx950 wide-read correction): {2*1024*sum(v)/len(v):.6g}")
        if k == "WRITE_SIZE":
            print(f"    -> HBM write bytes/launch (bytes; WRITE_SIZE KiB x 1024): {1024*sum(v)/len(v):.6g}")
for f in glob.glob(os.path.join(d, "bench_trace.log")):
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line)
            print("== bench line (profiled run):", json.dumps({k: j[k] for k in ("value", "ms_per_step", "roofline")}))
