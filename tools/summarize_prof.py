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
# the dominant kernel OF THE PATH: bench.py's ancillary blocks (the no-arithmetic stream probe, buffer
# fills) can outweigh the tendency launches in a short profiled run
path = [r for r in stats if "stream_probe" not in r["Name"] and "rocclr" not in r["Name"]]
dom = (path or stats)[0]["Name"] if stats else None
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
bench_line = None
for f in glob.glob(os.path.join(d, "bench_trace.log")):
    for line in open(f):
        if line.startswith("{"):
            bench_line = json.loads(line)
# The run starts with the placement trials of lh_tune_placement (the same kernel on other
# plane slots, some of them slow) and ends with the back-to-back block bench.py times with
# HIP events (roofline.kernel_ms over roofline.kernel_launches_timed launches): compare like with like.
disp = [r for r in rows("trace/**/*kernel_trace.csv") if r.get("Kernel_Name") == dom]
if disp and bench_line:
    disp.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = int(bench_line["roofline"].get("kernel_launches_timed", 0) or bench_line["roofline"].get("kernel_reps", 0)) or len(disp)
    last = disp[-n:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
    print(f"== dominant kernel, last {len(last)} dispatches (the HIP-event timed block of bench.py): "
          f"avg_ns={sum(dur) / len(dur):.0f} min_ns={min(dur)} max_ns={max(dur)}")
    alld = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in disp]
    print(f"   all {len(alld)} dispatches incl. placement trials and warm-up: avg_ns={sum(alld) / len(alld):.0f}")
if bench_line:
    print("== bench line (profiled run):", json.dumps({k: bench_line.get(k) for k in
                                                        ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min",
                                                         "roofline", "placement_tuning")}))
# machine-readable record for profiles/pmc_traffic.json
if bench_line and dom:
    rec = {"kernel": dom[:160]}
    for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ_INSTS_VALU"),
                     ("pmc_sq", "SQ_INSTS_SALU"), ("pmc_sq", "SQ_WAVES"), ("pmc_sq", "SQ_ACTIVE_INST_VALU"),
                     ("pmc_grbm", "GRBM_GUI_ACTIVE")):
        v = [float(r["Counter_Value"]) for r in rows(f"{sub}/**/*counter_collection.csv")
             if r.get("Kernel_Name") == dom and r["Counter_Name"] == key]
        if v:
            rec[key] = sum(v) / len(v)
    print("== record:", json.dumps(rec))
