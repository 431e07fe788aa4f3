#!/usr/bin/env python3
"""Copy the judged parts of tools/gpu_profile.sh outputs (gpurun_out/prof_<tag>_<workload>/) into
profiles/ and (re)write profiles/pmc_traffic.json from their PMC records.

  tools/collect_profiles.py r2 c2 c3 c4 c5      # tag, workloads

profiles/<round>_<workload>_summary.txt        the rocprofv3 --kernel-trace --stats + PMC summary
profiles/<round>_<workload>_kernel_stats.csv   rocprofv3's per-kernel statistics
profiles/pmc_traffic.json                      HBM bytes / VALU per launch of the dominant kernel,
                                               read by bench.py (roofline.traffic, valu_per_cell)
"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, workloads = sys.argv[1], sys.argv[2:]
rnd = {"r2": "round2", "r3": "round3", "r3f": "round3"}.get(tag, tag)
out_json = os.path.join(ROOT, "profiles", "pmc_traffic.json")
try:
    table = json.load(open(out_json))
except (OSError, ValueError):
    table = {}
for w in workloads:
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{w}")
    summ = os.path.join(d, "summary.txt")
    text = open(summ).read()
    dst = os.path.join(ROOT, "profiles", f"{rnd}_{w}_summary.txt")
    nozero = w.endswith("_nozero")          # the SURVEY 8(d) byte contract as traffic: LH_TUNE zero=0
    wl = w[:-len("_nozero")] if nozero else w
    bargs = f"--workload {wl}" + (" --no-known-zero" if nozero else "")
    with open(dst, "w") as fh:
        fh.write(f"# bash tools/gpu_profile.sh {tag}_{w} {bargs}   (MI355X, rocprofv3; one pass per PMC group)\n")
        fh.write(f"# = python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stepper --no-contract-regime {bargs}\n")
        fh.write(text)
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{rnd}_{w}_kernel_stats.csv"))
    rec = bench = None
    for line in text.splitlines():
        if line.startswith("== record:"):
            rec = json.loads(line[len("== record:"):])
        if line.startswith("== bench line (profiled run):"):
            bench = json.loads(line[len("== bench line (profiled run):"):])
    if not rec or not bench:
        print("no record for", w)
        continue
    # sizes from the bench log of the traced run
    full = None
    for line in open(os.path.join(d, "bench_trace.log")):
        if line.startswith("{"):
            full = json.loads(line)
    ncols, nlev = full["config"]["columns_per_gpu"], full["config"]["levels"]
    cells = ncols * nlev
    read_b = 2.0 * 1024.0 * rec["FETCH_SIZE"]       # KiB, and the gfx950 wide-read factor 2
    write_b = 1024.0 * rec["WRITE_SIZE"]
    table[w] = {
        "ncols": ncols, "nlev": nlev, "known_zero": not nozero,
        "read_bytes": read_b, "write_bytes": write_b, "total_bytes": read_b + write_b,
        "valu_per_cell": rec["SQ_INSTS_VALU"] * 64.0 / cells,
        "salu_per_cell": rec["SQ_INSTS_SALU"] * 64.0 / cells,
        # SQ_ACTIVE_INST_VALU counts quad-cycles (a wave64 instruction holds its SIMD for 4 cycles);
        # GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the launch was active; 1024 SIMDs
        "valu_busy": (4.0 * rec["SQ_ACTIVE_INST_VALU"] / (1024.0 * rec["GRBM_GUI_ACTIVE"] / 8.0)
                      if rec.get("SQ_ACTIVE_INST_VALU") and rec.get("GRBM_GUI_ACTIVE") else None),
        "gui_active_cycles_per_xcd": rec.get("GRBM_GUI_ACTIVE", 0) / 8.0 or None,
        "kernel": rec["kernel"],
        "source": f"profiles/{rnd}_{w}_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, "
                  f"separate passes; FETCH_SIZE x2 gfx950 correction)",
        "command": f"python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stepper --no-contract-regime {bargs}",
    }
    print(w, {k: table[w][k] for k in ("total_bytes", "valu_per_cell", "salu_per_cell")},
          "moved by bench:", full["roofline"]["bytes_moved_per_launch"])
with open(out_json, "w") as fh:
    json.dump(table, fh, indent=1)
