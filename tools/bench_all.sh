#!/bin/bash
# Run on the GPU box (via gpurun): one bench line per workload into gpurun_out/<tag>_bench_all.jsonl
# (+ C2 with the SURVEY 8(d) contract traffic restored and C3 with placement tuning).
# usage: tools/bench_all.sh <tag>
TAG=${1:-r}
OUT=gpurun_out/${TAG}_bench_all.jsonl
: > $OUT
run() { python bench.py --no-cpu-baseline "$@" >> $OUT 2>> gpurun_out/${TAG}_bench_all.err || echo "{\"failed\": \"$*\"}" >> $OUT; }
for w in c1 c2 c3 c4 c5 f3v64 f3v64p f3c32 f3c64 f3c32s f3c64s; do run --workload $w; done
run --workload c2 --no-known-zero --no-stepper
run --workload c3 --no-known-zero --no-stepper
run --workload c3 --placement-tune --no-stepper
python - $OUT <<'PY'
import json, sys
print("%-7s %-14s %10s %8s %8s %8s %8s %6s %6s %8s %8s %8s %9s %7s %8s" % ("wl", "flags", "value", "ms/step", "kern", "kmin", "fused", "frac", "kfrac", "probe", "step", "stages", "contr_ms", "cfrac", "adaptive"))
for line in open(sys.argv[1]):
    d = json.loads(line)
    if "failed" in d:
        print("FAILED", d["failed"]); continue
    r = d["roofline"]
    wl = d["config"]["workload"].split(":")[0]
    flags = ("zero=0 " if not r["theta_i_known_zero"] and r["bytes_moved_per_cell"] >= r["bytes_per_cell"] - 1 else "") + ("tuned" if d.get("placement_tuning") else "")
    ct = d.get("contract_traffic") or {}
    print("%-7s %-14s %10.3e %8.4f %8.4f %8.4f %8.4f %6.3f %6.3f %8.4f %8s %8s %9s %7s %8s" % (
        wl, flags, d["value"], d["ms_per_step"], r["kernel_ms"], r["kernel_ms_min"] or 0, r["fused_dt_kernel_ms"] or 0, r["frac"],
        r["kernel_frac"], (r["stream_probe"] or {}).get("ms", 0),
        ("%.4f" % d["ssprk33"]["ms_per_step"]) if "ssprk33" in d else "-",
        ("%.4f" % d["ssprk33_fused_stages"]["ms_per_step"]) if "ssprk33_fused_stages" in d else "-",
        ("%.4f" % ct["kernel_ms"]) if "kernel_ms" in ct else "-", ("%.3f" % ct["frac"]) if "frac" in ct else "-",
        ("%.4f" % d["adaptive_ssprk33"]["ms_per_step"]) if "adaptive_ssprk33" in d else "-"))
PY
