#!/bin/bash
# Run on the GPU box: rocprofv3 kernel trace of a short bench run WITH the stepper blocks (the persistent
# column stepper, the fused stages, the adaptive loop); prints the per-kernel statistics.
# usage: tools/gpu_trace_stepper.sh <tag> [bench args]
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-contract-regime $* > $OUT/bench.log 2>&1 || exit 1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $f $OUT/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/kernel_stats.csv")))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:8]:
    print("%-110s calls=%-4s avg_us=%10.1f pct=%s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
