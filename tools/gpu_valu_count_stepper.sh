#!/bin/bash
# PMC pass over a bench run WITH the persistent stepper: VALU / LDS instructions per wave of column_stepper_*
# usage (GPU box): tools/gpu_valu_count_stepper.sh <tag> [bench flags]
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-contract-regime $* > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "column_stepper" in k:
        w=sum(v["SQ_WAVES"])/len(v["SQ_WAVES"])
        print(k[:120], "launches=%d waves=%d"%(len(v["SQ_WAVES"]),w), " ".join("%s/wave=%.1f"%(c,sum(v[c])/len(v[c])/w) for c in ("SQ_INSTS_VALU","SQ_INSTS_LDS","SQ_INSTS_SALU")))
PY
grep -o '"ssprk33": {[^}]*}' $OUT/bench.log
