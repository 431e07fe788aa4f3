#!/bin/bash
# quick PMC pass: VALU instructions per launch of the dominant kernel + kernel time
OUT=$GRAFT_REPO_ROOT/gpurun_out/valu_$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stepper --no-contract-regime $* > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "rhs_kernel" in k or "column_stepper" in k:
        iv=sum(v["SQ_INSTS_VALU"])/len(v["SQ_INSTS_VALU"]); w=sum(v["SQ_WAVES"])/len(v["SQ_WAVES"])
        print(k, "VALU/launch=%.4g waves=%d VALU per wave=%.1f"%(iv,w,iv/w))
PY
grep -o '"kernel_ms": [0-9.]*' $OUT/bench.log
