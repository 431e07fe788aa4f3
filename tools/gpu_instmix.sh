#!/bin/bash
# Run on the GPU box (via gpurun): dynamic instruction mix and LDS activity of the rhs kernels of
# one bench workload (rocprofv3 PMC, one pass per counter group, no tracing).
# usage: tools/gpu_instmix.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-stepper --no-placement-tune --no-contract-regime $*"
pass() { # name counters...
    local name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B > $OUT/bench_$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/bench_$name.log; return 1; }
}
pass f64a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
pass f32a SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 || exit 1
pass ints SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
pass lds SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT || exit 1
pass busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1].get("SQ_INSTS_VALU", []))):
    if "rhs_kernel" not in k and "column_stepper" not in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = m.get("SQ_WAVES", 0)
    print("==", k[:130], " launches=%d waves=%d" % (len(v.get("SQ_INSTS_VALU", [])), waves))
    tot = m.get("SQ_INSTS_VALU", 0)
    for c in sorted(m):
        extra = ""
        if c.startswith("SQ_INSTS") and tot:
            extra = "  (%.1f %% of VALU)" % (100 * m[c] / tot) if "VALU_" in c else ""
        print("   %-28s %.6g%s" % (c, m[c], extra))
    named = sum(m.get(c, 0) for c in m if c.startswith("SQ_INSTS_VALU_"))
    if tot:
        print("   -> VALU not in any named class (moves, selects, compares, lane ops): %.6g (%.1f %%)" % (tot - named, 100 * (tot - named) / tot))
PY
