#!/bin/bash
# Run on the GPU box (via gpurun): same-process A/B of library builds on the workloads given.
# usage: LIBS="base product u1" tools/ab_round.sh <tag> [workload[:mode] ...]      mode = rhs | rhs_dt | step
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
LIBS=${LIBS:-base product}
for wm in "$@"; do
    w=${wm%%:*}; m=${wm##*:}; [ "$m" = "$wm" ] && m=rhs
    MODE=$m python tools/ab_libs.py $w $LIBS > $OUT/ab_${w}_$m.txt 2>&1
    grep -A8 "^workload" $OUT/ab_${w}_$m.txt
done
