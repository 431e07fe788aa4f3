#!/bin/bash
# Run on the GPU box (via gpurun): the rocprofv3 kernel-trace + PMC passes of tools/gpu_profile.sh for a list
# of workloads ("c2", "c2_nozero" = the contract-traffic regime, ...), then the instruction mixes.
# usage: tools/profile_round.sh <tag> <workload>... [-- <instmix workload>...]
TAG=$1; shift
cd $GRAFT_REPO_ROOT
mix=0
for w in "$@"; do
    if [ "$w" = "--" ]; then mix=1; continue; fi
    wl=${w%_nozero}; extra=""; [ "$wl" != "$w" ] && extra="--no-known-zero"
    if [ $mix = 0 ]; then
        bash tools/gpu_profile.sh ${TAG}_$w --workload $wl $extra > /dev/null 2>&1 || echo "profile $w FAILED"
        grep -E "dominant kernel, last|== dominant" gpurun_out/prof_${TAG}_$w/summary.txt | cut -c1-160
    else
        bash tools/gpu_instmix.sh ${TAG}_$w --workload $wl $extra > gpurun_out/instmix_${TAG}_$w.txt 2>&1 || echo "instmix $w FAILED"
        grep -E "^== |SQ_INSTS_VALU  " gpurun_out/instmix_${TAG}_$w.txt | cut -c1-150
    fi
done
