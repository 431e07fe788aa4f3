# A/B of the (reverted) multi-pass column launch: LH_TUNE passes= existed only in the experiment recorded in
# profiles/round2_resident_workgroups_negative.txt; kept as the record of how it was measured.
for w in c2 c3 c4 c5 f3c64; do for t in "passes=-1" "passes=0" "passes=2"; do LH_TUNE=$t python bench.py --workload $w --no-cpu-baseline --no-stepper --steps 150 > gpurun_out/pa.json 2>/dev/null || { echo FAIL $w $t; continue; }; python -c "
import json;d=json.load(open('gpurun_out/pa.json'));r=d['roofline']
print('$w','$t','value %.4e ms/step %.4f kern %.4f min %.4f fused %.4f'%(d['value'],d['ms_per_step'],r['kernel_ms'],r['kernel_ms_min'],r['fused_dt_kernel_ms']))"; done; done
