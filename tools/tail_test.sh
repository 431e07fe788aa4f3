# Per-launch fixed cost of the column launch: bench.py kernel time against the number of columns
# (524288 = one full round of 512 workgroups x 1024 lanes on MI355X).  Run on the GPU box via gpurun.
for n in 524288 786432 1000000 1048576 1572864; do python bench.py --ncols $n --no-cpu-baseline --no-stepper --steps 150 > gpurun_out/tt.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tt.json'));r=d['roofline'];n=$n
print(n, 'kernel_ms %.4f min %.4f fused %.4f'%(r['kernel_ms'], r['kernel_ms_min'], r['fused_dt_kernel_ms']), 'cells/s (kernel) %.3e'%(n*64/(r['kernel_ms']*1e-3)))"; done
