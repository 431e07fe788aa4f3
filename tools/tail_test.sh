for n in 1000000 1048576 524288 786432 1572864; do python bench.py --ncols $n --no-cpu-baseline --no-stepper --steps 150 > gpurun_out/tt.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/tt.json'));r=d['roofline'];n=$n
print(n, 'kernel_ms %.4f min %.4f  ns/cell-wave %.3f  fused %.4f'%(r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms']*1e6/(n*64/64)*1.0/1000*1000/1000, r['fused_dt_kernel_ms']), 'cells/s kernel %.3e'%(n*64/(r['kernel_ms']*1e-3)))"; done
