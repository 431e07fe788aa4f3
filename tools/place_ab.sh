for rep in 1 2; do
for w in c2 c3 c4; do
for t in "--placement-tune" ""; do
python bench.py --workload $w --no-cpu-baseline --no-stepper --steps 200 $t > gpurun_out/pl_tmp.json 2>/dev/null || exit 1
python - "$w" "$t" <<'PY'
import json,sys
d=json.load(open("gpurun_out/pl_tmp.json")); r=d["roofline"]
print(sys.argv[1], sys.argv[2] or "first-come", "ms_per_step %.4f kernel_ms %.4f med %.4f min %.4f fused %.4f probe %.4f place %s" % (d["ms_per_step"], r["kernel_ms"], r["kernel_ms_median"], r["kernel_ms_min"], r["fused_dt_kernel_ms"], r["stream_probe"]["ms"], d["placement_tuning"]))
PY
done; done; done
