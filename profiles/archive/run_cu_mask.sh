run() { timeout 150 python bench.py --steps 150 --warmup 30 --no-reproj --no-cpu-baseline --no-profile-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['host_ms_per_step']['cpu_cores_busy'], d['host_ms_per_step']['device_execute'], d['host_ms_per_step']['group_step_ms_min_mean_max'])"; }
run full
export ROC_GLOBAL_CU_MASK=0xffffffffffffffffffffffffffffffff
run half128
unset ROC_GLOBAL_CU_MASK
export HSA_CU_MASK=0:0-127
run hsa_half
