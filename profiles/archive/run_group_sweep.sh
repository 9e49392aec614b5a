#!/bin/bash
# sweep of stream groups x streams per group [x host threads per group] for the front-end bench (one summary line per configuration)
#   SWEEP="48 384;24 384 2" bash profiles/run_group_sweep.sh     (third field: --host-threads, default 1)
IFS=";" read -ra CFGS <<< "${SWEEP:-32 256;40 320;48 384;56 448;64 512}"
for cfg in "${CFGS[@]}"; do
  IFS=" " read -r g s t <<< "$cfg"; set -- $g $s ${t:-1}
  timeout 200 python bench.py --steps 120 --warmup 30 --groups $1 --streams $2 --host-threads $3 --no-reproj --no-cpu-baseline --no-profile-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('groups $1 streams $2 host-threads $3:', d['value'], 'fps  cores_busy', h['cpu_cores_busy'], ' device_execute', h['device_execute'], ' host_logic', h['host_logic'], ' group_step', h['group_step_ms_min_mean_max'])"
done
