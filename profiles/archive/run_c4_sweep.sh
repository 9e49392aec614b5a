for cfg in "16 64" "32 128" "48 192" "32 256"; do
  set -- $cfg
  timeout 250 python bench.py --width 1920 --height 1080 --features 500 --ring 16 --steps 60 --warmup 10 --groups $1 --streams $2 --no-reproj --no-cpu-baseline --no-profile-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); h=d['host_ms_per_step']; print('C4 groups $1 streams $2:', d['value'], 'fps  cores_busy', h['cpu_cores_busy'], ' device_execute', h['device_execute'], ' host_logic', h['host_logic'])"
done
