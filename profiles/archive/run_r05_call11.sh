#!/bin/bash
# Round 5, call 11: the container order as a sort (order_extend_parallel) on the device: device-tracker tests, then the stage times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c11
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 400 python -m pytest tests/test_gpu_device_tracker.py tests/test_gpu_stream.py tests/test_parity_at_scale.py tests/test_gpu_c4.py -m gpu -x -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
timeout 300 python bench.py $LIGHT --details $O/dev_details.json > $O/dev_line.json 2> $O/dev.err
python - <<PY
import json
d = json.loads(open("$O/dev_line.json").read().strip().splitlines()[-1])
dd = json.load(open("$O/dev_details.json"))
kc = dd.get("kernel_ceiling") or {}
print(d["value"], d.get("value_200steps"), d["config"]["engine"][:14], d["host"].get("cpu_cores_busy"), kc.get("exclusive_us_per_frame"))
print({k: [v.get("launches_per_step"), round(v["exclusive_us_per_launch"], 1)] for k, v in (kc.get("kernels") or {}).items() if "stage" in k or "lk_track" in k})
PY
