#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4r}
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_device_tracker.py -m gpu -x -q 2>&1 | tail -4 > $O/gputests.txt; cat $O/gputests.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-parity --details $O/dev_details.json > $O/dev.line 2> $O/dev.err
python - <<PY
import json
d = json.load(open("$O/dev_details.json"))
print("value", d["value"], d["host_ms_per_step"]["cpu_cores_busy"], d["host_ms_per_step"]["group_step_ms_min_mean_max"])
kc = d["kernel_ceiling"]; print("sum", kc["exclusive_us_per_frame"])
for k, v in kc["kernels"].items(): print("%-22s %5.2f %9.2f %8.4f" % (k, v["launches_per_step"], v["exclusive_us_per_launch"], v["exclusive_us_per_frame"]))
PY
