#!/bin/bash
# kernel-only ceiling table (exclusive time per kernel) of the current build at the default launch shape; $1 = output tag
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r4c}
O=$R/gpurun_out/$T
mkdir -p $O
cd $R
timeout 200 python bench.py --steps 40 --warmup 5 --no-reproj --no-cpu-baseline --no-parity --details $O/details.json > $O/line.json 2> $O/err.txt
python - <<PY
import json
d = json.load(open("$O/details.json"))
print("value", d["value"], "busy", d["host_ms_per_step"]["cpu_cores_busy"])
c = d["kernel_ceiling"]
print("sum exclusive us/frame", c["exclusive_us_per_frame"], "ceiling", c["ceiling_frames_per_s"])
for k, v in c["kernels"].items():
    print("%-24s launches/step %6.2f  us/launch %8.2f  us/frame %7.4f" % (k, v["launches_per_step"], v["exclusive_us_per_launch"], v["exclusive_us_per_frame"]))
print(json.dumps(d["rates"]))
PY
