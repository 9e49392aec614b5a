#!/bin/bash
# Round 5, call 6: whole GPU suite on the current tree (LDS caps, LK / pyramid / CLAHE diets, RANSAC subsets, order_extend, shared marginalizations
# by default), then the device engine's exclusive kernel times (stage kernels after the order_extend change).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c6
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 > $O/gputests.txt; cat $O/gputests.txt
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin"
timeout 300 python bench.py $LIGHT --engine device --details $O/dev_details.json > $O/dev_line.json 2> $O/dev.err
python - <<PY
import json
d = json.loads(open("$O/dev_line.json").read().strip().splitlines()[-1])
dd = json.load(open("$O/dev_details.json"))
kc = dd.get("kernel_ceiling") or {}
print(d["value"], d.get("value_200steps"), d["config"]["engine"][:14], d["host"].get("cpu_cores_busy"), kc.get("exclusive_us_per_frame"), (d.get("roofline") or {}).get("trk_stage"))
print({k: [v.get("launches_per_step"), round(v["exclusive_us_per_launch"], 1)] for k, v in (kc.get("kernels") or {}).items()})
PY
