#!/bin/bash
# Round 5, call 8: launch shapes after the queue view of call 7 (no LK launch in flight 24 % of the time with 12 groups, 42 % with 4): more groups
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c8
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --no-profile-pass"
run() {
  tag=$1; shift
  timeout 240 python bench.py $LIGHT "$@" --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}", d["value"], d["config"]["engine"][:12], d["config"]["streams_per_gpu"], d["config"]["groups_per_gpu"], d["host"].get("cpu_cores_busy"))
except Exception as e:
    print("${tag}", "failed", e)
PY
}
run tab_12x64
run tab_16x48 --groups 16
run tab_24x32 --groups 24
run tab_16x64 --groups 16 --streams 1024
run tab_24x48 --groups 24 --streams 1152
run tab_32x32 --groups 32 --streams 1024
run dev_8x128 --engine device --groups 8 --streams 1024
run dev_8x192 --engine device --groups 8 --streams 1536
run dev_12x128 --engine device --groups 12 --streams 1536
