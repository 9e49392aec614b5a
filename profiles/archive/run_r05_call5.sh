#!/bin/bash
# Round 5, call 5: same-box A/B of the CLAHE forms (ICG_CLAHE_LEGACY=1 vs default, interleaved) and the launch shapes of both engines
# after the kernel diets (groups x streams per group).  Light benches: 60 timed steps, no parity / twin / back-end blocks.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c5
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --no-profile-pass"
run() { # tag, env assignments..., then bench args after --
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 240 python bench.py $LIGHT "$@" --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/${tag}_line.json").read().strip().splitlines()[-1])
    print("${tag}", d["value"], d["config"]["engine"][:12], d["config"]["streams_per_gpu"], d["config"]["groups_per_gpu"], d["host"].get("cpu_cores_busy"))
except Exception as e:
    print("${tag}", "failed", e)
PY
}
run legacy1 ICG_CLAHE_LEGACY=1 --
run new1 ICG_X=0 --
run legacy2 ICG_CLAHE_LEGACY=1 --
run new2 ICG_X=0 --
run dev_4x192 ICG_X=0 -- --engine device
run dev_8x96 ICG_X=0 -- --engine device --groups 8
run dev_12x64 ICG_X=0 -- --engine device --groups 12
run dev_8x192 ICG_X=0 -- --engine device --groups 8 --streams 1536
run dev_6x192 ICG_X=0 -- --engine device --groups 6 --streams 1152
run tab_16x64 ICG_X=0 -- --groups 16 --streams 1024
run tab_12x96 ICG_X=0 -- --groups 12 --streams 1152
