#!/bin/bash
# Round 5, call 10: the template set-up cache of round 3 (ICG_LK_REUSE=1, table engine) re-measured now that the front-end is instruction-bound
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c10
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --engine table"
for tag in off1 on1 off2 on2; do
  case $tag in on*) export ICG_LK_REUSE=1;; *) export ICG_LK_REUSE=0;; esac
  timeout 240 python bench.py $LIGHT --details $O/${tag}_details.json > $O/${tag}_line.json 2> $O/${tag}.err
  python - <<PY
import json
d = json.loads(open("$O/${tag}_line.json").read().strip().splitlines()[-1])
dd = json.load(open("$O/${tag}_details.json"))
kc = dd.get("kernel_ceiling") or {}
print("${tag}", d["value"], dd.get("lk_setup_reuse"), (kc.get("kernels") or {}).get("lk_track_fb"), kc.get("exclusive_us_per_frame"))
PY
done
