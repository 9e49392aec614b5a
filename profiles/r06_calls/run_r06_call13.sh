#!/bin/bash
# Round 6, call 13: launch shape of the device engine re-swept on the round-6 kernels (768 streams, groups x streams per launch), 60 timed steps
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c13
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-reproj --no-c4 --no-engine-twin --no-cpu-baseline --no-parity --no-replay --no-profile-pass --no-dist"
for g in 4 3 6 2 8 4; do
  timeout 300 python bench.py $LIGHT --groups $g --details $O/g${g}_details.json > $O/g${g}_line.json 2> $O/g${g}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$O/g${g}_line.json").read().splitlines() if l.startswith("{")][-1])
print("groups", $g, "streams/launch", d["config"]["streams_per_gpu"] // $g, d["value"], d["ms_per_step"], d["host"]["cpu_cores_busy"])
PY
done
for s in 1024 1536; do
  timeout 300 python bench.py $LIGHT --groups 4 --streams $s --details $O/s${s}_details.json > $O/s${s}_line.json 2> $O/s${s}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$O/s${s}_line.json").read().splitlines() if l.startswith("{")][-1])
print("streams", $s, "groups 4", d["value"], d["ms_per_step"])
PY
done
