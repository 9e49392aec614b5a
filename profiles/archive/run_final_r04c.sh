#!/bin/bash
# the driver's command on the final tree (engine by host share: the track table on a 16-core rank) + profiles/collect.sh of the same default
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final3
mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
python - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print("driver", d["value"], d["config"]["engine"][:14], d["host"], d["parity"]["ok"], len(json.dumps(d)))
PY
bash profiles/collect.sh r04 > $O/collect.log 2>&1
tail -3 $O/collect.log
