#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final4
mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --details $O/device_details.json > $O/device_line.json 2> $O/device.err
python - <<PY
import json
for f in ("driver_line.json", "device_line.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["config"]["engine"][:14], d["host"], d["parity"]["ok"], len(json.dumps(d)), json.dumps(d["roofline"])[:600])
PY
