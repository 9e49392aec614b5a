#!/bin/bash
# Round 6, call 8: dead kernel variants deleted (LK pair / set-up cache, tile pyramid, CLAHE legacy), LK at 96 VGPRs without scratch:
# whole GPU suite (incl. tests/test_gpu_switches.py) and the driver's command
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c8
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 900 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -25 > $O/gputests.txt; cat $O/gputests.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
tail -3 $O/driver.err
python3 - <<PY
import json
d = json.loads(open("$O/driver_line.json").read().strip().splitlines()[-1])
print(d["value"], d.get("value_200steps"), d["ms_per_step"], d["roofline"])
print({k: d.get(k) for k in ("solve", "marg", "c4") if k in d})
PY
