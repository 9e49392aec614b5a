#!/bin/bash
# Round 6, call 33: the driver's command twice more on the final tree (another box): the spread of `value` / `value_200steps`
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c33
mkdir -p $O
cd $R
for rep in 3 4; do
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_run${rep}_details.json > $O/driver_run${rep}_line.json 2> $O/driver_run${rep}.err
  python3 - <<PY
import json
d = json.loads([l for l in open("$O/driver_run${rep}_line.json").read().splitlines() if l.startswith("{")][-1])
print("run $rep", d["value"], d.get("value_200steps"), d["ms_per_step"], (d.get("parity") or {}).get("ok"), d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["roofline"]["traffic"],
      (d.get("solve") or {}).get("batched", {}).get("value"), (d.get("marg") or {}).get("batched", {}).get("value"), (d.get("c4") or {}).get("frontend", {}).get("value"))
PY
done
