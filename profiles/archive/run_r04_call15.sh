#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4o}
mkdir -p $O
cd $R
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
for G in 4 12; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --groups $G --details $O/s20_g$G.json > $O/s20_g$G.line 2> $O/s20_g$G.err
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --engine table --details $O/s20_table.json > $O/s20_table.line 2> $O/s20_table.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["config"]["groups_per_gpu"], d["host"], d["step_stats"])
    except Exception as e:
        print(f, "failed", e)
PY
