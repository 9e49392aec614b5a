#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4p}
mkdir -p $O
cd $R
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
for G in 4 8; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --groups $G --details $O/s20_g$G.json > $O/s20_g$G.line 2> $O/s20_g$G.err
  ICG_GROUP_STAGGER=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --groups $G --details $O/s20_g${G}_nostagger.json > $O/s20_g${G}_nostagger.line 2> $O/s20_g${G}_nostagger.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["config"]["groups_per_gpu"], d["host"], d["step_stats"])
    except Exception as e:
        print(f, "failed", e)
PY
