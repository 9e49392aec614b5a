#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4q}
mkdir -p $O
cd $R
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
for P in 48 120 200; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --prime $P $Q --details $O/s20_p$P.json > $O/s20_p$P.line 2> $O/s20_p$P.err
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["prime"], d["host"], d["step_stats"])
    except Exception as e:
        print(f, "failed", e)
PY
