#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4h}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --engine device"
for cfgs in "12 768" "12 1536" "8 1024" "8 1536" "6 768" "6 1536" "4 1024"; do
  set -- $cfgs
  timeout 300 python bench.py $Q --groups $1 --streams $2 --details $O/d${1}x${2}.json > $O/d${1}x${2}.line 2> $O/d${1}x${2}.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py $Q --groups 8 --streams 1536 --details $O/q2.json > $O/q2.line 2> $O/q2.err
ICG_BENCH_TIMED_CPUS=1 timeout 300 python bench.py $Q --groups 8 --streams 1536 --details $O/q1.json > $O/q1.line 2> $O/q1.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["host"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".line", ".err")).read()[-500:])
PY
