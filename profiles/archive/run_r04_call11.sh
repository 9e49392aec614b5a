#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4k}
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_device_tracker.py -m gpu -x -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --engine device"
for cfgs in "12 768" "4 768" "8 1536"; do
  set -- $cfgs
  timeout 300 python bench.py $Q --groups $1 --streams $2 --details $O/d_${1}x${2}.json > $O/d_${1}x${2}.line 2> $O/d_${1}x${2}.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py $Q --groups 4 --streams 768 --details $O/q2_4x768.json > $O/q2_4x768.line 2> $O/q2_4x768.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["config"]["engine"][:12], d["host"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".line", ".err")).read()[-500:])
PY
