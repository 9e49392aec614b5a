#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4j}
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_device_tracker.py -m gpu -x -q 2>&1 | tail -6 > $O/gputests.txt; cat $O/gputests.txt
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
# the world-8 plan on one GPU: every thread confined to 2 CPUs, engine by host_plan's rule = device, candidates for its launch shape
for cfgs in "6 768" "4 768" "8 768" "12 768"; do
  set -- $cfgs
  ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py $Q --engine device --groups $1 --streams $2 --details $O/q2_${1}x${2}.json > $O/q2_${1}x${2}.line 2> $O/q2_${1}x${2}.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py $Q --engine table --groups 8 --streams 768 --details $O/q2_table.json > $O/q2_table.line 2> $O/q2_table.err
# table engine, 16 cores: host-looped RANSAC vs the one-launch form
timeout 300 python bench.py $Q --engine table --details $O/t_hostloop.json > $O/t_hostloop.line 2> $O/t_hostloop.err
ICG_RANSAC_DEVICE_LOOP=1 timeout 300 python bench.py $Q --engine table --details $O/t_devloop.json > $O/t_devloop.line 2> $O/t_devloop.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["config"]["engine"][:12], d["host"])
    except Exception as e:
        print(f, "failed", e, open(f.replace(".line", ".err")).read()[-500:])
PY
