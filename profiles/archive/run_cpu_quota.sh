#!/bin/bash
# Host-core sensitivity of the front-end throughput: the default bench under taskset with 2, 4, 8 and all CPUs.
# bench.py sizes its stream groups from the CPUs it may use (sharding.host_plan), so each line is what ONE rank gets on a node where
# that many host cores are available per GPU.  Prints: cpus, groups, streams, frames/s, cpu_cores_busy.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
for N in 2 4 8 16; do
  taskset -c 0-$((N-1)) python $R/bench.py --steps 100 --warmup 10 --no-reproj --no-cpu-baseline --no-profile-pass > $OUT/quota_$N.json 2> $OUT/quota_$N.err
  python - <<PY
import json
d = json.load(open("$OUT/quota_$N.json"))
print("cpus", $N, "groups", d["config"]["groups_per_gpu"], "streams", d["config"]["streams_per_gpu"], "frames/s", d["value"],
      "cpu_cores_busy", d["host_ms_per_step"]["cpu_cores_busy"], "job_step_ms median/p95", d["step_stats"]["job_step_ms"]["median"], d["step_stats"]["job_step_ms"]["p95"])
PY
done
