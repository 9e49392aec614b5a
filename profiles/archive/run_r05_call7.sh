#!/bin/bash
# Round 5, call 7: rocprofv3 kernel traces (no counters) of the light bench on both engines -> queue views (how often LK is in flight, kernels side by side)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c7
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
LIGHT="--gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --no-profile-pass"
for eng in table device; do
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$eng -o kt -- python $R/bench.py $LIGHT --engine $eng --details $O/${eng}_details.json > $O/${eng}_line.json 2> $O/${eng}.err
  KT=$(find $O/kt_$eng -name "*kernel_trace.csv" | head -1)
  [ -n "$KT" ] && python $R/profiles/analyze_trace.py "$KT" > $O/queue_view_$eng.json
  find $O/kt_$eng -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$eng.csv \;
  rm -rf $O/kt_$eng
  python - <<PY
import json
d = json.loads(open("$O/${eng}_line.json").read().strip().splitlines()[-1])
q = json.load(open("$O/queue_view_$eng.json"))
print("$eng", d["value"], q.get("mean_kernels_in_flight"), q.get("mean_queue_busy_frac"), q.get("lk_in_flight_histogram"), q.get("in_flight_histogram"))
print({k: (v["n"], v["mean_us"], v["share_of_queue_time"]) for k, v in list(q.get("kernels", {}).items())[:14]})
PY
done
