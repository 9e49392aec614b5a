#!/bin/bash
# Round 5, call 16: rocprofv3 --kernel-trace --stats of the HEADLINE REGION alone (the driver's steps / warm-up on the default engine, every other block
# and the exclusive-time pass off): every k_lk_track_fb launch of the trace has the launch shape the bench line's roofline block measures
# (192 streams per launch, four groups in flight), so the trace's average and bench.py's own avg_launch_us (HIP events) describe the same thing.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r5c16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-reproj --no-parity --no-engine-twin --no-profile-pass --details $O/details.json > $O/line.json 2> $O/err.txt
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
python - <<PY
import csv, json
d = json.loads(open("$O/line.json").read().strip().splitlines()[-1])
print("bench:", d["value"], "avg_launch_us", d["roofline"]["avg_launch_us"])
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:8]:
    print(r["Name"][:36], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
