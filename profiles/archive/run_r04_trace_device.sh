#!/bin/bash
# kernel trace + stats of the device engine at 8 x 192 + queue-level view
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=${1:-r4i}
OUT=$R/gpurun_out/$T; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -o kt -- python $R/bench.py --engine device --groups ${2:-8} --streams ${3:-1536} --no-cpu-baseline --no-reproj --no-profile-pass --no-parity --steps 40 --warmup 10 --details $OUT/details.json > $OUT/bench.json 2> $OUT/err.txt
f=$(find /tmp/kts -name "*kernel_trace.csv" | head -1)
python $R/profiles/analyze_trace.py "$f" > $OUT/queue_view.json
find /tmp/kts -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python - <<PY
import json, csv
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1]); print("value under rocprof", d["value"], d["host"])
rows = list(csv.DictReader(open("$OUT/kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]: print("%-40s calls %7s avg_us %9.1f  %5.1f%%" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
q = json.load(open("$OUT/queue_view.json")); print(json.dumps(q)[:1500])
PY
