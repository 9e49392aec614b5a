#!/bin/bash
# Round-4 final lines on the final bench.py (200 priming frames): the driver's command, and the same command under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final2
mkdir -p $O
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --details $O/driver_details.json > $O/driver_line.json 2> $O/driver.err
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --engine table --details $O/table20_details.json > $O/table20.line 2> $O/table20.err
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --details $O/q2s20_details.json > $O/q2s20.line 2> $O/q2s20.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/r04kt -o kt -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-replay --details $O/kt_details.json > $O/kt_line.json 2> $O/kt.err
find /tmp/r04kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
KT=$(find /tmp/r04kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python $R/profiles/analyze_trace.py "$KT" > $O/queue_view.json
cd $R
python - <<PY
import json, glob
for f in ["$O/driver_line.json", "$O/table20.line", "$O/q2s20.line", "$O/kt_line.json"]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["config"]["groups_per_gpu"], d["config"]["engine"][:12], d["host"], (d.get("parity") or {}).get("ok"), len(open(f).read()))
    except Exception as e:
        print(f, "failed", e)
PY
head -12 $O/kernel_stats.csv | cut -c1-150
