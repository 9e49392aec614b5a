#!/bin/bash
# final tree: whole GPU suite, the driver's command on the device engine (parity witness on) and the same on 2 confined CPUs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4final5
mkdir -p $O
cd $R
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests.txt; cat $O/gputests.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --details $O/device_details.json > $O/device_line.json 2> $O/device.err
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --details $O/q2_details.json > $O/q2.line 2> $O/q2.err
ICG_BENCH_TIMED_CPUS=1 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --engine device --no-reproj --no-cpu-baseline --no-profile-pass --no-parity --details $O/q1_details.json > $O/q1.line 2> $O/q1.err
python - <<PY
import json
for f in ("device_line.json", "q2.line", "q1.line"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["host"], (d.get("parity") or {}).get("ok"))
PY
