#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4g}
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_device_tracker.py -m gpu -x -q 2>&1 | tail -30 > $O/devtrk.txt; cat $O/devtrk.txt
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --engine device"
timeout 200 python bench.py $Q --groups 12 --streams 768 --details $O/d12x64.json > $O/d12x64.line 2> $O/d12x64.err
timeout 200 python bench.py $Q --no-parity --groups 16 --streams 1024 --details $O/d16x.json > $O/d16x.line 2> $O/d16x.err
ICG_BENCH_TIMED_CPUS=2 timeout 200 python bench.py $Q --no-parity --groups 12 --streams 768 --details $O/q2.json > $O/q2.line 2> $O/q2.err
timeout 200 python bench.py --steps 40 --warmup 5 --no-reproj --no-cpu-baseline --no-parity --engine device --details $O/prof.json > $O/prof.line 2> $O/prof.err
python - <<PY
import json
for n in ("d12x64.line","d16x.line","q2.line","prof.line"):
    try:
        d = json.loads(open("$O/%s" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["host"], (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "failed", e, open("$O/%s" % n.replace(".line", ".err")).read()[-600:])
try:
    d = json.load(open("$O/prof.json"))
    for k, v in sorted(d["kernels"].items(), key=lambda t: -t[1]["total_ms"]): print("%-24s launches %6d avg_us %9.2f total_ms %9.2f" % (k, v["launches"], v["avg_us"], v["total_ms"]))
except Exception as e: print("no kernel table", e)
PY
