#!/bin/bash
# device-resident tracker: first throughput numbers (groups x streams sweep, full host / 2 confined CPUs / 1 CPU)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r4e
mkdir -p $O
cd $R
Q="--steps 60 --warmup 5 --no-reproj --no-cpu-baseline --no-profile-pass --engine device"
run() { # tag, extra env, args
  tag=$1; shift
  env "$@" > /dev/null 2>&1
}
timeout 200 python bench.py $Q --groups 12 --streams 768 --details $O/d12x64.json > $O/d12x64.json.line 2> $O/d12x64.err
for cfgs in "24 768" "8 768" "16 1024" "32 1024"; do
  set -- $cfgs
  timeout 200 python bench.py $Q --no-parity --groups $1 --streams $2 --details $O/d${1}x.json > $O/d${1}x.line 2> $O/d${1}x.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 200 python bench.py $Q --no-parity --groups 12 --streams 768 --details $O/q2.json > $O/q2.line 2> $O/q2.err
ICG_BENCH_TIMED_CPUS=1 timeout 200 python bench.py $Q --no-parity --groups 12 --streams 768 --details $O/q1.json > $O/q1.line 2> $O/q1.err
ICG_BENCH_TIMED_CPUS=2 timeout 200 python bench.py $Q --no-parity --groups 24 --streams 768 --details $O/q2g24.json > $O/q2g24.line 2> $O/q2g24.err
timeout 200 python bench.py --steps 40 --warmup 5 --no-reproj --no-cpu-baseline --no-parity --engine device --details $O/prof.json > $O/prof.line 2> $O/prof.err
python - <<PY
import json
for n in ("d12x64.json.line","d24x.line","d8x.line","d16x.line","d32x.line","q2.line","q1.line","q2g24.line","prof.line"):
    try:
        d = json.loads(open("$O/%s" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["config"]["groups_per_gpu"], d["config"]["streams_per_gpu"], d["host"], (d.get("parity") or {}).get("ok"))
    except Exception as e:
        print(n, "failed", e, open("$O/%s" % n.replace(".line", ".err").replace(".json.err",".err")).read()[-600:])
try:
    d = json.load(open("$O/prof.json"))
    for k, v in sorted(d["kernels"].items(), key=lambda t: -t[1]["total_ms"]): print("%-24s launches %6d avg_us %9.2f total_ms %9.2f" % (k, v["launches"], v["avg_us"], v["total_ms"]))
except Exception as e: print("no kernel table", e)
PY
