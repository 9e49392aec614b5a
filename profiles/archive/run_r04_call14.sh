#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r4n}
mkdir -p $O
cd $R
Q="--no-reproj --no-cpu-baseline --no-profile-pass --no-parity"
for G in 4 8 12 16; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --groups $G --details $O/s20_g$G.json > $O/s20_g$G.line 2> $O/s20_g$G.err
done
for G in 8 12; do
  ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --steps 60 --warmup 5 $Q --groups $G --details $O/q2_g$G.json > $O/q2_g$G.line 2> $O/q2_g$G.err
done
ICG_BENCH_TIMED_CPUS=2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $Q --groups 8 --details $O/q2s20_g8.json > $O/q2s20_g8.line 2> $O/q2s20_g8.err
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/*.line")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["steps"], d["config"]["groups_per_gpu"], d["host"], d["step_stats"])
    except Exception as e:
        print(f, "failed", e)
PY
