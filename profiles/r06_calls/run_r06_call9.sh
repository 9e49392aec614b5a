#!/bin/bash
# Round 6, call 9: the batched solve inside the bench process (light front-end) against the stand-alone probe — the side thread is persistent now
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c9
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 200 python profiles/run_solve_batch_sweep.py 10 0 2>&1 | tee $O/sweep.txt
ICG_SOLVER_DEBUG=1 timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --prime 20 --streams 64 --no-c4 --no-replay --no-engine-twin --no-cpu-baseline --no-parity > $O/line.json 2> $O/err.txt
grep WindowSolverBatch $O/err.txt | tail -4
python3 - <<PY
import json
d = json.loads(open("$O/line.json").read().strip().splitlines()[-1])
print(d["value"], d.get("solve"), d.get("marg"))
PY
