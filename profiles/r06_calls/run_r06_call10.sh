#!/bin/bash
# Round 6, call 10: layout() and the chi-square count on the pool; whole-solve clock
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c10
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 200 python profiles/run_solve_batch_sweep.py 12 0 2>&1 | tee $O/sweep.txt
ICG_SOLVER_DEBUG=1 timeout 200 python profiles/run_solve_batch_only.py 2>&1 | grep "256 windows" | tail -2 | tee $O/phases.txt
