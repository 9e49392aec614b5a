#!/bin/bash
# Round 6, call 6: WindowSolverBatch wall time over repetitions and host thread counts
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c6
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
nproc; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core" 
timeout 300 python profiles/run_solve_batch_sweep.py 10 0 4 8 12 16 24 2>&1 | tee $O/sweep.txt
