#!/bin/bash
# Round 6, call 7: the serial part of a batched solve — per-call split of the assembly / reduction entry point (ICG_ABI_DEBUG)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c7
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
ICG_ABI_DEBUG=1 ICG_SOLVER_DEBUG=1 timeout 300 python profiles/run_solve_batch_only.py > $O/abi.out 2> $O/abi.err
grep "W=256" $O/abi.err | tail -12; grep WindowSolverBatch $O/abi.err | tail -2; cat $O/abi.out
