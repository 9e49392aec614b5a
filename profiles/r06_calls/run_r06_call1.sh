#!/bin/bash
# Round 6, call 1: the atomic-free assembly (csrc/reproj.hip k_asm_runs / k_asm_camera / k_asm_landmarks) on the MI355X.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/run_r06_call1.sh'
# 1. the back-end GPU tests with bitwise=True (lock-step replays, marginalization batch incl. the per-window oracle check)
# 2. WindowSolverBatch phase split at 256 C2 windows + rocprofv3 kernel stats of the same
# 3. the whole GPU suite
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c1
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests/test_gpu_reproj.py tests/test_gpu_solver.py tests/test_gpu_backend.py tests/test_gpu_zz_marg_batch.py tests/test_gpu_vio_replay.py -m gpu -q --durations=8 2>&1 | tail -40 > $O/backend_tests.txt; cat $O/backend_tests.txt
ICG_SOLVER_DEBUG=1 timeout 200 python profiles/run_solve_batch_only.py > $O/solve.out 2> $O/solve.err
grep -v "^$" $O/solve.err | tail -8; cat $O/solve.out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- python $R/profiles/run_solve_batch_only.py > $O/sb.out 2> $O/sb.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/sb_kernel_stats.csv; cat $O/sb_kernel_stats.csv | cut -c1-60,200-400 | head -20; fi
cd $R
timeout 600 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 > $O/gputests.txt; cat $O/gputests.txt
