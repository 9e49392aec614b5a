#!/bin/bash
# Round 6, call 5: device call beside the host half of a linearization (WindowSolverBatch)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c5
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests/test_gpu_reproj.py tests/test_gpu_solver.py tests/test_gpu_backend.py tests/test_gpu_zz_marg_batch.py tests/test_gpu_vio_replay.py tests/test_gpu_c4.py -m gpu -q -x 2>&1 | tail -15 > $O/backend_tests.txt; cat $O/backend_tests.txt
ICG_SOLVER_DEBUG=1 timeout 200 python profiles/run_solve_batch_only.py > $O/solve.out 2> $O/solve.err
grep -v "^$" $O/solve.err | tail -4; cat $O/solve.out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- python $R/profiles/run_solve_batch_only.py > $O/sb.out 2> $O/sb.err
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/sb_kernel_stats.csv; fi
python3 - <<PY
import csv
for r in csv.DictReader(open("$O/sb_kernel_stats.csv")):
    print(r['Name'][:40].ljust(40), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us avg", f"{float(r['MaxNs'])/1e3:9.1f} max", r['Percentage'])
PY
