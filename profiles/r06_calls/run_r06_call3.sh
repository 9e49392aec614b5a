#!/bin/bash
# Round 6, call 3: where the reduction kernel's time goes (zero-copy S over PCIe vs S resident + batched Cholesky on the device)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r6c3
mkdir -p $O
cd $R
export PYTHONPATH=$R/tests:$R/ic-gvins_amd:$R
timeout 600 python -m pytest tests/test_gpu_reproj.py tests/test_gpu_solver.py tests/test_gpu_backend.py tests/test_gpu_zz_marg_batch.py tests/test_gpu_vio_replay.py tests/test_gpu_c4.py -m gpu -q -x 2>&1 | tail -5 > $O/backend_tests.txt; cat $O/backend_tests.txt
for mode in host dev; do
  if [ $mode = dev ]; then export ICG_SOLVER_DEVICE_CHOLESKY=1; fi
  ICG_SOLVER_DEBUG=1 timeout 200 python profiles/run_solve_batch_only.py > $O/solve_$mode.out 2> $O/solve_$mode.err
  grep -v "^$" $O/solve_$mode.err | tail -2; cat $O/solve_$mode.out
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o sb -- python $R/profiles/run_solve_batch_only.py > $O/sb.out 2> $O/sb.err
  f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $O/sb_kernel_stats_$mode.csv; fi
  cd $R
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/sb_kernel_stats_$mode.csv")):
    print(r['Name'][:40].ljust(40), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:9.1f} us avg", f"{float(r['MaxNs'])/1e3:9.1f} max", r['Percentage'])
PY
done
